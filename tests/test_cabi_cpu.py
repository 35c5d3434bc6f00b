"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/a2f_hip.h declares."""
import os
import re

import a2f_amd
from a2f_amd import _build, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_declared_symbols():
    _build.build_library()
    lib = _lib.load()
    assert lib.a2f_version() == 100
    header = open(os.path.join(ROOT, "include", "a2f_hip.h")).read()
    declared = set(re.findall(r"\b(a2f_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name


def test_gemm_desc_layout_matches_header():
    """The ctypes mirror of a2f_gemm_desc must have the C compiler's layout: compile the header with gcc and compare the
    size and the offset of every field."""
    import ctypes
    import subprocess
    import tempfile
    fields = [f for f, _ in _lib.GemmDesc._fields_]
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "a2f_hip.h"\nint main(void) {\n  printf("%zu\\n", sizeof(a2f_gemm_desc));\n'
    prog += "".join(f'  printf("%zu\\n", offsetof(a2f_gemm_desc, {f}));\n' for f in fields) + "  return 0;\n}\n"
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "layout.c"), os.path.join(d, "layout")
        open(src, "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        out = [int(x) for x in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(_lib.GemmDesc)
    assert out[1:] == [getattr(_lib.GemmDesc, f).offset for f in fields]


def test_relayout_item_layout_matches_header():
    """Same check for a2f_relayout_item (the batched weight re-layout of the Audio2Mesh step)."""
    import ctypes
    import subprocess
    import tempfile
    fields = [f for f, _ in _lib.RelayoutItem._fields_]
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "a2f_hip.h"\nint main(void) {\n  printf("%zu\\n", sizeof(a2f_relayout_item));\n'
    prog += "".join(f'  printf("%zu\\n", offsetof(a2f_relayout_item, {f}));\n' for f in fields) + "  return 0;\n}\n"
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "layout.c"), os.path.join(d, "layout")
        open(src, "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        out = [int(x) for x in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(_lib.RelayoutItem)
    assert out[1:] == [getattr(_lib.RelayoutItem, f).offset for f in fields]


def test_mm_desc_layout_matches_header():
    """Same check for a2f_mm_desc / a2f_mm_operand (the bf16-MFMA GEMM family)."""
    import ctypes
    import subprocess
    import tempfile
    op_fields = [f for f, _ in _lib.MMOperand._fields_]
    fields = [f for f, _ in _lib.MMDesc._fields_]
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "a2f_hip.h"\nint main(void) {\n  printf("%zu\\n", sizeof(a2f_mm_operand));\n'
    prog += "".join(f'  printf("%zu\\n", offsetof(a2f_mm_operand, {f}));\n' for f in op_fields)
    prog += '  printf("%zu\\n", sizeof(a2f_mm_desc));\n'
    prog += "".join(f'  printf("%zu\\n", offsetof(a2f_mm_desc, {f}));\n' for f in fields) + "  return 0;\n}\n"
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "layout.c"), os.path.join(d, "layout")
        open(src, "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        out = [int(x) for x in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    n = len(op_fields)
    assert out[0] == ctypes.sizeof(_lib.MMOperand)
    assert out[1:1 + n] == [getattr(_lib.MMOperand, f).offset for f in op_fields]
    assert out[1 + n] == ctypes.sizeof(_lib.MMDesc)
    assert out[2 + n:] == [getattr(_lib.MMDesc, f).offset for f in fields]


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "audio2face-pytorch_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, os.path.join(dp, f)
