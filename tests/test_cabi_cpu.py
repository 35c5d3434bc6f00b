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
    import ctypes
    # 3 pointers+3*3 int64 ... : the struct is all 8-byte fields followed by ten int32
    assert ctypes.sizeof(_lib.GemmDesc) == 17 * 8 + 10 * 4 + 8


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "audio2face-pytorch_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, os.path.join(dp, f)
