"""CPU: the product library holds no packed-fp32 instruction whose low result selects the high register of a source pair - the instruction form that misread
its operand on MI355X while kernels of other queues overlapped the launch (built out with -fno-slp-vectorize, audio2face-pytorch_amd/_build.py)."""
import importlib.util
import os
import shutil

import pytest

import a2f_amd
from a2f_amd import _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scan_module():
    spec = importlib.util.spec_from_file_location("scan_packed_opsel", os.path.join(ROOT, "tools", "scan_packed_opsel.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_build_flags_leave_the_slp_vectoriser_off():
    if os.environ.get("A2F_BUILD_SLP") == "1":
        pytest.skip("ablation build requested")
    assert "-fno-slp-vectorize" in _build.FLAGS


def test_library_holds_no_packed_fp32_with_a_high_register_selected_for_the_low_result():
    scan = _scan_module()
    if not all(os.path.exists(os.path.join(scan.LLVM, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")):
        pytest.skip("ROCm's LLVM tools are not at " + scan.LLVM)
    lib = _build.build_library()
    tot, per_kernel = scan.scan(lib)
    assert tot["kernels"] > 100, tot                                  # the disassembly was really read (191 kernels at the time of writing)
    assert tot["op_sel low<-high"] == 0, dict(per_kernel)


def _resources_module():
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_register_budgets_that_decide_workgroups_per_cu():
    """The second __launch_bounds__ argument is waves per SIMD: an 8-wave workgroup needs <= 128 registers per lane for TWO workgroups to share a CU.  Round 5 found the
    bf16x6 GEMM (Audio2Mesh's forward convolutions) at 161 registers - one workgroup per CU whatever its LDS use - and the budgets below are what its fix, and the
    kernels that always ran two workgroups per CU, rely on (read from the code objects' metadata, tools/kernel_resources.py)."""
    res = _resources_module()
    if not all(os.path.exists(os.path.join(res.LLVM, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")) or shutil.which("c++filt") is None:
        pytest.skip("ROCm's LLVM tools / c++filt are not available")
    ks = res.kernels(_build.build_library())
    names = res.demangle([k["name"] for k in ks])
    assert len(ks) > 100
    seen = {"x6": 0, "x3": 0, "img": 0}
    for k, name in zip(ks, names):
        if "gemm_mfma_kernel<" in name:
            targs = [a.strip() for a in name.split("gemm_mfma_kernel<", 1)[1].split(">", 1)[0].split(",")]
            prec, cfg = int(targs[2]), int(targs[3])
            if cfg == 3 and prec == 2:                               # bf16x6, 128 x 128 on 8 waves, one LDS stage
                assert targs[5] == "true" and k["vgpr"] <= 128 and k["scratch"] <= 32, (name, k)
                seen["x6"] += 1
            if cfg == 3 and prec == 1:
                assert k["vgpr"] <= 128 and k["scratch"] == 0, (name, k)
                seen["x3"] += 1
        if "gemm_img16_kernel<" in name or "gemm_img_kernel<" in name:   # one 8-wave workgroup per CU by LDS: the whole 256-register budget, never a spill
            assert k["vgpr"] <= 256 and k["scratch"] == 0, (name, k)
            seen["img"] += 1
    assert seen["x6"] >= 4 and seen["x3"] >= 6 and seen["img"] == 3, seen
