#!/usr/bin/env python3
"""Register / LDS / scratch budget of every kernel inside liba2f_hip.so, read from the code objects' AMDGPU metadata notes (no GPU needed): what decides how many
waves a SIMD and how many workgroups a CU hold.  Round 5 found the bf16x6 GEMM at 161 registers = ONE 8-wave workgroup per CU whatever its LDS use (the second
__launch_bounds__ argument is waves per SIMD, not workgroups per CU); tests/test_build_cpu.py pins the budgets that matter.
usage: kernel_resources.py [library] [substring ...]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("A2F_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def _field(block: str, name: str, default=0):
    m = re.search(r"^\s*\.%s:\s*(\S+)" % re.escape(name), block, flags=re.M)
    if not m:
        return default
    v = m.group(1).strip("'\"")
    return int(v, 0) if re.fullmatch(r"(0x[0-9a-fA-F]+|\d+)", v) else v


def kernels(lib: str):
    """-> list of dicts: name (mangled), vgpr, agpr, sgpr, scratch (bytes per lane), lds (static bytes per workgroup), max_threads, waves_per_simd (by registers)."""
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        fat, piece, co = (os.path.join(tmp, n) for n in ("fat.bin", "bundle.bin", "code.co"))
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        data = open(fat, "rb").read()
        offs = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)] + [len(data)]
        for lo, hi in zip(offs[:-1], offs[1:]):
            open(piece, "wb").write(data[lo:hi])
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                   f"--input={piece}", f"--output={co}"], stderr=subprocess.DEVNULL)
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
            # one YAML list item per kernel under amdhsa.kernels: split on the item marker that precedes `.agpr_count` / `.args`
            body = notes.split("amdhsa.kernels:", 1)[1] if "amdhsa.kernels:" in notes else ""
            for blk in re.split(r"\n\s*- (?=\.(?:agpr_count|args):)", body)[1:]:
                name = _field(blk, "name", "")
                if not name:
                    continue
                v, a = _field(blk, "vgpr_count"), _field(blk, "agpr_count")
                total = v                                             # gfx90a+ metadata: vgpr_count is the unified count (arch VGPRs padded to 4 + AGPRs)
                gran = (total + 7) // 8 * 8
                out.append({"name": name, "vgpr": v, "agpr": a, "sgpr": _field(blk, "sgpr_count"), "scratch": _field(blk, "private_segment_fixed_size"),
                            "lds": _field(blk, "group_segment_fixed_size"), "max_threads": _field(blk, "max_flat_workgroup_size"),
                            "waves_per_simd": min(8, 512 // max(gran, 64)) if total else 8})
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return r.stdout.splitlines()
    except Exception:
        return list(names)


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = sys.argv[1:]
    lib = args.pop(0) if args and args[0].endswith(".so") else os.path.join(here, "audio2face-pytorch_amd", "liba2f_hip.so")
    ks = kernels(lib)
    for k, d in zip(ks, demangle([k["name"] for k in ks])):
        k["demangled"] = d
    ks.sort(key=lambda k: k["demangled"])
    print(f"# {lib}: {len(ks)} kernels; registers = unified VGPR + AGPR count per lane, waves/SIMD by registers alone (512 per SIMD lane, granule 8), LDS = static bytes")
    print(f"# {'regs':>4} {'agpr':>4} {'scratch':>7} {'lds':>6} {'w/SIMD':>6}  kernel")
    for k in ks:
        if args and not any(a in k["demangled"] for a in args):
            continue
        print(f"  {k['vgpr']:4d} {k['agpr']:4d} {k['scratch']:7d} {k['lds']:6d} {k['waves_per_simd']:6d}  {k['demangled'][:150]}")
