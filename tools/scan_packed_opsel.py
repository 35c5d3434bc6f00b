#!/usr/bin/env python3
"""Disassemble the gfx950 code objects inside liba2f_hip.so and count the packed-fp32 instructions whose LOW result selects the HIGH register of a source pair
(`v_pk_*_f32 ... op_sel:[..1..]`): the instruction form that misread its operand in the FaceFormer step.  The product build must hold none
(`-fno-slp-vectorize`, audio2face-pytorch_amd/_build.py); `A2F_BUILD_SLP=1` builds hold 86 (24 of them in layernorm_bwd_kernel<4>).
usage: scan_packed_opsel.py [library]     exit code 1 if any is found"""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("A2F_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def scan(lib: str):
    """-> (Counter of instruction classes, Counter kernel -> op_sel(lo<-hi) instructions)."""
    tot, per_kernel = collections.Counter(), collections.Counter()
    with tempfile.TemporaryDirectory() as tmp:
        fat, piece, co = (os.path.join(tmp, n) for n in ("fat.bin", "bundle.bin", "code.co"))
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        data = open(fat, "rb").read()
        offs = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)] + [len(data)]      # one bundle per object file
        for lo, hi in zip(offs[:-1], offs[1:]):
            open(piece, "wb").write(data[lo:hi])
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                   f"--input={piece}", f"--output={co}"], stderr=subprocess.DEVNULL)
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                if m:
                    cur = m.group(1)
                    continue
                if "v_pk_" not in line or "_f32" not in line:
                    continue
                tot["v_pk_*_f32"] += 1
                ops = re.search(r"op_sel:\[([01,]+)\]", line)
                if ops and "1" in ops.group(1):
                    tot["op_sel low<-high"] += 1
                    per_kernel[cur] += 1
                hi_sel = re.search(r"op_sel_hi:\[([01,]+)\]", line)
                if hi_sel and "0" in hi_sel.group(1):
                    tot["op_sel_hi high<-low"] += 1
            tot["kernels"] += len(re.findall(r"^[0-9a-f]+ <.+>:", dis, flags=re.M))
    return tot, per_kernel


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "audio2face-pytorch_amd", "liba2f_hip.so")
    tot, per_kernel = scan(lib)
    print(lib, dict(tot))
    for k, v in per_kernel.most_common(30):
        print(f"  {v:4d}  {k[:140]}")
    sys.exit(1 if tot["op_sel low<-high"] else 0)
