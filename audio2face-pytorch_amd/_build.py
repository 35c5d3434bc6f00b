"""Build liba2f_hip.so (gfx950) in-tree with hipcc.  No torch headers, no pybind: plain C ABI.

Every ``csrc/*.hip`` is compiled to its own object under ``build/obj`` (git-ignored), up to A2F_BUILD_JOBS files at a
time, and only when the source (or a shared header) is newer than the object; the objects are then linked into the
shared library.  ``force=True`` rebuilds everything.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "liba2f_hip.so")
OBJ_DIR = os.path.join(PKG_DIR, "..", "build", "obj")
SOURCES = ["capi.hip", "gemm.hip", "wgrad.hip", "voca_trunk.hip", "loss.hip", "pack.hip", "adam.hip", "bn.hip", "head_fused.hip",
           "ff_decoder.hip", "interp.hip", "util.hip", "gemm_mfma.hip", "gemm_img.hip", "rowops.hip", "attention.hip", "featenc.hip", "dropout.hip", "posconv.hip"]
# -fno-slp-vectorize: hipcc's SLP vectoriser packs neighbouring fp32 subtractions / multiplications into v_pk_*_f32.  When the common operand sits in the HIGH register
# of a pair, the LOW result selects it with op_sel[1] = 1 (`v_pk_add_f32 vD, vX, v[M0:M1] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]`).  On MI355X that form was caught
# reading 0 instead of the register in lanes 48..63 of one wave - only while kernels of other queues overlapped the launch, about once per 10 pipelined FaceFormer
# steps, always in layernorm_bwd_kernel (per-lane register dumps; 9 / 60 step pairs differ with the form, 0 / 70 with those four subtractions as scalar
# v_sub_f32 in an otherwise identical library, 0 / 200 + 0 / 100 with this flag).  Without the vectoriser the library holds no such instruction (86 before:
# tools/scan_packed_opsel.py, tests/test_build_cpu.py) and no workload is slower (VOCA 0.97 -> 0.95 ms, FaceFormer 12 x 3000 196 -> 192 ms, the rest equal).
# A2F_BUILD_SLP=1 builds the vulnerable form again (the reproducer of DESIGN 2.4 needs it; never for a product build).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
if os.environ.get("A2F_BUILD_SLP") != "1":
    FLAGS.append("-fno-slp-vectorize")

def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build liba2f_hip.so)")


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(PKG_DIR, "..", "include", "a2f_hip.h"))
    return [h for h in hs if os.path.exists(h)]


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in _sources()] + _headers()
    return any(os.path.getmtime(d) > t for d in deps)


def _compile_one(src: str, obj: str, verbose: bool):
    tmp = f"{obj}.tmp.{os.getpid()}"                       # unique per process: a concurrent builder never writes the same file
    cmd = [_hipcc()] + FLAGS + ["-c", src, "-o", tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {os.path.basename(src)}:\n" + r.stdout + r.stderr)
    os.replace(tmp, obj)
    return r.stderr


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Incremental build under an exclusive file lock: under torchrun every rank that finds a stale library would otherwise compile the
    same objects at once; the ranks that waited for the lock find the library fresh and return."""
    if not force and not needs_build():
        return LIB_PATH
    import fcntl
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(os.path.join(OBJ_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():            # another process built it while this one waited
                return LIB_PATH
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _deps_mtime(src: str) -> float:
    """Newest of the source and the headers it includes (transitively, quoted includes only; every source reaches include/a2f_hip.h
    through a2f_common.h) - a change to gemm_mfma.h rebuilds the two GEMM files, not all eighteen."""
    import re
    seen, todo, newest = set(), [src], 0.0
    while todo:
        f = os.path.normpath(todo.pop())
        if f in seen or not os.path.exists(f):
            continue
        seen.add(f)
        newest = max(newest, os.path.getmtime(f))
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(f).read(), flags=re.M):
            todo.append(os.path.join(os.path.dirname(f), inc))
    return newest


def _build_locked(force: bool, verbose: bool) -> str:
    jobs, objs = [], []
    for s in _sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ_DIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < _deps_mtime(src):
            jobs.append((src, obj))
    nj = max(1, int(os.environ.get("A2F_BUILD_JOBS", str(min(8, os.cpu_count() or 1)))))
    with ThreadPoolExecutor(max_workers=nj) as ex:
        for warn in ex.map(lambda so: _compile_one(so[0], so[1], verbose), jobs):
            if verbose and warn:
                print(warn, file=sys.stderr)
    tmp_lib = f"{LIB_PATH}.tmp.{os.getpid()}"
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_lib] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + r.stdout + r.stderr)
    os.replace(tmp_lib, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
