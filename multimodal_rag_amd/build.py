"""Build libmmrag.so (HIP, gfx950 only) in-tree with hipcc.

    python -m multimodal_rag_amd.build [--force]

hipcc cross-compiles gfx950 without a GPU.  Objects are cached under
multimodal_rag_amd/csrc/_obj and rebuilt when a source or header is newer.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libmmrag.so")
INCLUDE = os.path.join(os.path.dirname(PKG), "include")

SOURCES = ["common.hip", "search.hip", "search_qs.hip", "search_qsw.hip", "encoder.hip", "encoder_f32.hip", "image.hip", "microbench.hip", "host_merge.cpp", "tokenizer.cpp", "clip_bpe.cpp"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libmmrag.so cannot be built (ROCm toolchain required)")


def _headers_mtime() -> float:
    m = 0.0
    for d in (CSRC, INCLUDE):
        for f in os.listdir(d):
            if f.endswith((".h", ".inc")):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hdr_m = _headers_mtime()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m)
        if stale:
            cmd = [hipcc, "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-I", INCLUDE, "-c", src, "-o", obj]
            cmd[1:1] = os.environ.get("MMRAG_HIPCC_FLAGS", "").split()   # developer builds (e.g. -DMMRAG_QSW_DEV)
            if s.endswith(".cpp"):
                cmd.insert(1, "-x")
                cmd.insert(2, "hip")
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print("[mmrag build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed ({r.returncode}):\n{r.stdout}\n{r.stderr}")

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs)
    build_hostrows(force, run)
    return LIB


def build_hostrows(force: bool, run) -> str:
    """the CPython extension that builds a batch's result lists in C (csrc/hostrows.c; host only, plain C)"""
    import sysconfig

    src = os.path.join(CSRC, "hostrows.c")
    out = os.path.join(LIBDIR, "_hostrows" + sysconfig.get_config_var("EXT_SUFFIX"))
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        cc = shutil.which("gcc") or shutil.which("cc") or "gcc"
        run([cc, "-O2", "-shared", "-fPIC", "-Wall", "-I", sysconfig.get_paths()["include"], src, "-o", out])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
