"""Build libhypatia_hip.so for gfx950 with hipcc (in-tree; the .so travels to the GPU box with the
repository snapshot).  Used by __graft_entry__.build() and importable on its own:
    python hypatia.jl_amd/_build.py
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libhypatia_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = (["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result"]
         + os.environ.get("HYP_HIPCC_FLAGS", "").split())   # (extra -D switches for tuning experiments)


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "hypatia_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hm = _headers_mtime()
    jobs = []
    objs = []
    for src in _sources():
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src[:-4] + ".o")
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hm):
            jobs.append((sp, op))

    def compile_one(job):
        sp, op = job
        cmd = [HIPCC] + FLAGS + ["-c", sp, "-o", op]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (sp, r.stderr[-4000:]))
        return sp

    if jobs:
        if verbose:
            print("[hypatia.jl_amd] compiling %d HIP source(s) for gfx950 ..." % len(jobs), flush=True)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    need_link = bool(jobs) or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)
    if need_link:
        # RCCL is the one library dependency besides the HIP runtime: the exchange step of the multi-GPU path (rccl_comm.hip)
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
        if verbose:
            print("[hypatia.jl_amd] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
