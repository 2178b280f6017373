"""Build libsaunet_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m saunet_amd._build        (or __graft_entry__.build())

One object per .hip translation unit, rebuilt only when the source or a header is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libsaunet_hip.so")
SOURCES = ["conv.hip", "conv_igemm.hip", "conv_tile.hip", "conv_mm.hip", "dense_dgrad.hip", "dense_fwd.hip", "norm.hip", "pointwise.hip", "pool.hip", "gate.hip", "expand.hip", "loss.hip", "canny.hip", "augment.hip", "optim.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "mma_tiles.h"), os.path.join(ROOT, "include", "saunet_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, timing=False):
    """timing=True: the profiling build with the TSTAMP phase stamps compiled in (common.h) -> scripts/_ab/libsaunet_timing.so;
    select it at run time with SAUNET_HIP_LIB (never used by the product path)."""
    global OBJ, LIB, FLAGS
    if timing:
        OBJ = os.path.join(HERE, "_obj_timing")
        LIB = os.path.join(ROOT, "scripts", "_ab", "libsaunet_timing.so")
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        FLAGS = FLAGS + ["-DSAUNET_TIMING"]
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + HEADERS):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        return obj

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), 8)) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, timing="--timing" in sys.argv))
