"""Builds libomnipq_pointops.so (hand-written HIP for gfx950) in-tree with hipcc.

    python omni-pq_amd/build.py [--force]

No torch, no hipify, no cmake: one hipcc invocation per translation unit, one link.
hipcc cross-compiles gfx950 code objects on a machine without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libomnipq_pointops.so")
OBJDIR = os.path.join(HERE, "build")

# -ffp-contract=off: the index-producing kernels spell out every fma themselves (numerics
# contract in include/omnipq_pointops.h); nothing else may fuse.
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
          "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function",
          "-I", os.path.join(REPO, "include"), "-I", CSRC]

# per-file extra flags.  fps.hip: the SLP vectorizer packs the per-point f32 arithmetic into
# v_pk_* pairs, which doubles the live registers of the 8-points-per-thread variants (134 spilled
# VGPRs); scalar code needs 79 and runs the same instruction count.
EXTRA = {"fps.hip": ["-fno-slp-vectorize"]}


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    inc = os.path.join(REPO, "include")
    headers += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(OBJDIR, src[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [os.path.join(CSRC, src)] + headers):
            cmd = [hipcc()] + COMMON + EXTRA.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
