"""Builds the hand-written HIP library for gfx950 in-tree with hipcc -- twice, once per 16-bit element type:

    lib/libomnipq_pointops.so        e16 = bfloat16   (torch.autocast(bfloat16); also every index / f32 operator)
    lib/libomnipq_pointops_f16.so    e16 = IEEE half  (torch.autocast(float16); -DOMNIPQ_ELEM_F16, same sources, same entry points)

    python omni-pq_amd/build.py [--force]

No torch, no hipify, no cmake: one hipcc invocation per translation unit and element type, one link each.
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
LIB_F16 = os.path.join(LIBDIR, "libomnipq_pointops_f16.so")
OBJDIR = os.path.join(HERE, "build")
# (library, object sub-directory, extra defines)
VARIANTS = [(LIB, "bf16", []), (LIB_F16, "f16", ["-DOMNIPQ_ELEM_F16=1"])]

# -ffp-contract=off: the index-producing kernels spell out every fma themselves (numerics
# contract in include/omnipq_pointops.h); nothing else may fuse.
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
          "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function",
          "-I", os.path.join(REPO, "include"), "-I", CSRC]

# per-file extra flags.  fps.hip: the SLP vectorizer packs the per-point f32 arithmetic into
# v_pk_* pairs, which doubles the live registers of the 8-points-per-thread variants (134 spilled
# VGPRs); scalar code needs 79 and runs the same instruction count.
# attention.hip: -amdgpu-mfma-vgpr-form keeps the MFMA accumulators in the architectural registers.  The compiler's default
# put the output accumulators (which the online-softmax rescale also touches with VALU instructions) into AGPRs and copied them
# out and back around the matrix instructions of EVERY key block: 64 v_accvgpr_read / _write + 26 v_mov of the ~570 VALU
# instructions of a forward iteration.  With the flag: no copies, 160 / 184 / 204 registers instead of 192 / 212 / 236
# (forward: three waves per SIMD instead of two).
EXTRA = {"fps.hip": ["-fno-slp-vectorize"], "attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def sources_digest():
    """sha1 over the kernel sources (csrc/*.hip, csrc/*.h, include/*.h): stamped into the counter summaries under profiles/
    by tools/pmc_*.py and compared by bench.py, which labels a committed counter figure `stale` when the kernels have changed
    since it was taken."""
    import hashlib
    h = hashlib.sha1()
    inc = os.path.join(os.path.dirname(HERE), "include")
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    files += [os.path.join(inc, f) for f in sorted(os.listdir(inc)) if f.endswith(".h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def plan_aware_entry_points():
    """Names of the entry points whose declaration in include/omnipq_sa.h takes `const omnipq_row_plan *plan` (in front of
    the stream).  Evaluated at BUILD time and compiled into the library (csrc/capi.hip: omnipq_plan_aware_entry_points), so
    that a binding asks the library it actually loaded how its entry points are called instead of parsing a header that may
    belong to another build."""
    import re
    with open(os.path.join(REPO, "include", "omnipq_sa.h")) as fh:
        text = re.sub(r"/\*.*?\*/", " ", fh.read(), flags=re.S)
    return sorted(m.group(1) for m in re.finditer(r"\b(omnipq_\w+)\s*\(([^;{}()]*)\)\s*;", text)
                  if "omnipq_row_plan *plan" in m.group(2))


def _write_generated():
    """build/generated/plan_aware.inc: one string literal, rewritten only when its content changes."""
    gen = os.path.join(OBJDIR, "generated")
    os.makedirs(gen, exist_ok=True)
    path = os.path.join(gen, "plan_aware.inc")
    text = '"' + " ".join(plan_aware_entry_points()) + '"\n'
    if not os.path.exists(path) or open(path).read() != text:
        with open(path, "w") as fh:
            fh.write(text)
    return gen


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """-> path of the bf16 library (the f16 twin is built next to it: LIB_F16)."""
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    inc = os.path.join(REPO, "include")
    headers += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    gen = _write_generated()
    headers.append(os.path.join(gen, "plan_aware.inc"))
    procs, links = [], []
    for lib, sub, defines in VARIANTS:
        objdir = os.path.join(OBJDIR, sub)
        os.makedirs(objdir, exist_ok=True)
        objs = []
        compiled = False
        for src in sources():
            obj = os.path.join(objdir, src[:-4] + ".o")
            objs.append(obj)
            if force or _stale(obj, [os.path.join(CSRC, src)] + headers):
                cmd = [hipcc()] + COMMON + ["-I", gen] + defines + EXTRA.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
                if verbose:
                    print(" ".join(cmd))
                procs.append((src, subprocess.Popen(cmd)))
                compiled = True
        links.append((lib, objs, compiled))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    for lib, objs, compiled in links:
        if force or compiled or _stale(lib, objs):
            cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
