"""Builds libransac_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Two variants of the same sources:
  product  libransac_hip.so        what bench.py, smoke() and every product-path test load.  Reads NO environment
                                   variable; the tuning knobs a caller may legitimately set go through rh_set_option.
  diag     libransac_hip_diag.so   -DRH_DIAG: the A/B switches of the rounds' experiments (RH_NO_PIPELINE, RH_HOST_SAMPLER, ...),
                                   the skeleton-only launch (RH_G2_DBG), the fake-RCCL hook (RH_RCCL_LIB) and the rh_dbg_*
                                   audit entry points.  Tests marked `diag`, the fuzzers and the profiling tools load it
                                   (RH_LIB_VARIANT=diag, _lib.py).
Every translation unit is compiled on its own (in parallel) into _build/<variant>/ and re-used while its inputs --
the source, every header, the flags -- are unchanged; the link step follows.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libransac_hip.so")
SO_DIAG = os.path.join(HERE, "libransac_hip_diag.so")
OBJDIR = os.path.join(HERE, "_build")
SOURCES = ["kernels.hip", "score4.hip", "f32.hip", "cloud.hip", "kdorder.hip", "korder.hip", "driver.hip", "driver_extract.hip", "driver_windows.hip", "driver_store.hip", "mp.hip", "sampler.hip", "lsq.hip", "cc.hip", "octree.hip", "comm.hip", "options.cpp", "fit.cpp"]
# -ffp-contract=off: never fuse a*b+c -- inlier sets must match the CPU path bit for bit.
# -fno-slp-vectorize: the vectoriser pairs binary32 operations into v_pk_fma_f32 / v_pk_mul_f32 -- no faster than two
# plain ones on gfx950 (4.2 against 2 x 2.3 issue cycles, profiles/r3/ubench_valu_rates.txt) and every pair pays v_mov
# shuffles to line its operands up: 136 packed operations in the score kernel, cfg3 0.0929 -> 0.0890 ms, cfg5 0.388 -> 0.368.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]
VARIANTS = {"product": (SO, []), "diag": (SO_DIAG, ["-DRH_DIAG=1"])}


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


STAMP = SO + ".stamp"


def _headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(ROOT, "include", "ransac_hip.h")]


def _digest(paths, extra):
    h = hashlib.sha256()
    for d in paths:
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(extra.encode())
    return h.hexdigest()


def source_hash(variant="product"):
    """sha256 over the sources, the public header, the flags and this file: what the library was built from
    (modification times do not survive a copy of the tree, e.g. onto a GPU box)."""
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                  if f.endswith((".hip", ".h", ".cpp"))) + [os.path.join(ROOT, "include", "ransac_hip.h"), __file__]
    return _digest(deps, " ".join(FLAGS + VARIANTS[variant][1] + SOURCES + os.environ.get("RH_EXTRA_FLAGS", "").split()))


def _stamp(variant):
    return VARIANTS[variant][0] + ".stamp"


def needs_build(variant="product"):
    so = VARIANTS[variant][0]
    if not os.path.exists(so) or not os.path.exists(_stamp(variant)):
        return True
    with open(_stamp(variant)) as f:
        return f.read().strip() != source_hash(variant)


def _compile_one(args):
    src, obj, cmd, key, verbose = args
    keyf = obj + ".key"
    if os.path.exists(obj) and os.path.exists(keyf):
        with open(keyf) as f:
            if f.read().strip() == key:
                return src, False
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(keyf, "w") as f:
        f.write(key + "\n")
    return src, True


def build(force=False, verbose=False, variant="product", out=None, extra_flags=None, jobs=None):
    """Compile + link one variant.  `out` / `extra_flags`: an experiment's library next to the product one (its objects
    are keyed by their flags, so variants share nothing they should not)."""
    so, vflags = VARIANTS[variant]
    custom = out is not None or extra_flags
    if out is not None:
        so = out
    if not force and not custom and not needs_build(variant):
        return so
    extra = list(extra_flags or []) + os.environ.get("RH_EXTRA_FLAGS", "").split()
    flags = FLAGS + vflags + extra
    tag = variant if not extra else variant + "_" + hashlib.sha256(" ".join(extra).encode()).hexdigest()[:10]
    odir = os.path.join(OBJDIR, tag)
    os.makedirs(odir, exist_ok=True)
    hdrs = _headers()
    cc = hipcc()
    work = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(odir, s + ".o")
        key = _digest([src] + hdrs, " ".join(flags))
        cmd = [cc] + flags + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-x", "hip", "-c", src, "-o", obj]
        work.append((src, obj, cmd, "" if force else key, verbose))
    jobs = jobs or int(os.environ.get("RH_BUILD_JOBS", "0")) or min(8, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        for src, did in ex.map(_compile_one, work):
            if verbose and did:
                print("compiled", os.path.basename(src), flush=True)
    if force:   # (objects compiled with an empty key are re-keyed so that the next call can re-use them)
        for (src, obj, cmd, _k, _v) in work:
            with open(obj + ".key", "w") as f:
                f.write(_digest([src] + hdrs, " ".join(flags)) + "\n")
    link = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [w[1] for w in work] + ["-ldl", "-lpthread", "-o", so + ".tmp"]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    os.replace(so + ".tmp", so)
    if not custom:
        with open(_stamp(variant), "w") as f:
            f.write(source_hash(variant) + "\n")
    return so


def build_all(force=False, verbose=False):
    return [build(force=force, verbose=verbose, variant=v) for v in VARIANTS]


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if a in VARIANTS] or list(VARIANTS)
    for v in which:
        print(build(force="--force" in sys.argv, verbose=True, variant=v))
