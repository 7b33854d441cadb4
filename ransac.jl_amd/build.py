"""Builds libransac_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libransac_hip.so")
SOURCES = ["kernels.hip", "score4.hip", "f32.hip", "cloud.hip", "kdorder.hip", "korder.hip", "driver.hip", "driver_extract.hip", "driver_windows.hip", "driver_store.hip", "mp.hip", "sampler.hip", "lsq.hip", "cc.hip", "octree.hip", "comm.hip", "fit.cpp"]
# -ffp-contract=off: never fuse a*b+c -- inlier sets must match the CPU path bit for bit.
# -fno-slp-vectorize: the vectoriser pairs binary32 operations into v_pk_fma_f32 / v_pk_mul_f32 -- no faster than two
# plain ones on gfx950 (4.2 against 2 x 2.3 issue cycles, profiles/r3/ubench_valu_rates.txt) and every pair pays v_mov
# shuffles to line its operands up: 136 packed operations in the score kernel, cfg3 0.0929 -> 0.0890 ms, cfg5 0.388 -> 0.368.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


STAMP = SO + ".stamp"


def source_hash():
    """sha256 over the sources, the public header, the flags and this file: what the library was built from
    (modification times do not survive a copy of the tree, e.g. onto a GPU box)."""
    import hashlib
    h = hashlib.sha256()
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                  if f.endswith((".hip", ".h", ".cpp"))) + [os.path.join(ROOT, "include", "ransac_hip.h"), __file__]
    for d in deps:
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS + SOURCES).encode())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(SO) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    cmd = [hipcc()] + FLAGS + os.environ.get("RH_EXTRA_FLAGS", "").split() + ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
    for s in SOURCES:
        cmd += ["-x", "hip", os.path.join(CSRC, s)]
    cmd += ["-ldl", "-o", SO]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(source_hash() + "\n")
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(SO)
