#!/usr/bin/env python3
"""Randomised parity of the refit scan: rh_refit / rh_invalidate against the oracle on random clouds (sizes around the
word / tile boundaries, coordinate scales 1 .. 1e4, NaN / inf points, random enabled patterns), thresholds from tiny to
huge, shapes from jittered ground truth to degenerate -- with the culled scan (korder.hip: Morton order, box tests)
forced on clouds of every size, or the plain scan, and on Float32 clouds.  Run on a GPU box:
python tools/fuzz_refit.py [n_cases] [seed]; a fixed-seed slice runs in tests/test_fuzz_gpu.py."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ransac_jl_amd as R
from ransac_jl_amd import _lib as L, synth
from oracle import oracle as orc
from fuzz_score import KMAP, rand_shape


def one(case, rng, f32=False):
    scale = float(rng.choice([1.0, 100.0, 100.0, 1e4]))
    n = int(rng.choice([1, 63, 64, 65, 700, 4096, 8193, 20_000, 65_536, 150_001]))
    names = list(rng.choice(list(KMAP), size=int(rng.integers(1, 6))))
    xyz, nrm, truth = synth.make_cloud(n, names, float(rng.choice([0.0, 0.2, 0.5])), seed=7000 + case)
    xyz = xyz * (scale / 100.0)
    for t in truth:
        for k in ("point", "center", "apex"):
            if k in t: t[k] = np.asarray(t[k]) * (scale / 100.0)
        if "radius" in t: t["radius"] = t["radius"] * (scale / 100.0)
    bad_pts = int(rng.choice([0, 0, 1, 5]))
    for _ in range(min(bad_pts, n)):          # points no shape can hold: they must neither match nor hide a group
        xyz[int(rng.integers(0, n)), int(rng.integers(0, 3))] = float(rng.choice([np.nan, np.inf, -np.inf]))
    subs = synth.make_subsets(n, int(rng.choice([1, 2, 16])), seed=case) if n >= 16 else [np.arange(1, n + 1)]
    path = str(rng.choice(["culled", "culled", "culled", "scan"]))
    R.set_option("refit_path", path)          # rh_set_option: product and diag build alike
    if f32:
        xyz, nrm = xyz.astype(np.float32), nrm.astype(np.float32)
        pc = R.RANSACCloud(xyz, nrm, subs, force_eltype=np.float32)
        oc = orc.Cloud32(xyz, nrm, subs[0])
    else:
        pc = R.RANSACCloud(xyz, nrm, subs)
        oc = orc.Cloud(xyz, nrm, subs[0])
    p_en = float(rng.choice([1.0, 1.0, 0.9, 0.3, 0.01]))
    if p_en < 1.0:
        en = rng.random(n) < p_en
        pc.set_enabled(en)
        bits = np.zeros(((n + 63) // 64) * 64, dtype=np.uint8); bits[:n] = en
        oc.set_enabled(np.packbits(bits, bitorder="little").view(np.uint64))
    params = R.ransacparameters()
    for k in ("plane", "sphere", "cylinder", "cone"):
        params[k]["ϵ"] = float(rng.choice([1e-4, 0.05, 0.3, 2.0, 50.0])) * (scale / 100.0)
        params[k]["α"] = float(np.radians(rng.choice([0.5, 5.0, 30.0, 89.0, 120.0])))
    cp = R.params_to_c(params)
    op = orc.Params.from_buffer_copy(bytes(cp))
    ok, tot = True, 0
    for q in range(6):
        s = rand_shape(rng, truth, scale)
        if f32:
            R.lib().rh_shape_finalize_f32(C.byref(s))
        got = R.refit(s, pc, cp).inpoints
        exp = oc.refit(orc.Shape.from_buffer_copy(bytes(s)), op)
        ok = ok and np.array_equal(got, exp)
        tot += len(exp)
        if len(exp) and rng.integers(0, 2):   # invalidate_indexes!: both bit sets of the cloud follow
            R.invalidate_indexes(pc, exp)
            oc.invalidate(exp)
    ok = ok and np.array_equal(pc.enabled_chunks(), oc.get_enabled())
    return ok, "n=%d scale=%g path=%s f32=%d bad=%d p_en=%g inliers=%d" % (n, scale, path, f32, bad_pts, p_en, tot)


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(ncases):
        for f32 in (False, True):
            ok, desc = one(case, rng, f32=f32)
            if not ok:
                bad += 1
                print("MISMATCH", case, desc, flush=True)
        if case % 20 == 19:
            print("case", case + 1, "bad", bad, flush=True)
    print("cases", 2 * ncases, "failures", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
