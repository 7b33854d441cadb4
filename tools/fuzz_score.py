#!/usr/bin/env python3
"""Randomised parity of the batched score (counts + masks) against the oracle: cloud sizes around the
tile / group boundaries, random enabled patterns, thresholds from tiny to huge, candidates from the
ground truth (jittered) to arbitrary / degenerate ones, both score paths.  python tools/fuzz_score.py [n] [seed]"""
import os, sys, time
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ransac_jl_amd as R
from ransac_jl_amd import _lib as L, synth
from oracle import oracle as orc

KMAP = {"plane": 0, "sphere": 1, "cylinder": 2, "cone": 3}


def rand_shape(rng, truth, scale):
    mode = rng.integers(0, 10)
    kind = int(rng.integers(0, 4))
    s = L.Shape()
    s.kind = kind
    s.outwards = int(rng.integers(0, 2))
    if mode < 5 and truth:     # near a real primitive
        t = truth[int(rng.integers(0, len(truth)))]
        jit = float(rng.choice([0.0, 0.001, 0.01, 0.1]))
        j = lambda x: np.asarray(x, dtype=np.float64) * (1 + jit * rng.uniform(-1, 1, size=np.shape(x)))
        k = t["kind"]
        s.kind = KMAP[k]
        if k == "plane":
            pt, nv = j(t["point"]), j(t["normal"])
            if rng.integers(0, 4) == 0:   # the same plane given by a point far away (inside the plane): oz . p0 cancels, |p0| does not
                t1 = np.cross(nv, rng.normal(size=3)); t1 /= max(np.linalg.norm(t1), 1e-30)
                t2 = np.cross(nv / max(np.linalg.norm(nv), 1e-30), t1)
                pt = pt + t1 * 10.0 ** rng.uniform(2, 6.5) * rng.choice([-1, 1]) + t2 * 10.0 ** rng.uniform(2, 6.5) * rng.choice([-1, 1])
            v = list(pt) + list(nv)
        elif k == "sphere": v = list(j(t["center"])) + [float(j(t["radius"]))]
        elif k == "cylinder": v = list(j(t["axis"])) + list(j(t["center"])) + [float(j(t["radius"]))]
        else: v = list(j(t["apex"])) + list(j(t["axis"])) + [float(j(t["opang"]))]
    else:                      # arbitrary
        c = rng.uniform(-0.2, 1.2, 3) * scale
        a = rng.normal(size=3) * float(rng.choice([1.0, 1.0, 0.3, 3.0]))
        r = float(rng.choice([0.0, 1e-3, 1.0, 10.0, 1e3])) * (scale / 100) * rng.uniform(0.5, 1.5)
        if kind == 0: v = list(c) + list(a)
        elif kind == 1: v = list(c) + [r]
        elif kind == 2: v = list(a) + list(c) + [r]
        else: v = list(c) + list(a) + [float(rng.uniform(0.02, 3.0))]
        if mode == 9:          # degenerate: NaN / inf / zero axis somewhere
            v[int(rng.integers(0, len(v)))] = float(rng.choice([np.nan, np.inf, -np.inf, 0.0]))
    for i, x in enumerate(v):
        s.v[i] = float(x)
    R.lib().rh_shape_finalize(C.byref(s))
    return s


def one(case, rng, f32=False):
    scale = float(rng.choice([1.0, 100.0, 100.0, 1e4]))
    n = int(rng.choice([700, 4096, 8192, 8193, 20_000, 65_536, 150_001]))
    names = list(rng.choice(list(KMAP), size=int(rng.integers(1, 6))))
    xyz, nrm, truth = synth.make_cloud(n, names, float(rng.choice([0.0, 0.2, 0.5])), seed=5000 + case)
    xyz = xyz * (scale / 100.0)
    for t in truth:
        for k in ("point", "center", "apex"):
            if k in t: t[k] = np.asarray(t[k]) * (scale / 100.0)
        if "radius" in t: t["radius"] = t["radius"] * (scale / 100.0)
    r = int(rng.choice([1, 2, 3, 16]))
    subs = synth.make_subsets(n, r, seed=case)
    spath = str(rng.choice(["groups", "groups", "brute"]))
    R.set_option("score_path", spath)               # rh_set_option: product and diag build alike
    rr = int(rng.choice([0, 0, 4, 8, 12, 16]))      # rows of the v4 kernel: the library's own choice, or every instantiation
    R.set_option("s4_rows", rr if rr else None)
    stc = int(rng.choice([0, 1, 1, 2]))             # super-tile lists: by size (never, at these sizes), whenever possible, never
    R.set_option("st_cull", stc if stc else None)
    if f32:   # a Float32 cloud: binary32 arithmetic on both sides (oracle/orc_f32.c)
        xyz, nrm = xyz.astype(np.float32), nrm.astype(np.float32)
        pc = R.RANSACCloud(xyz, nrm, subs, force_eltype=np.float32)
        oc = orc.Cloud32(xyz, nrm, subs[0])
    else:
        pc = R.RANSACCloud(xyz, nrm, subs)
        oc = orc.Cloud(xyz, nrm, subs[0])
    en = rng.random(n) < float(rng.choice([1.0, 0.9, 0.3, 0.01]))
    pc.set_enabled(en)
    bits = np.zeros(((n + 63) // 64) * 64, dtype=np.uint8); bits[:n] = en
    oc.set_enabled(np.packbits(bits, bitorder="little").view(np.uint64))
    params = R.ransacparameters()
    for k in ("plane", "sphere", "cylinder", "cone"):
        params[k]["ϵ"] = float(rng.choice([1e-4, 0.05, 0.3, 2.0, 50.0])) * (scale / 100.0)
        params[k]["α"] = float(np.radians(rng.choice([0.5, 5.0, 30.0, 89.0, 120.0])))
    cp = R.params_to_c(params, score_mode=L.SCORE_F64, sphere_uses_enabled=bool(rng.integers(0, 2)))
    b = int(rng.choice([1, 7, 64, 65, 300]))
    arr = (L.Shape * b)(*[rand_shape(rng, truth, scale) for _ in range(b)])
    if f32:
        for i in range(b):
            R.lib().rh_shape_finalize_f32(C.byref(arr[i]))
    want_masks = bool(rng.integers(0, 2))
    got = R.score_batch(pc, arr, cp, want_masks=want_masks)
    oarr = (orc.Shape * b)()
    C.memmove(oarr, arr, C.sizeof(L.Shape) * b)
    exp = oc.score_batch(oarr, orc.Params.from_buffer_copy(bytes(cp)), want_masks=want_masks)
    if want_masks:
        ok = np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1])
        tot = int(np.sum(exp[0]))
    else:
        ok = np.array_equal(got, exp)
        tot = int(np.sum(exp))
    # the same batch through the device-buffer entry with F batches in flight (rh_set_option batches_in_flight): slices of the
    # batch, buffer k mod F for call k, no wait until the end -- every slice's counts (and masks) as the oracle has them
    F = int(rng.choice([1, 1, 2, 3, 4]))
    if F > 1 and ok:
        lib = R.lib()
        exp_c, exp_m = (exp if want_masks else (exp, None))
        w = (subs[0].size + 63) // 64
        d_sh = C.c_void_p()
        L.check(lib.rh_dev_alloc(pc._h, C.sizeof(L.Shape) * b, C.byref(d_sh)))
        L.check(lib.rh_dev_upload(pc._h, d_sh, C.cast(arr, C.c_void_p), C.sizeof(L.Shape) * b))
        d_cn, d_mk = [C.c_void_p() for _ in range(F)], [C.c_void_p() for _ in range(F)]
        for q in range(F):
            L.check(lib.rh_dev_alloc(pc._h, 4 * b, C.byref(d_cn[q])))
            if want_masks:
                L.check(lib.rh_dev_alloc(pc._h, 8 * max(1, w) * b, C.byref(d_mk[q])))
        R.set_option("batches_in_flight", F, cloud=pc)
        ncall = int(rng.integers(F, 3 * F + 1))
        sl = []
        for k in range(ncall):
            lo = int(rng.integers(0, b)); hi = int(rng.integers(lo + 1, b + 1))
            sl.append((lo, hi))
            L.check(lib.rh_score_batch_dev(pc._h, C.c_void_p(d_sh.value + lo * C.sizeof(L.Shape)), hi - lo, C.byref(cp), d_cn[k % F],
                                           d_mk[k % F] if want_masks else None))
        L.check(lib.rh_cloud_sync(pc._h))
        for k in range(ncall - F, ncall):
            lo, hi = sl[k]
            cn = np.zeros(hi - lo, dtype=np.int32)
            L.check(lib.rh_dev_download(pc._h, cn.ctypes.data_as(C.c_void_p), d_cn[k % F], 4 * (hi - lo)))
            ok = ok and np.array_equal(cn, exp_c[lo:hi])
            if want_masks and w > 0:
                mk = np.zeros((hi - lo, w), dtype=np.uint64)
                L.check(lib.rh_dev_download(pc._h, mk.ctypes.data_as(C.c_void_p), d_mk[k % F], 8 * w * (hi - lo)))
                ok = ok and np.array_equal(mk, exp_m[lo:hi])
        R.set_option("batches_in_flight", None, cloud=pc)
        for q in range(F):
            lib.rh_dev_free(pc._h, d_cn[q])
            if want_masks:
                lib.rh_dev_free(pc._h, d_mk[q])
        lib.rh_dev_free(pc._h, d_sh)
    return ok, "n=%d r=%d scale=%g path=%s R=%d lists=%d b=%d masks=%d in_flight=%d inliers=%d" % (n, r, scale, spath, rr, stc, b, want_masks, F, tot)


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
    bad, t0 = 0, time.time()
    for case in range(ncases):
        ok, desc = one(case, rng, f32=bool(os.environ.get("F32")))   # F32=1: Float32 clouds against the oracle's binary32 twin
        print("%s case %3d  %s" % ("ok  " if ok else "FAIL", case, desc), flush=True)
        bad += not ok
    print("%d cases, %d failures, %.0f s" % (ncases, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
