#!/usr/bin/env python3
"""Kernel A/B inside ONE process: several builds of the product library (each its own dlopen handle, its own cloud over the
same host arrays) score the same batch in alternating bursts, so that clock / thermal drift and box-to-box differences
hit them alike.  Prints the median and the spread of the per-burst step time (HIP events on each cloud's stream) and checks
that every build returns the same counts.
    python tools/ab_inproc.py variants/base.so ransac.jl_amd/libransac_hip.so        WL=cfg2|cfg3|cfg5  ROUNDS=12  STEPS=50
    MASKS=1: the step with the inlier masks written;  OPTS="s4_rows=8,..." applied to every build (rh_set_option);
    lib.so@batches_in_flight=2@...: options of that entry's cloud alone (the same library may be listed more than once)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ransac_jl_amd as R
from ransac_jl_amd import _lib as L, synth
import bench


def load(path):
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in L.SIGNATURES.items():
        if hasattr(lib, name):
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
    return lib


def main():
    paths = [a for a in sys.argv[1:] if ".so" in a]
    wl = os.environ.get("WL", "cfg3")
    rounds, steps = int(os.environ.get("ROUNDS", "12")), int(os.environ.get("STEPS", "50"))
    masks = bool(os.environ.get("MASKS"))
    prim = ["plane"] * 16 + ["sphere"] * 12 + ["cylinder"] * 12
    types = [R.FittedPlane, R.FittedSphere, R.FittedCylinder]
    n, seed, scanner, outl = 10_000_000, 3, None, 0.30
    if wl == "cfg2":
        prim = ["plane", "plane", "sphere", "sphere", "cylinder", "cylinder"]; n, seed, outl = 1_000_000, 2, 0.0
    if wl == "cfg5":
        prim += ["cone"] * 8; types += [R.FittedCone]; n, seed, scanner = 50_000_000, 5, [synth.BOX / 2] * 3
    n = int(os.environ.get("POINTS", n))        # POINTS=1000000: the workload's mix on a cloud of another size
    xyz, nrm, truth = synth.make_cloud(n, prim, outl, seed=seed, scanner=scanner)
    subs = synth.make_subsets(n, 32, seed=seed)
    sub1 = np.ascontiguousarray(subs[0], dtype=np.int64)
    cp = R.params_to_c(R.ransacparameters(types), score_mode=L.SCORE_F64)
    cands = synth.jittered_candidates(truth, 4096, seed=0)
    arr = bench.shapes_to_c(R, L, cands)
    dp, i64p = C.POINTER(C.c_double), C.POINTER(C.c_int64)
    S = sub1.size
    swords = (S + 63) // 64
    builds = []
    for p in paths:
        p, *own = p.split("@")
        lib = load(p)
        for kv in filter(None, os.environ.get("OPTS", "").split(",")):
            k, v = kv.split("=")
            assert lib.rh_set_option(None, k.encode(), int(v)) == 0, kv
        h = C.c_void_p()
        rc = lib.rh_cloud_create(xyz.ctypes.data_as(dp), nrm.ctypes.data_as(dp), n, sub1.ctypes.data_as(i64p), S, 0, C.byref(h))
        assert rc == 0, (p, lib.rh_last_error())
        for kv in own:
            k, v = kv.split("=")
            assert lib.rh_set_option(h, k.encode(), int(v)) == 0, kv
        d_sh, d_cn, d_mk = C.c_void_p(), C.c_void_p(), C.c_void_p()
        d_cnx = [C.c_void_p() for _ in range(3)]
        for d in d_cnx:
            assert lib.rh_dev_alloc(h, 4 * 4096, C.byref(d)) == 0
        assert lib.rh_dev_alloc(h, C.sizeof(L.Shape) * 4096, C.byref(d_sh)) == 0
        assert lib.rh_dev_alloc(h, 4 * 4096, C.byref(d_cn)) == 0
        d_mks = [C.c_void_p() for _ in range(4)]
        if masks:
            for d in d_mks:
                assert lib.rh_dev_alloc(h, 8 * swords * 4096, C.byref(d)) == 0
        assert lib.rh_dev_upload(h, d_sh, C.cast(arr, C.c_void_p), C.sizeof(L.Shape) * 4096) == 0
        builds.append(dict(path="@".join([p] + own), lib=lib, h=h, d_sh=d_sh, nbuf=max([int(kv.split("=")[1]) for kv in own if kv.startswith("batches_in_flight=")] + [2]), d_cn=d_cn, d_cn1=d_cnx[0], d_cn2=d_cnx[1], d_cn3=d_cnx[2], d_mks=d_mks, ms=[]))

    def burst(b, k):
        lib = b["lib"]
        assert lib.rh_timer_start(b["h"]) == 0
        for i in range(k):   # (F count buffers in turn: what a caller with F batches in flight does)
            rc = lib.rh_score_batch_dev(b["h"], b["d_sh"], 4096, C.byref(cp), b[("d_cn", "d_cn1", "d_cn2", "d_cn3")[i % b["nbuf"]]], b["d_mks"][i % b["nbuf"]] if masks else None)
            assert rc == 0, lib.rh_last_error()
        ms = C.c_float()
        assert lib.rh_timer_stop(b["h"], C.byref(ms)) == 0
        return ms.value / k
    for b in builds:       # warm-up: clocks, lazy allocations
        burst(b, 200)
    ref = None
    for b in builds:
        for key in ("d_cn", "d_cn1", "d_cn2", "d_cn3")[:b["nbuf"]]:
            cn = np.zeros(4096, dtype=np.int32)
            assert b["lib"].rh_dev_download(b["h"], cn.ctypes.data_as(C.c_void_p), b[key], 4 * 4096) == 0
            if ref is None:
                ref = cn
            assert np.array_equal(cn, ref), "counts differ between builds: %s" % b["path"]
    for r in range(rounds):
        order = builds if r % 2 == 0 else builds[::-1]
        for b in order:
            b["ms"].append(burst(b, steps))
    base = float(np.median(builds[0]["ms"]))
    for b in builds:
        m = np.asarray(b["ms"])
        print("%s %-40s median %.4f ms  min %.4f  max %.4f  (%+.2f%% vs first)  sum(counts) %d"
              % (wl + (" masks" if masks else ""), os.path.basename(b["path"]), np.median(m), m.min(), m.max(), 100 * (np.median(m) / base - 1), int(ref.sum())), flush=True)


if __name__ == "__main__":
    main()
