"""Writes score_statistics_vectors.json: known-answer vectors for the score statistics of the candidate loop --
estimatescore / hypergeomdev (confidenceintervals.jl:53-59, 71-74; paths under /root/reference/src) and prob
(utilities.jl:262) -- and the MARGIN of every extraction decision `prob(E(best), s, N, drawN) > prob_det`
(iterations.jl:114-123) the parity runs take.

The reference's tests hold no vector for these functions and Julia cannot run here, so:

* "exact" estimatescore vectors are hand-picked so that nothing is rounded anywhere: S1 = 2^k - 2 makes
  N' = -2 - S1 = -2^k (every division by N' is exact), and (Plength, sigma) are chosen so that
  x n (N'-x) (N'-n) / (N'-1) is an integer AND a perfect square below 2^53.  This script re-evaluates the formula
  in rational arithmetic (fractions.Fraction) and asserts that every intermediate is a binary64 value.  Both score
  modes (Julia's wrapping Int64 product / the Float64 fix) must return exactly these numbers.  One vector has
  sigma > S1 + 1 (a negative radicand -> the `sq_ < 0 ? 0` branch).
* "wrap" vectors are at the benchmark sizes, where the Int64 product wraps.  The wrapped form is a
  CHAIN of single IEEE operations (Int64 multiply modulo 2^64, one int -> float conversion, one division, one sqrt,
  one add, one division, one subtraction): no evaluation order is open, so Python's integers and correctly
  rounded float operations give the one possible answer, independently of the C code under test.
* "prob" vectors: (n/N)^k and the outer power are dyadic, so every libm and Julia's power_by_squaring agree.
* "decisions": the oracle's ransac() is run with its decision trace on (cfg1: the whole loop; cfg3: the bench's
  768-iteration prefix with all 40 extractions) and for every evaluation of the extraction test the distance
  |ppp - prob_det| is recorded, together with whether ANY combination of +-1 ulp on the two pow() results flips the
  decision (Julia >= 1.8 computes Float64^Int differently from libm in the last ulp).

tests/test_score_statistics_vectors.py checks the oracle and the product (rh_estimatescore, rh_prob: host code,
no GPU needed) against the vectors and re-derives the cfg1 decisions on every CPU run.
Run:  python tests/golden/make_score_statistics_vectors.py [--no-cfg3]
"""
import ctypes as C
import json
import math
import os
import sys
from fractions import Fraction as Fr

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "score_statistics_vectors.json")


def rep(x):
    """assert that the rational x is a binary64 value and return it as a float"""
    x = Fr(x)
    assert x == 0 or Fr(float(x)) == x, "not representable: %s" % x
    return float(x)


def exact_estimatescore(S, P, sigma):
    """the reference's formula in exact rational arithmetic; every intermediate must be a binary64 value"""
    N, x, n = -2 - S, -2 - P, -1 - sigma
    num = x * n * (N - x) * (N - n)
    assert abs(num) < 2 ** 53 and abs(x * n) < 2 ** 53       # no Int64 wrap, exact as Float64 products too
    # Float64 mode multiplies left to right: every partial product is an integer below 2^53 as well
    assert abs(x * n * (N - x)) < 2 ** 53
    q = Fr(num, N - 1)
    if q < 0:
        sq = Fr(0)     # (any negative quotient, rounded or not, takes the `sq_ < 0 ? zero` branch)
    else:
        rep(q)
        assert q.denominator == 1
        r = math.isqrt(q.numerator)
        assert r * r == q.numerator, "not a perfect square"
        sq = Fr(r)
    gmin, gmax = Fr(x * n + sq, N), Fr(x * n - sq, N)
    rep(x * n + sq); rep(x * n - sq); rep(gmin); rep(gmax)
    a, b = -1 - gmin, -1 - gmax
    lo, hi = min(a, b), max(a, b)
    return dict(S1=S, P=P, sigma=sigma, radicand=float(q), min=rep(lo), max=rep(hi), E=rep((lo + hi) / 2))


def wrap_estimatescore(S, P, sigma):
    """Julia's Int arithmetic: products modulo 2^64, then single correctly rounded Float64 operations"""
    def w64(v):
        v &= (1 << 64) - 1
        return v - (1 << 64) if v >= (1 << 63) else v
    N, x, n = -2 - S, -2 - P, -1 - sigma
    xn = w64(x * n)
    prod = w64(w64(xn * (N - x)) * (N - n))
    sq_ = float(prod) / float(N - 1)
    sq = 0.0 if sq_ < 0 else math.sqrt(sq_)
    gmin = (float(xn) + sq) / float(N)
    gmax = (float(xn) - sq) / float(N)
    a, b = -1 - gmin, -1 - gmax
    lo, hi = min(a, b), max(a, b)
    wrapped = (x * n * (N - x) * (N - n)) != prod
    return dict(S1=S, P=P, sigma=sigma, wrapped=bool(wrapped), min=lo, max=hi, E=(lo + hi) / 2)


def exact_prob(n, s, N, k):
    a = Fr(n, N) ** k
    b = (1 - a) ** s
    rep(Fr(n, N)); rep(a); rep(1 - a); rep(b);
    return dict(n=float(n), s=s, N=N, k=k, value=rep(1 - b))


def decisions(name, params_kw, seed, itermax=None):
    from oracle import oracle as orc
    import ransac_jl_amd as R
    from ransac_jl_amd import _lib as L, synth
    c = synth.config(name)
    subs = synth.make_subsets(c["xyz"].shape[0], c["r"], c["seed"])
    if name == "cfg1":
        cp = R.params_to_c(R.ransacparameters([R.FittedPlane, R.FittedSphere]), **params_kw)
    else:
        types = [R.FittedPlane, R.FittedSphere, R.FittedCylinder]
        cp = R.params_to_c(R.ransacparameters(types, iteration={"minsubsetN": 4096, "itermax": itermax, "τ": 900, "prob_det": 0.9}),
                           **params_kw)
    oc = orc.Cloud(c["xyz"], c["nrm"], subs[0])
    cap = 1 << 16
    buf = np.zeros(4 * cap)
    oc.L.orc_trace_set.argtypes = [C.POINTER(C.c_double), C.c_int64]
    oc.L.orc_trace_set.restype = None
    oc.L.orc_trace_count.restype = C.c_int64
    oc.L.orc_trace_set(buf.ctypes.data_as(C.POINTER(C.c_double)), cap)
    res = oc.ransac(orc.Params.from_buffer_copy(bytes(cp)), seed=seed)
    ntr = int(oc.L.orc_trace_count())
    oc.L.orc_trace_set(None, 0)
    assert ntr <= cap
    tr = buf[: 4 * ntr].reshape(-1, 4)
    N, k, pd = c["xyz"].shape[0], int(cp.drawN), float(cp.prob_det)
    rows, flips = [], 0
    for it, scr, s, ppp in tr:
        a = math.pow(scr / N, k)
        base = 1 - a
        b = math.pow(base, s)
        assert 1 - b == ppp
        dec = ppp > pd
        flip = False
        for da in (-1, 0, 1):
            aa = a if da == 0 else math.nextafter(a, math.inf if da > 0 else -math.inf)
            bb0 = math.pow(1 - aa, s)
            for db in (-1, 0, 1):
                bb = bb0 if db == 0 else math.nextafter(bb0, math.inf if db > 0 else -math.inf)
                flip |= ((1 - bb) > pd) != dec
        flips += flip
        rows.append((abs(ppp - pd), int(it), float(scr), int(s), float(ppp), bool(dec)))
    rows.sort()
    ext = sum(1 for r in rows if r[5])
    return dict(workload=name, seed=seed, iterations=int(res["iterations"]), shapes=len(res["shapes"]), decisions=len(rows),
                extractions=ext, prob_det=pd, min_margin=rows[0][0] if rows else None, flips_under_1ulp_pow=int(flips),
                nearest=[dict(margin=r[0], iteration=r[1], E_best=r[2], s=r[3], ppp=r[4], extracted=r[5]) for r in rows[:5]])


def main():
    out = {"doc": __doc__.split("\n\n")[0]}
    picks = [(6, 7, 3), (6, 10, 1), (14, 32, 7), (30, 52, 15), (62, 102, 31), (126, 342, 63), (254, 2310, 127), (510, 1278, 151),
             (510, 1278, 359), (1022, 2558, 39), (2046, 33790, 681), (2046, 33790, 1365), (6, 6, 6), (1022, 1022, 1022)]
    out["estimatescore_exact"] = [exact_estimatescore(*p) for p in picks]
    # sigma beyond S1 + 1: (N' - n) > 0 makes the radicand negative -> sq = 0 (cannot happen in a run; pins the branch)
    v = exact_estimatescore(6, 10, 9)
    assert v["radicand"] < 0
    out["estimatescore_exact"].append(v)
    out["estimatescore_wrap"] = [wrap_estimatescore(*p) for p in
                                 [(31250, 1_000_000, 200), (31250, 1_000_000, 400), (31250, 1_000_000, 5000), (312500, 10_000_000, 0),
                                  (312500, 10_000_000, 3000), (312500, 10_000_000, 312500), (1562500, 50_000_000, 12345),
                                  (25000, 50_000, 12000)]]
    assert sum(v["wrapped"] for v in out["estimatescore_wrap"]) >= 6 and not out["estimatescore_wrap"][-1]["wrapped"]
    out["prob_exact"] = [exact_prob(512, 2, 1024, 3), exact_prob(768, 3, 1024, 2), exact_prob(256, 1, 1024, 3), exact_prob(1024, 5, 1024, 3),
                         exact_prob(0, 7, 1024, 3), exact_prob(640, 2, 1024, 1), exact_prob(512, 4, 2048, 2)]
    out["decisions"] = [decisions("cfg1", {}, 1234)]
    if "--no-cfg3" not in sys.argv:
        import ransac_jl_amd as R
        from ransac_jl_amd import _lib as L
        out["decisions"].append(decisions("cfg3", dict(score_mode=L.SCORE_F64, sphere_uses_enabled=True, sampling_streams=1), 1234, itermax=768))
    else:
        try:
            out["decisions"] += [d for d in json.load(open(OUT))["decisions"] if d["workload"] != "cfg1"]
        except (OSError, KeyError, ValueError):
            pass
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    for d in out["decisions"]:
        print("%s: %d decisions, %d extractions, smallest |ppp - prob_det| %.3g, flips under +-1 ulp pow: %d" %
              (d["workload"], d["decisions"], d["extractions"], d["min_margin"], d["flips_under_1ulp_pow"]))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
