"""CPU prototype: how many (candidate, 64-point group) pairs survive the box tests for different
orderings of subset 1 (3-D k-d leaves vs 6-D position+normal k-d leaves) and with an extra normal-box test.
Not part of the product or the tests: a design aid for the score kernel."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ransac_jl_amd import synth

EPS, COSA = 0.3, np.cos(np.radians(5.0))


def kd_order(keys, leaf=64):
    """keys: (s, d) array; median split along the widest axis, left child a multiple of `leaf`"""
    s = keys.shape[0]
    order = np.arange(s)
    stack = [(0, s)]
    while stack:
        lo, hi = stack.pop()
        cnt = hi - lo
        if cnt <= leaf:
            continue
        sub = keys[order[lo:hi]]
        ext = sub.max(0) - sub.min(0)
        ax = int(np.argmax(ext))
        nl = ((cnt // leaf + 1) // 2) * leaf
        part = np.argpartition(sub[:, ax], nl - 1 if nl < cnt else cnt - 1)
        order[lo:hi] = order[lo:hi][part]
        stack.append((lo + nl, hi))
        stack.append((lo, lo + nl))
    return order


def kd_order_hybrid(P, N, tile=256, leaf=64):
    """position k-d down to `tile`-point nodes, then split each tile by NORMAL (widest normal axis, median) into leaves"""
    order = kd_order(P, leaf=tile)
    s = P.shape[0]
    out = order.copy()
    for lo in range(0, s, tile):
        hi = min(s, lo + tile)
        idx = order[lo:hi]
        sub_order = kd_order(N[idx], leaf=leaf)
        out[lo:hi] = idx[sub_order]
    return out


def boxes(a, leaf=64):
    g = a.shape[0] // leaf
    a = a[: g * leaf].reshape(g, leaf, -1)
    mn, mx = a.min(1), a.max(1)
    return 0.5 * (mn + mx), 0.5 * (mx - mn)


def survivors(cands, pc, ph, nc, nh, P, N, use_normal, leaf=64):
    """returns per kind: total pairs, surviving pairs, pairs with >= 1 true inlier, lanes in band / passing"""
    g = pc.shape[0]
    hr = np.linalg.norm(ph, axis=1)
    nhr = np.linalg.norm(nh, axis=1)
    out = {}
    Pg = P[: g * leaf].reshape(g, leaf, 3)
    Ng = N[: g * leaf].reshape(g, leaf, 3)
    for kind, outw, v in cands:
        v = np.asarray(v, dtype=float)
        st = out.setdefault(kind, dict(pairs=0, surv=0, live=0, inl=0, band=0))
        sgn = 1.0 if outw else -1.0
        if kind == "plane":
            p0, Nn = v[0:3], v[3:6]
            oz = Nn / np.linalg.norm(Nn)
            d = (pc - p0) @ oz
            ext = ph @ np.abs(oz)
            keep = np.abs(d) <= ext + EPS
            if use_normal:
                mx = nc @ Nn + nh @ np.abs(Nn)
                keep &= mx > COSA
            idx = np.nonzero(keep)[0]
            dd = (Pg[idx] - p0) @ oz
            band = np.abs(dd) < EPS
            inl = band & ((Ng[idx] @ Nn) > COSA)
        elif kind == "sphere":
            o, R = v[0:3], v[3]
            a = np.abs(pc - o)
            dmin = np.linalg.norm(np.maximum(a - ph, 0), axis=1)
            dmax = np.linalg.norm(a + ph, axis=1)
            keep = (dmin <= R + EPS) & (dmax >= R - EPS)
            if use_normal:
                t = (pc - o) * sgn
                M = np.einsum("ij,ij->i", t, nc) + np.einsum("ij,ij->i", np.abs(nc), ph) + np.einsum("ij,ij->i", np.abs(t), nh) + np.einsum("ij,ij->i", ph, nh)
                dm = np.maximum(dmin, R - EPS)
                keep &= M > COSA * dm
            idx = np.nonzero(keep)[0]
            q = Pg[idx] - o
            nr = np.linalg.norm(q, axis=2)
            band = np.abs(nr - R) < EPS
            inl = band & (sgn * np.einsum("gli,gli->gl", q, Ng[idx]) / nr > COSA)
        elif kind == "cylinder":
            ax, c0, R = v[0:3], v[3:6], v[6]
            t = pc - c0
            qc = t - np.outer(t @ ax, ax)
            rho = np.linalg.norm(qc, axis=1)
            keep = (rho <= R + EPS + hr) & (rho >= R - EPS - hr)
            if use_normal:
                A = np.eye(3) - np.outer(ax, ax)
                An = nc @ A            # A symmetric: (A^T nc)
                qs = qc * sgn
                M = np.einsum("ij,ij->i", qs, nc) + np.einsum("ij,ij->i", np.abs(An), ph) + np.einsum("ij,ij->i", np.abs(qs), nh) + hr * nhr
                dm = np.maximum(rho - hr, R - EPS)
                keep &= M > COSA * np.maximum(dm, 0)
            idx = np.nonzero(keep)[0]
            tt = Pg[idx] - c0
            q = tt - (tt @ ax)[..., None] * ax
            nr = np.linalg.norm(q, axis=2)
            band = np.abs(nr - R) < EPS
            inl = band & (sgn * np.einsum("gli,gli->gl", q, Ng[idx]) / nr > COSA)
        else:
            continue
        st["pairs"] += g
        st["surv"] += idx.size
        st["live"] += int(inl.any(axis=1).sum())
        st["inl"] += int(inl.sum())
        st["band"] += int(band.sum())
    return out


def main():
    npts = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    ncand = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    prim = ["plane"] * 16 + ["sphere"] * 12 + ["cylinder"] * 12
    xyz, nrm, truth = synth.make_cloud(npts, prim, 0.30, seed=3)
    subs = synth.make_subsets(npts, 32, seed=3)
    P0, N0 = xyz[subs[0] - 1], nrm[subs[0] - 1]
    cands = synth.jittered_candidates(truth, 4096, seed=0)[:ncand]
    for name, lam in (("kd3", None), ("hyb256", 256), ("hyb512", 512), ("hyb1024", 1024)):
        t0 = time.time()
        order = kd_order(P0) if lam is None else kd_order_hybrid(P0, N0, tile=lam)
        P, N = P0[order], N0[order]
        pc, ph = boxes(P)
        nc, nh = boxes(N)
        for un in (False, True):
            r = survivors(cands, pc, ph, nc, nh, P, N, un)
            line = "%-12s normal_test=%d " % (name, un)
            for k, st in r.items():
                line += " | %s surv %.2f%% live %.2f%% band-lanes/surv-group %.1f inl %d" % (
                    k, 100 * st["surv"] / st["pairs"], 100 * st["live"] / st["pairs"], st["band"] / max(1, st["surv"]), st["inl"])
            print(line, "(%.0fs)" % (time.time() - t0), flush=True)
        print("   mean pos radius %.2f, mean normal radius %.3f" % (np.linalg.norm(ph, axis=1).mean(), np.linalg.norm(nh, axis=1).mean()))


if __name__ == "__main__":
    main()
