#!/usr/bin/env python3
"""CPU prototype (numpy, idealised tests without margins) behind DESIGN.md 4.2: would a leaf order that also splits on the
NORMALS (6-D k-d tree, normals weighted by kappa) and a normal box test cull more (candidate, group) pairs than the position-only
order?  cfg3 batch: position-only 8.3 % of the pairs survive; with a normal test on the same order 7.7 %; 6-D orders 7.9 %
(kappa = 10) .. 14 % (kappa = 100) -- planes cull 2x better, cylinders 2x worse.  Not built.
python tools/proto/cull6d_proto.py [cfg3|cfg5] [points]"""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ransac_jl_amd as R
from ransac_jl_amd import synth
np.seterr(all='ignore')
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
if cfg == 'cfg3':
    prim = ["plane"] * 16 + ["sphere"] * 12 + ["cylinder"] * 12; n = 10_000_000; seed = 3; scanner = None
else:
    prim = ["plane"] * 16 + ["sphere"] * 12 + ["cylinder"] * 12 + ["cone"] * 8; n = 50_000_000; seed = 5; scanner = [synth.BOX / 2] * 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else n
t0 = time.time()
xyz, nrm, truth = synth.make_cloud(n, prim, 0.30, seed=seed, scanner=scanner)
subs = synth.make_subsets(n, 32, seed=seed)
idx = subs[0] - 1
P, N = xyz[idx], nrm[idx]
S = len(idx)
print("scene", time.time() - t0, "S", S)
cands = synth.jittered_candidates(truth, 4096, seed=0)

def kd_order(keys):
    """keys: S x D; balanced k-d, widest axis, leaves of 64"""
    order = np.arange(len(keys))
    stack = [(0, len(keys))]
    while stack:
        lo, hi = stack.pop()
        cnt = hi - lo
        if cnt <= 64: continue
        sub = keys[order[lo:hi]]
        ax = np.argmax(sub.max(0) - sub.min(0))
        nl = ((cnt // 64 + 1) // 2) * 64
        part = np.argpartition(sub[:, ax], nl)
        order[lo:hi] = order[lo:hi][part]
        stack.append((lo + nl, hi)); stack.append((lo, lo + nl))
    return order

def boxes(order):
    G = (S + 63) // 64
    pad = G * 64 - S
    o = np.concatenate([order, np.repeat(order[-1], pad)])
    Pg, Ng = P[o].reshape(G, 64, 3), N[o].reshape(G, 64, 3)
    pc, ph = (Pg.max(1) + Pg.min(1)) / 2, (Pg.max(1) - Pg.min(1)) / 2
    nc, nh = (Ng.max(1) + Ng.min(1)) / 2, (Ng.max(1) - Ng.min(1)) / 2
    return pc, ph, nc, nh

eps, cosa = 0.3, np.cos(np.radians(5.0))
def survive(c, bx, use_n):
    pc, ph, nc, nh = bx
    kind, outw, v = c
    v = np.asarray(v, dtype=float)
    hr = np.linalg.norm(ph, axis=1); nr = np.linalg.norm(nh, axis=1); ncn = np.linalg.norm(nc, axis=1)
    sg = 1.0 if outw else -1.0
    if kind == "plane":
        p0, nn = v[:3], v[3:6]; oz = nn / np.linalg.norm(nn)
        d = (pc - p0) @ oz
        ok = np.abs(d) <= eps + ph @ np.abs(oz)
        okn = nc @ nn + nh @ np.abs(nn) >= cosa
    elif kind == "sphere":
        o, Rr = v[:3], v[3]
        a = np.abs(pc - o); dmin = np.linalg.norm(np.maximum(a - ph, 0), axis=1); dmax = np.linalg.norm(a + ph, axis=1)
        ok = (dmin <= Rr + eps) & (dmax >= Rr - eps)
        u = pc - o; d = np.linalg.norm(u, axis=1); uh = u / d[:, None]
        okn = (d <= hr) | (sg * np.einsum('ij,ij->i', uh, nc) + nr + np.minimum(2.0, 2 * hr / d) * (ncn + nr) >= cosa)
    elif kind == "cylinder":
        a, c0, Rr = v[:3], v[3:6], v[6]
        t = pc - c0; q = t - np.outer(t @ a, a); rho = np.linalg.norm(q, axis=1)
        lip = max(1.0, abs(1 - a @ a)) * hr
        ok = (rho <= Rr + eps + lip) & (rho >= Rr - eps - lip)
        uh = q / rho[:, None]
        okn = (rho <= lip) | (sg * np.einsum('ij,ij->i', uh, nc) + nr + np.minimum(2.0, 2 * lip / rho) * (ncn + nr) >= cosa)
    else:
        apex, ax, om = v[:3], v[3:6], v[6]; ax = ax / np.linalg.norm(ax)
        c_, s_ = np.cos(-om / 2), np.sin(-om / 2)
        w = pc - apex; h = w @ ax; q = w - np.outer(h, ax); rho = np.linalg.norm(q, axis=1)
        D = c_ * rho + s_ * h
        ok = np.abs(D) <= eps + hr
        er = q / rho[:, None]
        g = c_ * er + s_ * ax
        okn = (rho <= hr) | (sg * np.einsum('ij,ij->i', g, nc) + nr + abs(c_) * np.minimum(2.0, 2 * hr / rho) * (ncn + nr) >= cosa)
    return (ok & okn) if use_n else ok

def rate(bx, use_n, label):
    tot = {}
    for c in cands[::8]:
        k = c[0]
        sv = survive(c, bx, use_n)
        a = tot.setdefault(k, [0, 0]); a[0] += int(sv.sum()); a[1] += len(sv)
    print("%-28s" % label, {k: round(v[0] / v[1], 4) for k, v in tot.items()}, "all %.4f" % (sum(v[0] for v in tot.values()) / sum(v[1] for v in tot.values())))

t0 = time.time()
oA = kd_order(P)
print("kd3", time.time() - t0)
bA = boxes(oA)
rate(bA, False, "3-D k-d, position test")
rate(bA, True, "3-D k-d, + normal test")
for kappa in (10.0, 25.0, 50.0, 100.0):
    oB = kd_order(np.hstack([P, N * kappa]))
    bB = boxes(oB)
    rate(bB, True, "6-D k-d kappa=%g, both" % kappa)
    rate(bB, False, "6-D k-d kappa=%g, pos only" % kappa)
