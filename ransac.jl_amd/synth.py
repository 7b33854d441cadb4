"""Synthetic multi-primitive clouds.  The reference ships no
generator (`examplepc3` is external, NEWS.md:43); this one defines the bench and
parity inputs: numpy default_rng(seed), scene box [0,100]^3, float64, unit
normals; inliers uniform by area on each primitive, displaced along the true
normal by N(0, 0.02), normals perturbed by <= 2 degrees; outliers uniform in the
box with uniformly random unit normals."""
import numpy as np

BOX = 100.0
SIGMA = 0.02
MAX_NORMAL_DEG = 2.0


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _frame(rng):
    z = _unit(rng.normal(size=3))
    a = np.array([1.0, 0, 0]) if abs(z[0]) < 0.9 else np.array([0, 1.0, 0])
    x = _unit(np.cross(z, a))
    return x, np.cross(z, x), z


def _perturb(n, rng):
    m = n.shape[0]
    ang = np.radians(MAX_NORMAL_DEG) * np.sqrt(rng.random(m))
    t = _unit(np.cross(n, rng.normal(size=(m, 3))))
    return _unit(n * np.cos(ang)[:, None] + t * np.sin(ang)[:, None])


# every primitive: draw its parameters, then its points (two steps, the same draws in the same order as
# one call -- the scanner-density sampler below needs the second step on its own)
def plane_params(rng, size=None):
    x, y, z = _frame(rng)
    size = size if size is not None else rng.uniform(20, 50)
    c = rng.uniform(0.25 * BOX, 0.75 * BOX, size=3)
    return dict(kind="plane", point=c, normal=z, _x=x, _y=y, _size=size)


def plane_points(t, m, rng):
    c, x, y, z, size = t["point"], t["_x"], t["_y"], t["normal"], t["_size"]
    uv = rng.uniform(-size / 2, size / 2, size=(m, 2))
    p = c + uv[:, :1] * x + uv[:, 1:] * y + rng.normal(0, SIGMA, size=(m, 1)) * z
    n = np.repeat(z[None], m, 0)
    return p, _perturb(n, rng)


def sphere_params(rng, radius=None):
    r = radius if radius is not None else rng.uniform(5, 15)
    c = rng.uniform(r + 1, BOX - r - 1, size=3)
    return dict(kind="sphere", center=c, radius=r, outwards=True)


def sphere_points(t, m, rng):
    c, r = t["center"], t["radius"]
    d = _unit(rng.normal(size=(m, 3)))
    p = c + (r + rng.normal(0, SIGMA, size=(m, 1))) * d
    return p, _perturb(d, rng)


def cylinder_params(rng):
    x, y, z = _frame(rng)
    r, h = rng.uniform(3, 10), rng.uniform(20, 50)
    c = rng.uniform(0.3 * BOX, 0.7 * BOX, size=3)
    return dict(kind="cylinder", axis=z, center=c, radius=r, outwards=True, _x=x, _y=y, _h=h)


def cylinder_points(t, m, rng):
    x, y, z, c, r, h = t["_x"], t["_y"], t["axis"], t["center"], t["radius"], t["_h"]
    th = rng.uniform(0, 2 * np.pi, size=m)
    tt = rng.uniform(-h / 2, h / 2, size=m)
    d = np.cos(th)[:, None] * x + np.sin(th)[:, None] * y
    p = c + tt[:, None] * z + (r + rng.normal(0, SIGMA, size=(m, 1))) * d
    return p, _perturb(d, rng)


def cone_params(rng):
    x, y, z = _frame(rng)
    half = np.radians(rng.uniform(10, 35))
    h1 = rng.uniform(25, 45)
    apex = rng.uniform(0.3 * BOX, 0.7 * BOX, size=3)
    return dict(kind="cone", apex=apex, axis=z, opang=2 * half, outwards=True, _x=x, _y=y, _h0=5.0, _h1=h1)


def cone_points(t, m, rng):
    x, y, z, apex, half, h0, h1 = t["_x"], t["_y"], t["axis"], t["apex"], t["opang"] / 2, t["_h0"], t["_h1"]
    # uniform by area: slant distance density ~ s
    s = np.sqrt(rng.uniform(h0 ** 2, h1 ** 2, size=m))
    th = rng.uniform(0, 2 * np.pi, size=m)
    radial = np.cos(th)[:, None] * x + np.sin(th)[:, None] * y
    gen = np.cos(half) * z + np.sin(half) * radial          # generator direction
    nrm = np.cos(half) * radial - np.sin(half) * z          # outward surface normal
    p = apex + s[:, None] * gen + rng.normal(0, SIGMA, size=(m, 1)) * nrm
    return p, _perturb(nrm, rng)


_PARAMS = {"plane": plane_params, "sphere": sphere_params, "cylinder": cylinder_params, "cone": cone_params}
_POINTS = {"plane": plane_points, "sphere": sphere_points, "cylinder": cylinder_points, "cone": cone_points}


def _public(t):
    return {k: v for k, v in t.items() if not k.startswith("_")}


def plane_patch(m, rng, size=None):
    t = plane_params(rng, size)
    p, n = plane_points(t, m, rng)
    return p, n, _public(t)


def sphere(m, rng, radius=None):
    t = sphere_params(rng, radius)
    p, n = sphere_points(t, m, rng)
    return p, n, _public(t)


def cylinder(m, rng):
    t = cylinder_params(rng)
    p, n = cylinder_points(t, m, rng)
    return p, n, _public(t)


def cone(m, rng):
    t = cone_params(rng)
    p, n = cone_points(t, m, rng)
    return p, n, _public(t)


def _centre(t):
    if t["kind"] == "plane":
        return t["point"]
    if t["kind"] == "cone":
        return t["apex"] + 0.5 * t["_h1"] * t["axis"]
    return t["center"]


_GEN = {"plane": plane_patch, "sphere": sphere, "cylinder": cylinder, "cone": cone}


def make_cloud(n_total, primitives, outlier_frac=0.0, seed=0, weights=None, scanner=None):
    """primitives: list of kind names.  Returns (xyz[N,3], nrm[N,3], truth list).
    scanner: a 3-vector -> scanned-scene-style density (cfg5): a primitive's share of the inliers falls off
    as 1 / max(d, 10)^2 with the distance d of its centre from that point (all parameters are drawn first,
    then the points; within a primitive the points stay uniform by area)."""
    rng = np.random.default_rng(seed)
    n_out = int(round(n_total * outlier_frac))
    n_in = n_total - n_out
    k = len(primitives)
    w = np.ones(k) if weights is None else np.asarray(weights, dtype=np.float64)
    pre = None
    if scanner is not None:
        sc = np.asarray(scanner, dtype=np.float64)
        pre = [_PARAMS[kind](rng) for kind in primitives]
        w = w / np.maximum(np.array([np.linalg.norm(_centre(t) - sc) for t in pre]), 10.0) ** 2
    sizes = np.floor(n_in * w / w.sum()).astype(np.int64)
    sizes[0] += n_in - sizes.sum()
    P, N, truth = [], [], []
    for i, (kind, m) in enumerate(zip(primitives, sizes)):
        if pre is None:
            p, n, t = _GEN[kind](int(m), rng)
        else:
            p, n = _POINTS[kind](pre[i], int(m), rng)
            t = _public(pre[i])
        t["n_points"] = int(m)
        P.append(p); N.append(n); truth.append(t)
    if n_out:
        P.append(rng.uniform(0, BOX, size=(n_out, 3)))
        N.append(_unit(rng.normal(size=(n_out, 3))))
    xyz = np.ascontiguousarray(np.concatenate(P), dtype=np.float64)
    nrm = np.ascontiguousarray(np.concatenate(N), dtype=np.float64)
    perm = rng.permutation(n_total)  # interleave primitives like a scan would
    return xyz[perm], nrm[perm], truth


def make_subsets(n, r, seed=0):
    """RANSACCloud(vertices, normals, numofsubsets) split: randperm into r
    contiguous slices, the last takes the remainder (octree.jl:129-135). 1-based."""
    rng = np.random.default_rng(seed + 7919)
    alls = rng.permutation(n).astype(np.int64) + 1
    ssl = n // r
    subs = [alls[i * ssl:(i + 1) * ssl] for i in range(r - 1)]
    subs.append(alls[(r - 1) * ssl:])
    return subs


# the BASELINE.json / SURVEY.md 8(d) configurations
def config(name):
    if name == "cfg1":   # 50k plane + sphere, r = 2 (faithful-parity config, no Int64 wrap)
        rng = np.random.default_rng(1234)
        p1, n1, t1 = plane_patch(25000, rng)
        p2, n2, t2 = sphere(25000, rng, radius=10.0)
        xyz, nrm = np.concatenate([p1, p2]), np.concatenate([n1, n2])
        perm = rng.permutation(50000)
        return dict(xyz=np.ascontiguousarray(xyz[perm]), nrm=np.ascontiguousarray(nrm[perm]),
                    truth=[t1, t2], r=2, seed=1234)
    if name == "cfg2":   # 1M, 2 planes + 2 spheres + 2 cylinders, r = 32, B = 4096
        prim = ["plane", "plane", "sphere", "sphere", "cylinder", "cylinder"]
        xyz, nrm, truth = make_cloud(1_000_000, prim, 0.0, seed=2)
        return dict(xyz=xyz, nrm=nrm, truth=truth, r=32, seed=2)
    if name in ("cfg3", "cfg4"):  # 10M = 7M inliers over 40 primitives + 3M outliers
        prim = ["plane"] * 16 + ["sphere"] * 12 + ["cylinder"] * 12
        xyz, nrm, truth = make_cloud(10_000_000, prim, 0.30, seed=3)
        return dict(xyz=xyz, nrm=nrm, truth=truth, r=32, seed=3)
    if name == "cfg5":   # 50M with cones, scanner at the box centre
        prim = ["plane"] * 16 + ["sphere"] * 12 + ["cylinder"] * 12 + ["cone"] * 8
        xyz, nrm, truth = make_cloud(50_000_000, prim, 0.30, seed=5, scanner=[BOX / 2] * 3)
        return dict(xyz=xyz, nrm=nrm, truth=truth, r=32, seed=5)
    raise KeyError(name)


def jittered_candidates(truth, b, seed=0, jitter=0.01):
    """Candidate microbenchmark batch: ground-truth primitives with
    parameters jittered by 1 %, cycled across the truth list.  Returns a list of
    (kind, outwards, v[<=7]) tuples in the rh_shape field order."""
    rng = np.random.default_rng(seed + 99)
    out = []
    for i in range(b):
        t = truth[i % len(truth)]
        j = lambda x: np.asarray(x, dtype=np.float64) * (1 + jitter * rng.uniform(-1, 1, size=np.shape(x)))
        if t["kind"] == "plane":
            out.append(("plane", False, list(j(t["point"])) + list(_unit(j(t["normal"])))))
        elif t["kind"] == "sphere":
            out.append(("sphere", bool(rng.integers(0, 8) > 0), list(j(t["center"])) + [float(j(t["radius"]))]))
        elif t["kind"] == "cylinder":
            out.append(("cylinder", bool(rng.integers(0, 8) > 0),
                        list(_unit(j(t["axis"]))) + list(j(t["center"])) + [float(j(t["radius"]))]))
        else:
            out.append(("cone", bool(rng.integers(0, 8) > 0),
                        list(j(t["apex"])) + list(_unit(j(t["axis"]))) + [float(j(t["opang"]))]))
    return out
