"""A second, independent restatement of the four per-point tests (vectorised numpy, written from
the reference source text: shapes/plane.jl:82-130, sphere.jl:144-172, cylinder.jl:194-221,
cone.jl:68-85 + 132-153, utilities.jl:19-43,61-64) checked bit for bit against the C oracle.
numpy float64 element-wise ops are IEEE and never fused, so equal operation order => equal bits."""
import math

import numpy as np
import pytest

from oracle import oracle as orc
from ransac_jl_amd import synth


def dot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def norm(a):
    return np.sqrt((a[..., 0] * a[..., 0] + a[..., 1] * a[..., 1]) + a[..., 2] * a[..., 2])


def normalize(a):
    return (1.0 / norm(a))[..., None] * a


def cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


# The comparisons promote to float64 first: that is a no-op for float64 arrays and what Julia does when a Float32
# quantity meets a Float64 threshold (the Float32 twin below runs the same functions on float32 arrays).
def f64(x):
    return np.asarray(x, dtype=np.float64)


def compat_plane(point, normal, P, N, eps, cosa):
    o_z = normalize(normal)
    d = dot(o_z[None, :], P - point)
    return (f64(dot(normal[None, :], N)) > cosa) & (f64(np.abs(d)) < eps)


def compat_sphere(o, R, outw, P, N, eps, cosa):
    u = normalize(P - o) if outw else normalize(o - P)
    return (f64(dot(u, N)) > cosa) & (f64(np.abs(norm(P - o) - R)) < eps)


def compat_cylinder(a, c, R, outw, P, N, eps, cosa):
    cn = (P - a[None, :] * dot(a[None, :], P - c)[:, None]) - c
    band = f64(np.abs(norm(cn) - R)) < eps
    u = normalize(cn)
    if not outw:
        u = -u
    return band & (f64(dot(u, N)) > cosa)


def compat_cone(apex, axis, opang, outw, P, N, eps, cosa, cs=None):
    to_point = apex - P
    to_pointn = normalize(to_point)
    rot_ax = normalize(cross(np.broadcast_to(axis, P.shape), to_pointn))
    comp_n = normalize(cross(np.broadcast_to(axis, P.shape), rot_ax))
    nv = normalize(rot_ax)                                   # rodriguesrad re-normalizes
    th = -opang / 2
    c, s = (math.cos(th), math.sin(th)) if cs is None else cs   # cs: the host-computed pair of the record (Float32 twin)
    nn = nv[:, :, None] * nv[:, None, :]
    R = nn + c * (np.eye(3, dtype=P.dtype)[None] - nn)
    R[:, 0, 1] -= s * nv[:, 2]; R[:, 0, 2] += s * nv[:, 1]   # pluscrossprod!
    R[:, 1, 0] += s * nv[:, 2]; R[:, 1, 2] -= s * nv[:, 0]
    R[:, 2, 0] -= s * nv[:, 1]; R[:, 2, 1] += s * nv[:, 0]
    rc = np.stack([(R[:, i, 0] * comp_n[:, 0] + R[:, i, 1] * comp_n[:, 1]) + R[:, i, 2] * comp_n[:, 2] for i in range(3)], axis=-1)
    cn = normalize(rc)
    dist = dot(-cn, -to_point)
    par = f64(dot(cn if outw else -cn, N)) > cosa
    return par & (f64(np.abs(dist)) < eps)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_numpy_twin_equals_c_oracle(seed):
    prim = ["plane", "sphere", "cylinder", "cone"]
    xyz, nrm, truth = synth.make_cloud(20_000, prim, 0.3, seed=200 + seed)
    sub = np.arange(1, 20_001, dtype=np.int64)
    oc = orc.Cloud(xyz, nrm, sub)
    p = orc.default_params()
    with np.errstate(all="ignore"):
        for name, outw, v in synth.jittered_candidates(truth, 24, seed=seed):
            v = np.asarray(v, dtype=np.float64)
            k = {"plane": orc.PLANE, "sphere": orc.SPHERE, "cylinder": orc.CYLINDER, "cone": orc.CONE}[name]
            shape = orc.make_shape(k, outw, v)
            eps, cosa = p.eps[k], p.cos_alpha[k]
            if name == "plane":
                m = compat_plane(v[0:3], v[3:6], xyz, nrm, eps, cosa)
            elif name == "sphere":
                m = compat_sphere(v[0:3], v[3], outw, xyz, nrm, eps, cosa)
            elif name == "cylinder":
                m = compat_cylinder(v[0:3], v[3:6], v[6], outw, xyz, nrm, eps, cosa)
            else:
                m = compat_cone(v[0:3], v[3:6], v[6], outw, xyz, nrm, eps, cosa)
            cnt, inp = oc.scorecandidate(shape, p)
            assert cnt == int(m.sum()), name
            assert np.array_equal(inp, np.nonzero(m)[0] + 1), name


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_numpy_float32_twin_equals_c_oracle_f32(seed):
    """The same restatement on float32 arrays (numpy keeps float32 through every element-wise operation; Python scalars
    do not upcast) against oracle/orc_f32.c, the binary32 twin used for Float32 clouds: bit for bit."""
    prim = ["plane", "sphere", "cylinder", "cone"]
    xyz, nrm, truth = synth.make_cloud(20_000, prim, 0.3, seed=300 + seed)
    x32, n32 = xyz.astype(np.float32), nrm.astype(np.float32)
    sub = np.arange(1, 20_001, dtype=np.int64)
    oc = orc.Cloud32(x32, n32, sub)
    p = orc.default_params(sphere_uses_enabled=1)
    F = np.float32
    with np.errstate(all="ignore"):
        for name, outw, v in synth.jittered_candidates(truth, 24, seed=seed):
            k = {"plane": orc.PLANE, "sphere": orc.SPHERE, "cylinder": orc.CYLINDER, "cone": orc.CONE}[name]
            shape = orc.make_shape32(k, outw, v)
            v = np.array([shape.v[i] for i in range(9)]).astype(F)     # the binary32 fields of the Float32 shape
            eps, cosa = p.eps[k], p.cos_alpha[k]
            if name == "plane":
                m = compat_plane(v[0:3], v[3:6], x32, n32, eps, cosa)
            elif name == "sphere":
                m = compat_sphere(v[0:3], v[3], outw, x32, n32, eps, cosa)
            elif name == "cylinder":
                m = compat_cylinder(v[0:3], v[3:6], v[6], outw, x32, n32, eps, cosa)
            else:
                m = compat_cone(v[0:3], v[3:6], v[6], outw, x32, n32, eps, cosa, cs=(v[7], v[8]))
            assert m.dtype == bool
            counts = oc.score_batch([shape], p)
            ex = oc.refit(shape, p)
            assert counts[0] == int(m.sum()), name
            assert np.array_equal(ex, np.nonzero(m)[0] + 1), name

