"""Writes compat_exact_vectors.json: hand-derived known-answer vectors for the four per-point compatibility
tests of the hot path (compatiblesPlane plane.jl:114-130 + project2plane :82-95, compatiblesSphere
sphere.jl:144-172, compatiblesCylinder cylinder.jl:194-221, compatiblesCone cone.jl:132-153 + project2cone
:68-85; paths under /root/reference/src), for scorecandidate's use of the enabled bits (Q4, sphere.jl:121,131)
and for validatecone's missing abs (Q11, cone.jl:93).

The reference's tests hold no vector for these functions and Julia cannot run here, so the
vectors are derived BY HAND from the source text, two kinds of them:

* "exact": coordinates are dyadic rationals, directions axis-aligned with power-of-two lengths, radii
  Pythagorean (3-4-5), so that EVERY intermediate of the reference's formula is exactly representable in
  binary64.  The outcome then does not depend on the evaluation order, on FMA contraction or on
  reciprocal-vs-divide -- i.e. not on the unpinned StaticArrays semantics -- and the vector pins the
  comparison itself: `<` (not `<=`) at exactly eps, `>` (not `>=`) at exactly cos(alpha), abs(), the sign
  convention of inward shapes, and quirks Q4 / Q11 / Q12.  This script re-evaluates each formula in exact
  rational arithmetic (fractions.Fraction), asserts that every intermediate is representable and that the
  hand-derived expectation is what the formula gives.
* "exact_distance": the distance side is exact as above (Pythagorean radial vector (3,4,0), norm 5) and sits at /
  2^-50 below eps, while the angle side (which needs inv(5), not exact) is ~1.0 against a threshold of 0.5.
* "robust": general-position cases (Pythagorean unit directions such as (2,-1,2)/3, 45 / 30 degree cones)
  where rounding cannot matter because the compared quantity is >= 1e-3 away from its threshold; the
  expectation comes from the closed-form geometry stated in the derivation (evaluated here in float64 and
  asserted to keep that margin).  They pin the meaning of the parameters: which normal the angle test uses,
  that `opang` is the FULL opening angle, that the cone's axis points from the apex into the cone, that the
  mirror nappe is not accepted, that a non-unit cylinder axis is used as is.

tests/test_compat_exact_vectors.py runs them through the oracle, through every rounding-order variant of the
oracle (CPU) and through the HIP kernels (score, both paths, and refit; -m gpu).
Run once: python tests/golden/make_compat_exact_vectors.py; the JSON is committed.
"""
import json
import math
import os
from fractions import Fraction as Fr

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "compat_exact_vectors.json")
NAN = "nan"


# ---------------------------------------------------------------- exact evaluation (Fraction) ----
class Inexact(Exception):
    pass


def rep(x):
    """assert that the rational x is a binary64 value (normal range) and return it"""
    if x == NAN:
        return x
    x = Fr(x)
    if x != 0 and Fr(float(x)) != x:
        raise Inexact("not representable: %s" % x)
    return x


def fsqrt(x):
    if x == NAN:
        return NAN
    n, d = x.numerator, x.denominator
    rn, rd = math.isqrt(n), math.isqrt(d)
    if rn * rn != n or rd * rd != d:
        raise Inexact("sqrt(%s) is irrational" % x)
    return rep(Fr(rn, rd))


def mul(a, b):
    return NAN if NAN in (a, b) else rep(a * b)


def add(a, b):
    return NAN if NAN in (a, b) else rep(a + b)


def sub(a, b):
    return NAN if NAN in (a, b) else rep(a - b)


def inv(a):
    # inv(0) = Inf, and Inf * 0 = NaN in normalize(zero vector): every later comparison is false
    return NAN if a == NAN or a == 0 else rep(1 / a)


def dot(a, b):   # any association gives the same value when every partial sum is representable: check them all
    pr = [mul(x, y) for x, y in zip(a, b)]
    if NAN in pr:
        return NAN
    for s in (pr[0] + pr[1], pr[1] + pr[2], pr[0] + pr[2]):
        rep(s)
    return rep(pr[0] + pr[1] + pr[2])


def norm(a):
    return fsqrt(dot(a, a))


def normalize(a):
    i = inv(norm(a))
    if i == NAN:
        return [NAN] * 3
    nr = norm(a)
    for x in a:   # a / norm (the "div" reading) must give the same value
        rep(x / nr)
    return [mul(i, x) for x in a]


def cross(a, b):
    def ms(p, q, r, s):
        return sub(mul(p, q), mul(r, s))
    return [ms(a[1], b[2], a[2], b[1]), ms(a[2], b[0], a[0], b[2]), ms(a[0], b[1], a[1], b[0])]


def vsub(a, b):
    return [sub(x, y) for x, y in zip(a, b)]


def neg(a):
    return [NAN if x == NAN else -x for x in a]


def gt(a, b):
    return False if NAN in (a, b) else a > b


def lt(a, b):
    return False if NAN in (a, b) else a < b


def fabs(a):
    return NAN if a == NAN else abs(a)


def compat_plane(sh, p, n, eps, cosa):            # plane.jl:82-95, 128; utilities.jl:115-117
    point, normal = sh["v"][0:3], sh["v"][3:6]
    o_z = normalize(normal)
    d = dot(o_z, vsub(p, point))
    return gt(dot(normal, n), cosa) and lt(fabs(d), eps)


def compat_sphere(sh, p, n, eps, cosa):           # sphere.jl:164-166
    o, R = sh["v"][0:3], sh["v"][3]
    u = normalize(vsub(p, o)) if sh["outwards"] else normalize(vsub(o, p))
    return gt(dot(u, n), cosa) and lt(fabs(sub(norm(vsub(p, o)), R)), eps)


def compat_cylinder(sh, p, n, eps, cosa):         # cylinder.jl:207-214
    a, c, R = sh["v"][0:3], sh["v"][3:6], sh["v"][6]
    sd = dot(a, vsub(p, c))
    cn = vsub(vsub(p, [mul(x, sd) for x in a]), c)
    if lt(fabs(sub(norm(cn), R)), eps):
        u = normalize(cn)
        return gt(dot(u if sh["outwards"] else neg(u), n), cosa)
    return False


def compat_cone_degenerate(sh, p, n, eps, cosa):  # cone.jl:68-85 up to the first NaN (exact cases are NaN cases)
    apex, axis = sh["v"][0:3], sh["v"][3:6]
    tn = normalize(vsub(apex, p))
    rot_ax = normalize(cross(axis, tn)) if NAN not in tn else [NAN] * 3
    if NAN in rot_ax:
        return False
    raise Inexact("cone case is not degenerate")


EXACT = {"plane": compat_plane, "sphere": compat_sphere, "cylinder": compat_cylinder, "cone": compat_cone_degenerate}


# ---------------------------------------------------------------- closed-form geometry (robust) ----
def robust_values(sh, p, n):
    """(distance-side value, angle-side value) from the geometry the formulas implement, in float64"""
    import numpy as np
    v = np.array(sh["v"], dtype=float)
    p, n = np.array(p, dtype=float), np.array(n, dtype=float)
    k = sh["kind"]
    if k == "plane":
        N = v[3:6]
        return abs(np.dot(N / np.linalg.norm(N), p - v[0:3])), float(np.dot(N, n))
    if k == "sphere":
        d = p - v[0:3]
        u = d / np.linalg.norm(d)
        return abs(np.linalg.norm(d) - v[3]), float(np.dot(u if sh["outwards"] else -u, n))
    if k == "cylinder":
        a, c = v[0:3], v[3:6]
        q = p - a * np.dot(a, p - c) - c            # the literal formula: the axis is NOT normalised
        u = q / np.linalg.norm(q)
        return abs(np.linalg.norm(q) - v[6]), float(np.dot(u if sh["outwards"] else -u, n))
    apex, a, half = v[0:3], v[3:6], v[6] / 2
    t = p - apex
    h = np.dot(t, a)
    rad = t - h * a
    rho = np.linalg.norm(rad)
    e = rad / rho
    cn = e * math.cos(half) - a * math.sin(half)    # outward surface normal of the generatrix through p
    return abs(h * math.sin(half) - rho * math.cos(half)), float(np.dot(cn if sh["outwards"] else -cn, n))


# ------------------------------------------------------------------------------------- vectors ----
E, H = 0.25, 0.5          # eps and cos(alpha) of the exact vectors (cos_alpha is passed as a number at the C ABI)
t50, t53, t54 = 2.0 ** -50, 2.0 ** -53, 2.0 ** -54
cases = []


def add_case(name, kind, v, outwards, p, n, eps, cosa, expect, mode, why):
    cases.append({"name": name, "kind": kind, "v": [float(x) for x in v], "outwards": bool(outwards),
                  "p": [float(x) for x in p], "n": [float(x) for x in n], "eps": float(eps), "cos_alpha": float(cosa),
                  "expect": bool(expect), "mode": mode, "derivation": why})


# ---- plane: point (1,2,3), stored normal (0,0,2) [not unit: Q12], o_z = inv(2) * (0,0,2) = (0,0,1) exactly
PL = [1, 2, 3, 0, 0, 2]
add_case("plane_d_eq_eps", "plane", PL, True, [5, -7, 3.25], [0, 0, 1], E, H, False, "exact",
         "plane.jl:128 abs(p[3]) < eps: v = p - point = (4,-9,0.25), d = dot((0,0,1), v) = 0.25 = eps, strict < fails")
add_case("plane_d_below_eps", "plane", PL, True, [5, -7, 3.25 - t50], [0, 0, 1], E, H, True, "exact",
         "d = 0.25 - 2^-50 < eps; dot(normal, n) = 2 > 0.5")
add_case("plane_d_eq_minus_eps", "plane", PL, True, [5, -7, 2.75], [0, 0, 1], E, H, False, "exact",
         "d = -0.25, abs(d) = eps, strict < fails (abs is applied, plane.jl:128)")
add_case("plane_d_above_minus_eps", "plane", PL, True, [5, -7, 2.75 + t50], [0, 0, 1], E, H, True, "exact",
         "d = -(0.25 - 2^-50), abs(d) < eps")
add_case("plane_dot_eq_cos", "plane", PL, True, [5, -7, 3], [0, 0, 0.25], E, H, False, "exact",
         "utilities.jl:116 dot(v1, v2) > cos(alpha): dot((0,0,2),(0,0,0.25)) = 0.5 = cos_alpha, strict > fails")
add_case("plane_dot_above_cos", "plane", PL, True, [5, -7, 3], [0, 0, 0.25 + t54], E, H, True, "exact",
         "dot = 0.5 + 2^-53 > 0.5; d = 0")
add_case("plane_q12_angle_uses_stored_normal", "plane", PL, True, [5, -7, 3], [0, 0, 0.375], E, H, True, "exact",
         "Q12, plane.jl:128 isparallel(plane.normal, n): dot((0,0,2),(0,0,0.375)) = 0.75 > 0.5; with the NORMALISED "
         "normal it would be 0.375 and fail")
add_case("plane_q12_distance_uses_normalised_normal", "plane", PL, True, [5, -7, 3.1875], [0, 0, 1], E, H, True, "exact",
         "Q12, plane.jl:85 o_z = normalize(plane.normal): d = 0.1875 < 0.25; with the stored normal (0,0,2) it would be "
         "0.375 and fail")
add_case("plane_opposite_normal", "plane", PL, True, [5, -7, 3], [0, 0, -1], E, H, False, "exact",
         "dot(normal, n) = -2: the plane test is orientation sensitive")
add_case("plane_axis_y", "plane", [0, 0, 0, 0, -4, 0], True, [1, 0.125, 9], [0, -0.5, 0], E, H, True, "exact",
         "o_z = inv(4) * (0,-4,0) = (0,-1,0); d = -0.125, abs < 0.25; dot((0,-4,0),(0,-0.5,0)) = 2 > 0.5")

# ---- sphere: centre (1,1,1)
SPC = [1, 1, 1]
add_case("sphere_dist_eq_eps", "sphere", SPC + [4.75], True, [4, 5, 1], [0.6, 0.8, 0], E, H, False, "exact_distance",
         "sphere.jl:164 abs(norm(p-o) - R) < eps: p - o = (3,4,0), norm = sqrt(9+16) = 5 exactly, |5 - 4.75| = 0.25 = eps "
         "exactly, strict < fails (the angle side, ~1.0, is irrelevant)")
add_case("sphere_dist_below_eps", "sphere", SPC + [4.75 + t50], True, [4, 5, 1], [0.6, 0.8, 0], E, H, True, "exact_distance",
         "|5 - (4.75 + 2^-50)| = 0.25 - 2^-50 < eps exactly; angle side dot((3,4,0)/5, n) ~ 1.0, far from 0.5")
add_case("sphere_dot_eq_cos", "sphere", SPC + [4], True, [1, 1, 5], [0, 0, 0.5], E, H, False, "exact",
         "p - o = (0,0,4), norm 4, inv 0.25, normalize = (0,0,1) exactly; dot = 0.5 = cos_alpha, strict > fails; |4-4| = 0")
add_case("sphere_dot_above_cos", "sphere", SPC + [4], True, [1, 1, 5], [0, 0, 0.5 + t53], E, H, True, "exact",
         "dot = 0.5 + 2^-53 > 0.5")
add_case("sphere_inward_accepts_inward_normal", "sphere", SPC + [4], False, [1, 1, 5], [0, 0, -0.75], E, H, True, "exact",
         "sphere.jl:166 outwards = false: normalize(o - p) = (0,0,-1), dot with (0,0,-0.75) = 0.75 > 0.5")
add_case("sphere_inward_rejects_outward_normal", "sphere", SPC + [4], False, [1, 1, 5], [0, 0, 0.75], E, H, False, "exact",
         "dot((0,0,-1),(0,0,0.75)) = -0.75")
add_case("sphere_outward_rejects_inward_normal", "sphere", SPC + [4], True, [1, 1, 5], [0, 0, -0.75], E, H, False, "exact",
         "sphere.jl:164 outwards = true: normalize(p - o) = (0,0,1), dot = -0.75")
add_case("sphere_point_at_centre", "sphere", SPC + [0.125], True, [1, 1, 1], [0, 0, 1], E, H, False, "exact",
         "p = o: norm = 0, |0 - 0.125| < 0.25 passes, but normalize(0) = inv(0) * 0 = NaN and NaN > cos_alpha is false")
add_case("sphere_general_in", "sphere", [10, 20, 30, 9], True, [10 + 6, 20 - 3, 30 + 6], [2 / 3, -1 / 3, 2 / 3], 0.3, 0.9, True, "robust",
         "p - o = (6,-3,6) = 9 * (2,-1,2)/3: norm 9 = R (distance side 0), unit direction (2,-1,2)/3 = n (angle side 1)")
add_case("sphere_general_far", "sphere", [10, 20, 30, 9], True, [10 + 6.4, 20 - 3.2, 30 + 6.4], [2 / 3, -1 / 3, 2 / 3], 0.3, 0.9, False, "robust",
         "p - o = 9.6 * (2,-1,2)/3: |9.6 - 9| = 0.6 > 0.3")

# ---- cylinder: axis (0,0,1) through (1,1,0)
CY = [0, 0, 1, 1, 1, 0]
add_case("cyl_dist_eq_eps", "cylinder", CY + [4.75], True, [4, 5, 7], [0.6, 0.8, 0], E, H, False, "exact_distance",
         "cylinder.jl:207-209: p - c = (3,4,7), dot(a, .) = 7, p - a*7 - c = (3,4,0), norm 5 exactly, |5 - 4.75| = eps, < fails")
add_case("cyl_dist_below_eps", "cylinder", CY + [4.75 + t50], True, [4, 5, 7], [0.6, 0.8, 0], E, H, True, "exact_distance",
         "|5 - R| = 0.25 - 2^-50 < eps; angle side ~ 1.0")
add_case("cyl_dot_eq_cos", "cylinder", CY + [4], True, [1, 5, 9], [0, 0.5, 0], E, H, False, "exact",
         "curr_norm = (0,4,0), norm 4 = R; normalize = (0,1,0); dot = 0.5 = cos_alpha, strict > fails")
add_case("cyl_dot_above_cos", "cylinder", CY + [4], True, [1, 5, 9], [0, 0.5 + t53, 0], E, H, True, "exact",
         "dot = 0.5 + 2^-53")
add_case("cyl_inward_accepts_inward_normal", "cylinder", CY + [4], False, [1, 5, 9], [0, -0.75, 0], E, H, True, "exact",
         "cylinder.jl:213 -normalize(curr_norm) = (0,-1,0); dot = 0.75")
add_case("cyl_inward_rejects_outward_normal", "cylinder", CY + [4], False, [1, 5, 9], [0, 0.75, 0], E, H, False, "exact",
         "dot((0,-1,0),(0,0.75,0)) = -0.75")
add_case("cyl_axis_not_normalised", "cylinder", [0, 0, 2, 0, 0, 0, 5], True, [4, 0, 1], [0.8, 0, -0.6], E, H, True, "robust",
         "cylinder.jl:207 uses the axis as stored: dot(a, p-c) = 2, a*2 = (0,0,4), curr_norm = (4,0,-3), norm 5 = R, direction "
         "(0.8,0,-0.6) = n.  With a unit axis the radial vector would be (4,0,0), |4 - 5| = 1 > eps: rejected")
add_case("cyl_point_on_axis", "cylinder", CY + [0.125], True, [1, 1, 3], [0, 1, 0], E, H, False, "exact",
         "curr_norm = 0: |0 - 0.125| < 0.25 passes, normalize(0) = NaN, NaN > cos_alpha false")
add_case("cyl_general_in", "cylinder", [2 / 3, -1 / 3, 2 / 3, 10, 20, 30, 6], True,
         [10 + 6 * (2 / 3) + 5 * (2 / 3), 20 + 6 * (2 / 3) - 5 * (1 / 3), 30 - 6 * (1 / 3) + 5 * (2 / 3)],
         [2 / 3, 2 / 3, -1 / 3], 0.3, 0.9, True, "robust",
         "axis a = (2,-1,2)/3, e = (2,2,-1)/3 is a unit vector orthogonal to a; p = c + 6 e + 5 a lies on the cylinder of "
         "radius 6 with outward normal e")
add_case("cyl_general_wrong_radius", "cylinder", [2 / 3, -1 / 3, 2 / 3, 10, 20, 30, 6.5], True,
         [10 + 6 * (2 / 3) + 5 * (2 / 3), 20 + 6 * (2 / 3) - 5 * (1 / 3), 30 - 6 * (1 / 3) + 5 * (2 / 3)],
         [2 / 3, 2 / 3, -1 / 3], 0.3, 0.9, False, "robust", "same point, R = 6.5: |6 - 6.5| = 0.5 > 0.3")

# ---- cone: apex at the origin, axis +z, opang = pi/2 (half angle 45 deg); dist = (h - rho) / sqrt(2)
r2 = math.sqrt(0.5)
KO = [0, 0, 0, 0, 0, 1, math.pi / 2]
c10 = math.cos(math.radians(10))
add_case("cone_above_surface_in", "cone", KO, True, [3, 0, 4], [r2, 0, -r2], 0.75, c10, True, "robust",
         "cone.jl:68-85: h = 4, rho = 3, |dist| = |h sin45 - rho cos45| = 0.7071 < 0.75; surface normal e_rho cos45 - a sin45 = n")
add_case("cone_above_surface_out", "cone", KO, True, [3, 0, 4], [r2, 0, -r2], 0.70, c10, False, "robust", "0.7071 > 0.70")
add_case("cone_below_surface_abs", "cone", KO, True, [4, 0, 3], [r2, 0, -r2], 0.75, c10, True, "robust",
         "dist = -0.7071: compatiblesCone takes abs(dist) (cone.jl:148)")
add_case("cone_inward_accepts_inward_normal", "cone", KO, False, [3, 0, 3], [-r2, 0, r2], 0.1, c10, True, "robust",
         "point on the surface (dist 0), outwards = false: dot(-current_normal, n) = 1")
add_case("cone_inward_rejects_outward_normal", "cone", KO, False, [3, 0, 3], [r2, 0, -r2], 0.1, c10, False, "robust", "dot = -1")
add_case("cone_mirror_nappe_rejected", "cone", KO, True, [3, 0, -3], [r2, 0, r2], 0.75, c10, False, "robust",
         "p lies on the mirror image of the cone (h = -3): the formula gives |dist| = |(-3 - 3)/sqrt2| = 4.24, no second nappe")
add_case("cone_normal_5deg_off", "cone", KO, True, [3, 0, 3],
         [math.cos(math.radians(50)), 0, -math.sin(math.radians(50))], 0.1, c10, True, "robust",
         "n = surface normal turned by 5 deg in the meridian plane: dot = cos 5deg = 0.996 > cos 10deg = 0.985")
add_case("cone_normal_15deg_off", "cone", KO, True, [3, 0, 3],
         [math.cos(math.radians(60)), 0, -math.sin(math.radians(60))], 0.1, c10, False, "robust", "dot = cos 15deg = 0.966 < 0.985")
add_case("cone_opang_is_full_angle", "cone", [0, 0, 0, 0, 0, 1, math.pi / 3], True, [6 * math.tan(math.pi / 6), 0, 6],
         [math.cos(math.pi / 6), 0, -math.sin(math.pi / 6)], 0.01, c10, True, "robust",
         "cone.jl:76 rotates by -opang/2: with opang = 60 deg the surface is at rho = h tan 30deg.  Read as a half angle the "
         "surface would be at rho = h tan 60deg and this point 3.46 away")
add_case("cone_point_on_axis", "cone", KO, True, [0, 0, 4], [1, 0, 0], 10.0, -1.0, False, "exact",
         "to_pointn = (0,0,-1); cross(axis, to_pointn) = 0, normalize(0) = NaN: every comparison false even with eps = 10, cos_alpha = -1")
add_case("cone_point_at_apex", "cone", KO, True, [0, 0, 0], [1, 0, 0], 10.0, -1.0, False, "exact",
         "to_point = 0, normalize(0) = NaN")
add_case("cone_general_in", "cone", [1, 2, 3, 2 / 3, -1 / 3, 2 / 3, 2 * math.atan(0.75)], True,
         [1 + 8 * (2 / 3) + 6 * (2 / 3), 2 - 8 * (1 / 3) + 6 * (2 / 3), 3 + 8 * (2 / 3) - 6 * (1 / 3)],
         [0.8 * (2 / 3) - 0.6 * (2 / 3), 0.8 * (2 / 3) + 0.6 * (1 / 3), -0.8 * (1 / 3) - 0.6 * (2 / 3)], 0.05, c10, True, "robust",
         "axis a = (2,-1,2)/3, e = (2,2,-1)/3 orthogonal to it, half angle atan(3/4) (cos 0.8, sin 0.6): p = apex + 8 a + 6 e has h = 8, rho = 6 "
         "= h tan(half): on the surface; outward normal = 0.8 e - 0.6 a = n")
add_case("cone_general_off", "cone", [1, 2, 3, 2 / 3, -1 / 3, 2 / 3, 2 * math.atan(0.75)], True,
         [1 + 8 * (2 / 3) + 7 * (2 / 3), 2 - 8 * (1 / 3) + 7 * (2 / 3), 3 + 8 * (2 / 3) - 7 * (1 / 3)],
         [0.8 * (2 / 3) - 0.6 * (2 / 3), 0.8 * (2 / 3) + 0.6 * (1 / 3), -0.8 * (1 / 3) - 0.6 * (2 / 3)], 0.5, c10, False, "robust",
         "rho = 7: |h sin - rho cos| = |4.8 - 5.6| = 0.8 > 0.5")

# ---- verify ------------------------------------------------------------------------------------
for c in cases:
    if c["mode"] == "exact":
        sh = {"kind": c["kind"], "outwards": c["outwards"], "v": [Fr(x) for x in c["v"]]}
        got = EXACT[c["kind"]](sh, [Fr(x) for x in c["p"]], [Fr(x) for x in c["n"]], Fr(c["eps"]), Fr(c["cos_alpha"]))
        assert got == c["expect"], (c["name"], got)
    else:
        dv, av = robust_values(c, c["p"], c["n"])
        md, ma = dv - c["eps"], av - c["cos_alpha"]
        exact_d = c["mode"] == "exact_distance"
        assert (md < 0 and ma > 0) == c["expect"] or exact_d, (c["name"], dv, av)
        if exact_d:   # the distance side sits exactly at / 2^-50 below eps (Pythagorean radius): checked in rationals
            v = [Fr(x) for x in c["v"]]
            p = [Fr(x) for x in c["p"]]
            if c["kind"] == "sphere":
                nr = norm(vsub(p, v[0:3])); R = v[3]
            else:
                a, cc = v[0:3], v[3:6]
                sd = dot(a, vsub(p, cc))
                nr = norm(vsub(vsub(p, [mul(x, sd) for x in a]), cc)); R = v[6]
            assert lt(fabs(sub(nr, R)), Fr(c["eps"])) == c["expect"] and ma > 0.4, c["name"]
        else:   # both sides keep a margin no rounding order can cross
            assert abs(md) > 1e-3 and abs(ma) > 1e-3, (c["name"], md, ma)

# ---- cloud-level vectors: scorecandidate and the enabled bits (Q4) ------------------------------
score_cases = [{
    "name": "q4_sphere_ignores_enabled",
    "derivation": "sphere.jl:121,131: `ens` is built and never used -> a disabled compatible point still counts; "
                  "plane.jl:66 / cylinder.jl:177 / cone.jl:161 apply `cp .& ens`.  Three points, all compatible with all four "
                  "shapes below are NOT needed: every shape gets its own compatible pair, point 2 of each pair is disabled.",
    "eps": E, "cos_alpha": H,
    "points": [[5, -7, 3], [6, -7, 3], [1, 1, 5], [1, 5, 1], [1, 5, 9], [5, 1, 9]],
    "normals": [[0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 1, 0], [0, 1, 0], [1, 0, 0]],
    "disabled_1based": [2, 4, 6],
    "shapes": [
        {"kind": "plane", "v": PL, "outwards": True, "compatible_1based": [1, 2],
         "count_reference": 1, "count_fixed": 1},
        {"kind": "sphere", "v": SPC + [4.0], "outwards": True, "compatible_1based": [3, 4],
         "count_reference": 2, "count_fixed": 1},
        {"kind": "cylinder", "v": CY + [4.0], "outwards": True, "compatible_1based": [4, 5, 6],
         "count_reference": 1, "count_fixed": 1},
    ],
}]

# ---- fit-level vector: validatecone's distance check has no abs (Q11) ---------------------------
# cone through three points at h = rho (half angle 45 deg about +z, apex at the origin) with their exact surface
# normals; a 4th point (drawN = 4) far INSIDE the cone has dist = (h - rho)/sqrt2 = +1.77 > eps -> rejected, a 4th
# point far OUTSIDE has dist = -1.77: `calcs[i][1] > eps` (cone.jl:93) is false, so the fit is accepted although
# |dist| = 1.77 >> eps = 0.3.
def cone_pt(th, h, rho):
    return [rho * math.cos(th), rho * math.sin(th), h]


def cone_nrm(th):
    return [r2 * math.cos(th), r2 * math.sin(th), -r2]


ths = [0.3, 2.1, 4.4]
fit_cases = [{
    "name": "q11_validatecone_no_abs",
    "derivation": "cone.jl:89-96: only dist > eps rejects.  4th point at (rho, h) = (3, 0.5): dist = (0.5 - 3)/sqrt2 = -1.77 -> "
                  "accepted; at (0.5, 3): dist = +1.77 -> rejected",
    "eps": 0.3, "alpha_deg": 5.0,
    "p3": [cone_pt(t, 5 + i, 5 + i) for i, t in enumerate(ths)], "n3": [cone_nrm(t) for t in ths],
    "p4_outside": cone_pt(1.0, 0.5, 3.0), "p4_inside": cone_pt(1.0, 3.0, 0.5), "n4": cone_nrm(1.0),
    "expect_fit_with_outside": True, "expect_fit_with_inside": False,
    "expect_cone": {"apex": [0, 0, 0], "axis": [0, 0, 1], "opang": math.pi / 2, "rel_tol": 1e-9},
}]

json.dump({"_doc": "hand-derived known-answer vectors; generator + derivations: tests/golden/make_compat_exact_vectors.py",
           "compat": cases, "score": score_cases, "fit": fit_cases}, open(OUT, "w"), indent=1)
print(len(cases), "compat vectors (%d exact)" % sum(c["mode"] == "exact" for c in cases))
