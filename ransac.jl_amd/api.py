"""Host-side mirror of the reference's API for the hot path -- same names, argument meaning
and error behaviour as cserteGT3/RANSAC.jl v0.6.0 (paths in comments are under
/root/reference/src), with every data-parallel step routed through the C ABI of
libransac_hip.so.  Nothing here computes a score or an inlier set on the CPU."""
import ctypes as C
import math

import numpy as np

from . import _lib as L
from ._lib import CONE, CYLINDER, PLANE, SPHERE, check, lib


# ----------------------------------------------------------------- options ----
def _opt_value(key, value):
    if value is None:
        return L.OPTION_UNSET
    if isinstance(value, str):
        table = {"score_path": L.SCORE_PATH, "refit_path": L.REFIT_PATH}.get(key)
        if table is None or value not in table:
            raise ValueError("option %r has no value %r" % (key, value))
        return table[value]
    return int(value)


def set_option(key, value, cloud=None):
    """rh_set_option (include/ransac_hip.h): a tuning option for one cloud or, with cloud=None, process-wide.  value: an
    int, one of the names of score_path ("auto" / "brute" / "groups") and refit_path ("auto" / "scan" / "culled"), or None
    to clear the setting.  The library itself reads no environment variable."""
    check(lib().rh_set_option(None if cloud is None else cloud._h, key.encode(), _opt_value(key, value)))


def get_option(key, cloud=None):
    """the option's value as the library sees it for this cloud, or None when nobody has set it"""
    v, st = C.c_int64(), C.c_int32()
    check(lib().rh_get_option(None if cloud is None else cloud._h, key.encode(), C.byref(v), C.byref(st)))
    return v.value if st.value else None


class option:
    """with option("refit_path", "scan"): ...   -- set, then put back what was there"""

    def __init__(self, key, value, cloud=None):
        self.key, self.value, self.cloud = key, value, cloud

    def __enter__(self):
        # what the cloud / the process itself holds (not what a cloud inherits): cleared again on exit when it held nothing
        self.old = get_option(self.key, self.cloud) if self.cloud is None else None
        set_option(self.key, self.value, self.cloud)
        return self

    def __exit__(self, *exc):
        set_option(self.key, self.old, self.cloud)
        return False


# ------------------------------------------------------------------ shapes ----
class FittedShape:
    """abstract supertype of all fitted shapes (fitting.jl:6)"""
    kind = None

    def to_c(self):
        raise NotImplementedError


def _vec(a):
    a = np.asarray(a, dtype=np.float64).reshape(3)
    return a


class FittedPlane(FittedShape):  # shapes/plane.jl:8-11
    kind = PLANE

    def __init__(self, point, normal):
        self.point, self.normal = _vec(point), _vec(normal)

    def to_c(self):
        return _mk(PLANE, False, list(self.point) + list(self.normal))

    def __repr__(self):
        return "FittedPlane(normal: %s, point: %s)" % (self.normal, self.point)


class FittedSphere(FittedShape):  # shapes/sphere.jl:9-13
    kind = SPHERE

    def __init__(self, center, radius, outwards):
        self.center, self.radius, self.outwards = _vec(center), float(radius), bool(outwards)

    def to_c(self):
        return _mk(SPHERE, self.outwards, list(self.center) + [self.radius])

    def __repr__(self):
        return "FittedSphere(center: %s, R: %s, %s)" % (self.center, self.radius, "outwards" if self.outwards else "inwards")


class FittedCylinder(FittedShape):  # shapes/cylinder.jl:11-16
    kind = CYLINDER

    def __init__(self, axis, center, radius, outwards):
        self.axis, self.center, self.radius, self.outwards = _vec(axis), _vec(center), float(radius), bool(outwards)

    def to_c(self):
        return _mk(CYLINDER, self.outwards, list(self.axis) + list(self.center) + [self.radius])

    def __repr__(self):
        return "FittedCylinder(center: %s, axis: %s, R: %s, %s)" % (
            self.center, self.axis, self.radius, "outwards" if self.outwards else "inwards")


class FittedCone(FittedShape):  # shapes/cone.jl:11-19
    kind = CONE

    def __init__(self, apex, axis, opang, outwards):
        self.apex, self.axis, self.opang, self.outwards = _vec(apex), _vec(axis), float(opang), bool(outwards)

    def to_c(self):
        return _mk(CONE, self.outwards, list(self.apex) + list(self.axis) + [self.opang])

    def __repr__(self):
        return "FittedCone(apex: %s, axis: %s, w: %s, %s)" % (
            self.apex, self.axis, self.opang, "outwards" if self.outwards else "inwards")


def _mk(kind, outwards, v):
    s = L.Shape()
    s.kind = kind
    s.outwards = int(bool(outwards))
    for i, x in enumerate(v):
        s.v[i] = float(x)
    lib().rh_shape_finalize(C.byref(s))
    return s


def shape_from_c(s):
    v = list(s.v)
    if s.kind == PLANE:
        return FittedPlane(v[0:3], v[3:6])
    if s.kind == SPHERE:
        return FittedSphere(v[0:3], v[3], bool(s.outwards))
    if s.kind == CYLINDER:
        return FittedCylinder(v[0:3], v[3:6], v[6], bool(s.outwards))
    if s.kind == CONE:
        return FittedCone(v[0:3], v[3:6], v[6], bool(s.outwards))
    raise ValueError("unknown kind %d" % s.kind)


def strt(x):  # fitting.jl:66; shapes/*.jl `strt`
    return L.KIND_NAMES[x.kind]


DEFAULT_SHAPE_DICT = {"plane": FittedPlane, "cone": FittedCone, "cylinder": FittedCylinder, "sphere": FittedSphere}
_KIND_OF = {FittedPlane: PLANE, FittedSphere: SPHERE, FittedCylinder: CYLINDER, FittedCone: CONE}


class ExtractedShape:  # fitting.jl:81-84
    def __init__(self, shape, inpoints):
        self.shape = shape
        self.inpoints = np.asarray(inpoints, dtype=np.int64)

    def __repr__(self):
        return "Cand: (%s), %d ps" % (strt(self.shape), len(self.inpoints))


class ConfidenceInterval:  # confidenceintervals.jl:1-6
    def __init__(self, x, y):
        if x > y:
            raise ValueError("out of order")
        self.min, self.max, self.E = float(x), float(y), (float(x) + float(y)) / 2

    def __repr__(self):
        return "CI: [%s, %s]" % (self.min, self.max)


def E(x):  # confidenceintervals.jl:43
    return x.E


def notsoconfident(x, y):  # confidenceintervals.jl:20-22
    return ConfidenceInterval(min(x, y), max(x, y))


def estimatescore(S1length, Plength, sigma, score_mode=L.SCORE_INT64_WRAP):  # confidenceintervals.jl:71-74
    lo, hi, e = C.c_double(), C.c_double(), C.c_double()
    check(lib().rh_estimatescore(int(S1length), int(Plength), int(sigma), score_mode,
                                 C.byref(lo), C.byref(hi), C.byref(e)))
    ci = ConfidenceInterval.__new__(ConfidenceInterval)
    ci.min, ci.max, ci.E = lo.value, hi.value, e.value
    return ci


def prob(n, s, N, k):  # utilities.jl:262
    return lib().rh_prob(float(n), int(s), int(N), int(k))


# -------------------------------------------------------------- parameters ----
def defaultshapeparameters(T):  # shapes/*.jl defaultshapeparameters
    a = math.radians(5)
    if T is FittedPlane:
        return {"plane": {"ϵ": 0.3, "α": a}}
    if T is FittedSphere:
        return {"sphere": {"ϵ": 0.3, "α": a, "sphere_par": 0.02}}
    if T is FittedCylinder:
        return {"cylinder": {"ϵ": 0.3, "α": a}}
    if T is FittedCone:
        return {"cone": {"ϵ": 0.3, "α": a, "minconeopang": math.radians(2)}}
    raise TypeError(T)


def defaultiterationparameters(shape_types):  # utilities.jl:332-347
    return {"iteration": {"drawN": 3, "minsubsetN": 15, "prob_det": 0.9, "shape_types": list(shape_types),
                          "τ": 900, "itermax": 1000, "extract_s": "nofminset", "terminate_s": "nofminset"}}


def defaultcommonparameters():  # utilities.jl:368-373
    return {"common": {"collin_threshold": 0.2, "parallelthrdeg": 1.0}}


def defaultparameters(shape_types):  # utilities.jl:391-399
    p = defaultiterationparameters(shape_types)
    p.update(defaultcommonparameters())
    for T in shape_types:
        p.update(defaultshapeparameters(T))
    return p


DEFAULT_PARAMETERS = defaultparameters([FittedPlane, FittedCone, FittedCylinder, FittedSphere])  # RANSAC.jl:94

_ALIASES = {"eps": "ϵ", "epsilon": "ϵ", "alpha": "α", "tau": "τ"}


def _norm_keys(d):
    return {_ALIASES.get(k, k): v for k, v in d.items()}


def ransacparameters(p=None, **kwargs):  # utilities.jl:425-464
    """ransacparameters(; kw...), ransacparameters(p; kw...) or ransacparameters([types]; kw...).
    Nested dicts stand in for NamedTuples; `eps`/`alpha`/`tau` are accepted for ϵ/α/τ."""
    if p is None:
        base = DEFAULT_PARAMETERS
    elif isinstance(p, (list, tuple)):
        base = defaultparameters(list(p))
    else:
        base = p
    newp = {k: dict(v) for k, v in base.items()}
    for a, val in kwargs.items():
        old = newp.get(a, {})
        merged = dict(old)
        merged.update(_norm_keys(val))
        newp[a] = merged
    return newp


_S = {"lengthC": L.S_LENGTHC, "allcand": L.S_ALLCAND, "nofminset": L.S_NOFMINSET}


def params_to_c(params, score_mode=L.SCORE_INT64_WRAP, sphere_uses_enabled=False, sampling_streams=0,
                octree_sampling=False, octree_max_depth=10):
    """Flatten the nested parameter dict into rh_params.  Shapes absent from the dict keep the
    library defaults (they are never used: `shape_types` selects what is fitted)."""
    c = L.Params()
    lib().rh_default_params(C.byref(c))
    for name, kind in (("plane", PLANE), ("sphere", SPHERE), ("cylinder", CYLINDER), ("cone", CONE)):
        if name in params:
            sp = _norm_keys(params[name])
            c.eps[kind] = sp.get("ϵ", c.eps[kind])
            c.alpha[kind] = sp.get("α", c.alpha[kind])
    if "sphere" in params:
        c.sphere_par = _norm_keys(params["sphere"]).get("sphere_par", c.sphere_par)
    if "cone" in params:
        c.minconeopang = _norm_keys(params["cone"]).get("minconeopang", c.minconeopang)
    if "common" in params:
        c.collin_threshold = params["common"].get("collin_threshold", c.collin_threshold)
        c.parallelthrdeg = params["common"].get("parallelthrdeg", c.parallelthrdeg)
    if "iteration" in params:
        it = _norm_keys(params["iteration"])
        c.drawN = it.get("drawN", c.drawN)
        c.minsubsetN = it.get("minsubsetN", c.minsubsetN)
        c.prob_det = it.get("prob_det", c.prob_det)
        c.tau = int(it.get("τ", c.tau))
        c.itermax = int(it.get("itermax", c.itermax))
        c.extract_s = _S[str(it.get("extract_s", "nofminset")).lstrip(":")]
        c.terminate_s = _S[str(it.get("terminate_s", "nofminset")).lstrip(":")]
        if "shape_types" in it:
            st = it["shape_types"]
            c.n_shape_types = len(st)
            for i, T in enumerate(st):
                c.shape_types[i] = _KIND_OF[T] if T in _KIND_OF else int(T)
    c.score_mode = score_mode
    c.sphere_uses_enabled = int(bool(sphere_uses_enabled))
    c.sampling_streams = int(sampling_streams)
    c.octree_sampling = int(bool(octree_sampling))
    c.octree_max_depth = int(octree_max_depth)
    lib().rh_params_finalize(C.byref(c))
    return c


def _cparams(params):
    return params if isinstance(params, L.Params) else params_to_c(params)


# ------------------------------------------------------------------ octree ----
class OctreeCell:
    """A cell of the reference's octree (RegionTrees.Cell with OctreeNode data: octree.jl:147-150): `.data.incellpoints`
    (1-based indices, a numpy array), `.data.depth` (root = 1), `.boundary` = (origin, widths), `.children`, `.parent`."""

    class _Data:
        def __init__(self, cell):
            self._c = cell

        @property
        def depth(self):
            return self._c._info()[2]

        @property
        def incellpoints(self):
            n = self._c._info()[5]
            out = np.zeros(max(1, n), dtype=np.int64)
            check(lib().rh_octree_node_points(self._c._t._h, self._c.index, _p(out, C.c_int64), n))
            return out[:n]

    def __init__(self, tree, index):
        self._t, self.index = tree, int(index)
        self.data = OctreeCell._Data(self)

    def _info(self):
        o, w = np.zeros(3), np.zeros(3)
        d, par, npts = C.c_int32(), C.c_int32(), C.c_int64()
        ch = (C.c_int32 * 8)()
        check(lib().rh_octree_node_info(self._t._h, self.index, _p(o, C.c_double), _p(w, C.c_double), C.byref(d), C.byref(par), ch, C.byref(npts)))
        return o, w, d.value, par.value, list(ch), npts.value

    @property
    def boundary(self):
        o, w = self._info()[:2]
        return o, w

    @property
    def parent(self):
        par = self._info()[3]
        return None if par < 0 else OctreeCell(self._t, par)

    @property
    def children(self):
        ch = self._info()[4]
        return None if ch[0] < 0 else [OctreeCell(self._t, k) for k in ch]

    def isleaf(self):
        return self._info()[4][0] < 0

    def __eq__(self, other):
        return isinstance(other, OctreeCell) and other._t is self._t and other.index == self.index

    def __hash__(self):
        return hash((id(self._t), self.index))

    def __repr__(self):
        i = self._info()
        return "OctreeNode: %d ps, %d d" % (i[5], i[2])


class _Octree:
    def __init__(self, vertices):
        self._h = C.c_void_p()
        v = np.asarray(vertices)
        if v.dtype == np.float32:   # a Float32 cloud's tree: the geometry in binary32, as the reference computes it there
            self.vertices = np.ascontiguousarray(v, dtype=np.float32).reshape(-1, 3)
            check(lib().rh_octree_build_f32(_p(self.vertices, C.c_float), self.vertices.shape[0], C.byref(self._h)))
            return
        self.vertices = _f64(vertices).reshape(-1, 3)
        check(lib().rh_octree_build(_p(self.vertices, C.c_double), self.vertices.shape[0], C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().rh_octree_destroy(h)
            except (AttributeError, TypeError):
                pass
            self._h = None


def buildoctree(vertices):
    """buildoctree(vertices) -> the root Cell (octree.jl:237-244).  float32 vertices: the binary32 tree (rh_octree_build_f32)."""
    return OctreeCell(_Octree(vertices), 0)


def octreedepth(pc_or_cell):
    """octreedepth(pc) / octreedepth(cell) (octree.jl:212-230): the depth of the deepest leaf."""
    cell = pc_or_cell.octree if hasattr(pc_or_cell, "octree") else pc_or_cell
    d = C.c_int32()
    check(lib().rh_octree_info(cell._t._h, None, C.byref(d), None))
    return d.value


def findleaf(cell, p):
    """findleaf(pc.octree, p) (RegionTrees; fitting.jl:397)."""
    q = _f64(p).reshape(3)
    out = C.c_int32()
    check(lib().rh_octree_findleaf(cell._t._h, _p(q, C.c_double), C.byref(out)))
    return OctreeCell(cell._t, out.value)


def getnthcell(c, n):
    """getnthcell(c, n) (octree.jl:11-22): the ancestor of c (or c) at depth n, None if there is none."""
    out = C.c_int32()
    check(lib().rh_octree_getnthcell(c._t._h, c.index, int(n), C.byref(out)))
    return None if out.value < 0 else OctreeCell(c._t, out.value)


def iswithinrectangle(rect, p):
    """iswithinrectangle(rect, p) (octree.jl:187-196); rect = (origin, widths): vmin < p <= vmax on every axis."""
    o, w = (np.asarray(rect[0], dtype=np.float64), np.asarray(rect[1], dtype=np.float64))
    q = np.asarray(p, dtype=np.float64)
    return bool(np.all(o < q) and np.all(o + w >= q))


def cell_enabled_points(pc, cell):
    """cell.data.incellpoints[pc.isenabled[cell.data.incellpoints]] (fitting.jl:405-407), gathered on the device."""
    cap = cell._info()[5]
    out = np.zeros(max(1, cap), dtype=np.int64)
    n = C.c_int64()
    check(lib().rh_octree_cell_enabled(pc._h, cell._t._h, cell.index, _p(out, C.c_int64), cap, C.byref(n)))
    return out[: n.value]


# ------------------------------------------------------------------- cloud ----
def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class RANSACCloud:
    """RANSACCloud(vertices, normals, numofsubsets | subsets) (octree.jl:37-138).  The points,
    normals, subset 1 and the enabled bits live in HBM on `device`; the host keeps the arrays
    it was given (for the O(1) minimal-set fits) and the subset index lists."""

    def __init__(self, vertices, normals, subsets, device=0, seed=None, force_eltype=None):
        """force_eltype = numpy.float32: a Float32 cloud (octree.jl:102-109) -- scoring and refit then compute in
        binary32 like the reference does on such a cloud, and so does ransac() (fits included, all four kinds); refit_lsq stays Float64-only."""
        self.is_f32 = force_eltype is not None and np.dtype(force_eltype) == np.float32
        if force_eltype is not None and not self.is_f32 and np.dtype(force_eltype) != np.float64:
            raise ValueError("force_eltype must be float32 or float64")
        if self.is_f32:
            self.vertices32 = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
            self.normals32 = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
            vertices, normals = self.vertices32.astype(np.float64), self.normals32.astype(np.float64)
        self.vertices = _f64(vertices).reshape(-1, 3)
        self.normals = _f64(normals).reshape(-1, 3)
        assert self.vertices.shape == self.normals.shape, "Every point must have a normal."
        self.size = self.vertices.shape[0]
        if isinstance(subsets, (int, np.integer)):
            assert subsets > 0, "At least 1 subset please!"
            rng = np.random.default_rng(seed)
            alls = rng.permutation(self.size).astype(np.int64) + 1   # randperm(l): octree.jl:131
            ssl = self.size // int(subsets)
            subs = [alls[i * ssl:(i + 1) * ssl] for i in range(int(subsets) - 1)]
            subs.append(alls[(int(subsets) - 1) * ssl:])
            subsets = subs
        self.subsets = [np.ascontiguousarray(s, dtype=np.int64) for s in subsets]
        self.device = device
        h = C.c_void_p()
        s1 = self.subsets[0]
        if self.is_f32:
            check(lib().rh_cloud_create_f32(_p(self.vertices32, C.c_float), _p(self.normals32, C.c_float), self.size,
                                            _p(s1, C.c_int64), s1.size, device, C.byref(h)))
        else:
            check(lib().rh_cloud_create(_p(self.vertices, C.c_double), _p(self.normals, C.c_double), self.size,
                                        _p(s1, C.c_int64), s1.size, device, C.byref(h)))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().rh_cloud_destroy(h)
            except (AttributeError, TypeError):
                pass   # interpreter shutdown: the module globals are gone, the process frees the device memory
            self._h = None

    @property
    def nchunks(self):
        return (self.size + 63) // 64

    @property
    def octree(self):
        """pc.octree (octree.jl:47, built by the constructor there; here on first use: rh_ransac never reads it)"""
        if getattr(self, "_octree", None) is None:
            self._octree = buildoctree(self.vertices32 if self.is_f32 else self.vertices)   # (a Float32 cloud: the binary32 tree)
        return self._octree

    @property
    def isenabled(self):
        """pc.isenabled as a bool array (a copy; write it back with set_enabled)."""
        ch = np.zeros(max(1, self.nchunks), dtype=np.uint64)
        check(lib().rh_cloud_get_enabled(self._h, _p(ch, C.c_uint64), self.nchunks))
        bits = np.unpackbits(ch[: self.nchunks].view(np.uint8), bitorder="little")
        return bits[: self.size].astype(bool)

    def enabled_chunks(self):
        ch = np.zeros(max(1, self.nchunks), dtype=np.uint64)
        check(lib().rh_cloud_get_enabled(self._h, _p(ch, C.c_uint64), self.nchunks))
        return ch[: self.nchunks]

    def set_enabled(self, mask_or_chunks):
        a = np.asarray(mask_or_chunks)
        if a.dtype == np.uint64:
            ch = np.ascontiguousarray(a).reshape(-1)
            if ch.size != self.nchunks:
                raise ValueError("set_enabled: %d chunks given, the cloud has %d (ceil(size / 64))" % (ch.size, self.nchunks))
        else:
            if a.size != self.size:
                raise ValueError("set_enabled: mask of length %d for a cloud of %d points" % (a.size, self.size))
            bits = np.zeros(self.nchunks * 64, dtype=np.uint8)
            bits[: self.size] = a.reshape(-1).astype(np.uint8)
            ch = np.packbits(bits, bitorder="little").view(np.uint64)
        n_given = ch.size   # the library checks this against its own word count
        ch = np.ascontiguousarray(ch if ch.size else np.zeros(1, dtype=np.uint64))
        check(lib().rh_cloud_set_enabled(self._h, _p(ch, C.c_uint64), n_given))

    def enable_all(self):
        check(lib().rh_cloud_enable_all(self._h))

    def set_stream(self, hip_stream):
        """Enqueue this cloud's work on the caller's HIP stream (an int handle, e.g.
        torch.cuda.current_stream().cuda_stream; None = back to the cloud's own stream)."""
        check(lib().rh_cloud_set_stream(self._h, C.c_void_p(hip_stream or 0), 0 if hip_stream is None else 1))

    def count_enabled(self):
        out = C.c_int64()
        check(lib().rh_cloud_count_enabled(self._h, C.byref(out)))
        return out.value

    def __repr__(self):
        return "RANSACCloud of size %d & %d subsets" % (self.size, len(self.subsets))


# ---------------------------------------------------------------- hot path ----
def fit(T, p, n, pc, params):
    """fit(::Type{T}, p, n, pc, params) -> T or None (shapes/*.jl `fit`).  Float32 points (a Float32 cloud's, or numpy
    float32 arrays) are fitted in Float32 like Julia fits SVector{3,Float32}s (rh_fit_f32)."""
    f32 = bool(getattr(pc, "is_f32", False)) or (getattr(p, "dtype", None) == np.float32 and getattr(n, "dtype", None) == np.float32)
    if f32:
        p, n = np.asarray(p, dtype=np.float32), np.asarray(n, dtype=np.float32)
    p, n = _f64(p).reshape(-1, 3), _f64(n).reshape(-1, 3)
    assert p.shape[0] > 2, "At least 3 point is needed."
    assert p.shape == n.shape, "Size must be the same."
    out, ok = L.Shape(), C.c_int32()
    kind = _KIND_OF[T] if T in _KIND_OF else int(T)
    check((lib().rh_fit_f32 if f32 else lib().rh_fit)(kind, _p(p, C.c_double), _p(n, C.c_double), p.shape[0], C.byref(_cparams(params)),
                                                     C.byref(out), C.byref(ok)))
    return shape_from_c(out) if ok.value else None


def _shape_array(cands, f32=False):
    arr = (L.Shape * max(1, len(cands)))()
    for i, s in enumerate(cands):
        arr[i] = s if isinstance(s, L.Shape) else s.to_c()
        if f32:
            lib().rh_shape_finalize_f32(C.byref(arr[i]))
    return arr


def shape_f32(s):
    """The Float32 form of a shape (what `fit` returns on a Float32 cloud): fields rounded to binary32, a cone's
    cos / sin of -opang/2 as binary32 (rh_shape_finalize_f32).  Returns the C record."""
    cs = L.Shape.from_buffer_copy(bytes(s if isinstance(s, L.Shape) else s.to_c()))
    lib().rh_shape_finalize_f32(C.byref(cs))
    return cs


def score_batch(pc, candidates, params, want_masks=False):
    """One launch for the whole batch, all shape kinds (replaces the loop of scorecandidates!,
    fitting.jl:181-190).  Returns counts[b] (and masks[b, ceil(S/64)] over subset positions)."""
    b = len(candidates)
    arr = candidates if isinstance(candidates, C.Array) else _shape_array(candidates, getattr(pc, "is_f32", False))
    counts = np.zeros(max(1, b), dtype=np.int32)
    w = (pc.subsets[0].size + 63) // 64
    masks = np.zeros((max(1, b), max(1, w)), dtype=np.uint64) if want_masks else None
    check(lib().rh_score_batch(pc._h, arr, b, C.byref(_cparams(params)), _p(counts, C.c_int32),
                               _p(masks, C.c_uint64) if want_masks else None))
    if want_masks:
        if w == 0:
            return counts[:b], masks[:b, :0]
        flat = masks.reshape(-1)[: b * w].reshape(b, w) if masks.shape[1] != w else masks[:b]
        return counts[:b], flat
    return counts[:b]


def scorecandidate(pc, candidate, subsetID, params):
    """scorecandidate(pc, candidate, subsetID, params) -> (ConfidenceInterval, inpoints)
    (shapes/plane.jl:61-71 ...).  Only subset 1 lives on the device, as only subset 1 is ever
    scored by the reference (iterations.jl:95)."""
    if subsetID != 1:
        raise ValueError("only subsetID == 1 is resident on the device (iterations.jl:95)")
    cp = _cparams(params)
    counts, masks = score_batch(pc, [candidate], cp, want_masks=True)
    s1 = pc.subsets[0]
    bits = np.unpackbits(masks[0].view(np.uint8), bitorder="little")[: s1.size].astype(bool)
    inpoints = s1[bits]
    assert inpoints.size == counts[0]
    return estimatescore(s1.size, pc.size, int(counts[0]), cp.score_mode), inpoints


def push2candidatesandlevels(candidates, candidate, levels, current_level):  # utilities.jl:473-481
    """Push `candidate` (one shape or a list of shapes) and its octree level."""
    if isinstance(candidate, (list, tuple)):
        candidates.extend(candidate)
        levels.extend([current_level] * len(candidate))
    else:
        candidates.append(candidate)
        levels.append(current_level)


def forcefitshapes(points, normals, parameters, candidates, level_array, octree_lev, pc):  # forcefitshapes!: fitting.jl:165-173
    """Call `fit` for every type of iteration.shape_types, in that order, and append what fits."""
    for T in parameters["iteration"]["shape_types"]:
        fitted = fit(T, points, normals, pc, parameters)
        if fitted is not None:
            push2candidatesandlevels(candidates, fitted, level_array, octree_lev)


def findAABB(points):  # utilities.jl:125-136
    """(min, max) corners of the axis-aligned box of the points (host helper; the device builds its own boxes)."""
    a = _f64(points)
    return a.min(axis=0), a.max(axis=0)


def smallestdistance(points):  # utilities.jl:187-199 (exported, unused by ransac(): iterations.jl:57 is a comment)
    a = _f64(points)
    assert a.shape[0] > 1, "At least two point is needed for that."
    d = np.linalg.norm(a[:, None, :] - a[None, :, :], axis=2)
    return float(d[~np.eye(a.shape[0], dtype=bool)].min())


def setfloattype(nt, T):  # utilities.jl:488-504
    """Convert every real-but-not-integer value of a nested parameter dict to numpy type T."""
    out = {}
    for k, v in nt.items():
        if isinstance(v, dict):
            out[k] = setfloattype(v, T)
        elif isinstance(v, (float, np.floating)) and not isinstance(v, bool):
            out[k] = T(v)
        else:
            out[k] = v
    return out


class IterationCandidates:  # fitting.jl:94-131
    """Struct of arrays of the scored candidates: shapes, scores (ConfidenceInterval), inpoints."""

    def __init__(self):
        self.shapes, self.scores, self.inpoints = [], [], []

    def __len__(self):
        return len(self.shapes)

    def __repr__(self):
        return "IterationCandidates: %d candidates" % len(self)


def recordscore(ic, shape, score, inpoints):  # recordscore!: fitting.jl:114-119
    ic.shapes.append(shape)
    ic.scores.append(score)
    ic.inpoints.append(inpoints)
    return ic


def deleteat(ic, arg):  # deleteat!(ic, arg): fitting.jl:126-131 (1-based index or ascending list of them)
    idx = [arg] if isinstance(arg, (int, np.integer)) else list(arg)
    for i in sorted(idx, reverse=True):
        for a in (ic.shapes, ic.scores, ic.inpoints):
            del a[i - 1]
    return ic


def findhighestscore(A):  # fitting.jl:140-158: first maximum of E (strict >), overlap flag
    if len(A) == 0:
        return {"index": 0, "overlap": False}
    ind, highest = 1, E(A.scores[0])
    for i, sc in enumerate(A.scores, start=1):
        if E(sc) > highest:
            highest, ind = E(sc), i
    best = A.scores[ind - 1]
    for i, sc in enumerate(A.scores, start=1):
        if i != ind and (sc.min <= best.max and best.min <= sc.max):   # isoverlap: confidenceintervals.jl:29-36
            return {"index": ind, "overlap": True}
    return {"index": ind, "overlap": False}


def scorecandidates(pc, iterationcandidates, candidates, subsetID, params, octree_levels=None):
    """scorecandidates! (fitting.jl:181-190) with the sequential loop replaced by ONE batched launch; the
    results are recorded in candidate order and the two input lists are emptied like the reference does.
    (`levelscore` is not kept: it never influences a result)"""
    if subsetID != 1:
        raise ValueError("only subsetID == 1 is resident on the device (iterations.jl:95)")
    if candidates:
        cp = _cparams(params)
        counts, masks = score_batch(pc, candidates, cp, want_masks=True)
        s1 = pc.subsets[0]
        for cand, cnt, m in zip(candidates, counts, masks):
            bits = np.unpackbits(m.view(np.uint8), bitorder="little")[: s1.size].astype(bool)
            recordscore(iterationcandidates, cand, estimatescore(s1.size, pc.size, int(cnt), cp.score_mode), s1[bits])
    del candidates[:]
    if octree_levels is not None:
        del octree_levels[:]


def removeinvalidshapes(pc, candidates):  # removeinvalidshapes!: fitting.jl:209-221
    en = pc.isenabled
    toremove = [i for i, ip in enumerate(candidates.inpoints, start=1) if ip.size and not en[ip - 1].all()]
    deleteat(candidates, toremove)


def refit(s, pc, params):
    """refit(s, pc, params) -> ExtractedShape (shapes/plane.jl:137-143 ...)."""
    out = np.zeros(max(1, pc.size), dtype=np.int64)
    n = C.c_int64()
    cs = s if isinstance(s, L.Shape) else (shape_f32(s) if getattr(pc, "is_f32", False) else s.to_c())
    check(lib().rh_refit(pc._h, C.byref(cs), C.byref(_cparams(params)), _p(out, C.c_int64), pc.size, C.byref(n)))
    return ExtractedShape(s if not isinstance(s, L.Shape) else shape_from_c(s), out[: n.value].copy())


def refit_lsq(s, pc, params, max_iter=10):
    """Least-squares refit of `s` to its compatible points within 3*eps (the step the reference omits,
    docs/src/ransac.md:163-168).  Returns (shape, n_used, rms, iterations)."""
    out, n, rms, it = L.Shape(), C.c_int64(), C.c_double(), C.c_int32()
    cs = s if isinstance(s, L.Shape) else s.to_c()
    check(lib().rh_refit_lsq(pc._h, C.byref(cs), C.byref(_cparams(params)), max_iter, C.byref(out), C.byref(n),
                             C.byref(rms), C.byref(it)))
    return shape_from_c(out), n.value, rms.value, it.value


def invalidate_indexes(pc, indexlist):  # invalidate_indexes!: fitting.jl:197-202
    idx = np.ascontiguousarray(indexlist, dtype=np.int64)
    check(lib().rh_invalidate(pc._h, _p(idx, C.c_int64), idx.size))


def select_enabled(pc, ranks):
    """k-th enabled point of the root cell: the `enabled_inds[nexti]` of fitting.jl:405-422."""
    r = np.ascontiguousarray(ranks, dtype=np.int64)
    out = np.zeros(max(1, r.size), dtype=np.int64)
    check(lib().rh_select_enabled(pc._h, _p(r, C.c_int64), r.size, _p(out, C.c_int64)))
    return out[: r.size]


def sample_sets(pc, drawN, rng, k):
    """samplepointcloud4!(pc, ...) (fitting.jl:383-430) k times in a row on `rng` (an _lib.Rng: rh_rng_seed / an injected
    stream) -- one launch for the k minimal sets of an iteration, the generator advanced exactly as k sequential calls would
    advance it.  Returns (idx, ok, level): k x drawN point indices (1-based), the reference's two return values per set."""
    k = int(k)
    idx = np.zeros((max(1, k), int(drawN)), dtype=np.int64)
    ok = np.zeros(max(1, k), dtype=np.int32)
    lev = np.zeros(max(1, k), dtype=np.int32)
    check(lib().rh_sample_sets(pc._h, int(drawN), C.byref(rng), k, _p(idx, C.c_int64), _p(ok, C.c_int32), _p(lev, C.c_int32)))
    return idx[:k], ok[:k].astype(bool), lev[:k]


def fit_sets(pc, sets, ok, params):
    """forcefitshapes! (fitting.jl:165-173) over the minimal sets of an iteration in one call (rh_fit_sets): returns (list of
    rh_shape in candidate order, index of the set each came from)."""
    sets = np.ascontiguousarray(sets, dtype=np.int64)
    k, drawN = sets.shape
    okc = None if ok is None else np.ascontiguousarray(ok, dtype=np.int32)
    cp = _cparams(params)
    cap = max(1, k * max(1, cp.n_shape_types))
    arr = (L.Shape * cap)()
    so = np.zeros(cap, dtype=np.int32)
    n = C.c_int32()
    f32 = bool(getattr(pc, "is_f32", False))
    xyz = pc.vertices if not f32 else np.ascontiguousarray(pc.vertices32, dtype=np.float64)
    nrm = pc.normals if not f32 else np.ascontiguousarray(pc.normals32, dtype=np.float64)
    check(lib().rh_fit_sets(_p(xyz, C.c_double), _p(nrm, C.c_double), _p(sets, C.c_int64), None if okc is None else _p(okc, C.c_int32),
                            k, drawN, C.byref(cp), 1 if f32 else 0, arr, _p(so, C.c_int32), cap, C.byref(n)))
    return [arr[i] for i in range(n.value)], so[: n.value]


class _ResultOwner:
    """Keeps an rh_result (and with it the pinned block of index lists) alive; frees it on collection."""

    def __init__(self, res):
        self.res = res

    def __del__(self):
        try:
            lib().rh_result_free(C.byref(self.res))
        except Exception:   # interpreter shutdown
            pass


class MpGroup:
    """The processes of one node that run ransac() on ONE scene together (rh_mp, include/ransac_hip.h): one process per
    GPU, each with a replica of the cloud.  Collective constructor: every rank passes the same name."""

    def __init__(self, name, rank, world, slot_bytes=0):
        self.rank, self.world = int(rank), int(world)
        h = C.c_void_p()
        check(lib().rh_mp_open(name.encode(), self.rank, self.world, int(slot_bytes), C.byref(h)))
        self._h = h

    def allgather(self, payload):
        """bytes of equal length from every rank -> list of `world` bytes objects, in rank order"""
        payload = bytes(payload)
        out = C.create_string_buffer(len(payload) * self.world)
        check(lib().rh_mp_allgather(self._h, payload, len(payload), out))
        return [out.raw[i * len(payload):(i + 1) * len(payload)] for i in range(self.world)]

    def close(self):
        if getattr(self, "_h", None):
            lib().rh_mp_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ransac(pc, params, setenabled=False, reset_rand=False, seed=1234, stream=None,
           score_mode=L.SCORE_INT64_WRAP, sphere_uses_enabled=False, sampling_streams=0, octree_sampling=False,
           return_stats=False, mp=None):
    """ransac(pc, params[, setenabled]; reset_rand) -> (Vector{ExtractedShape}, seconds)
    (iterations.jl:14-21, 35-162).  `reset_rand` reseeds the generator with 1234 like
    Random.seed!(1234); `stream` injects raw 64-bit draws (rand(1:n) = 1 + floor(u*n/2^64)).
    mp: an MpGroup -- the loop is then run by all its ranks together on this one scene (rh_ransac_mp: the minimal
    sets of every iteration dealt round-robin to the ranks); every rank gets the same result as a single process."""
    if setenabled:
        pc.enable_all()
    cp = params if isinstance(params, L.Params) else params_to_c(params, score_mode, sphere_uses_enabled, sampling_streams, octree_sampling)
    rng = L.Rng()
    lib().rh_rng_seed(C.byref(rng), 1234 if reset_rand else seed)
    keep = None
    if stream is not None:
        keep = np.ascontiguousarray(stream, dtype=np.uint64)
        rng.stream = _p(keep, C.c_uint64)
        rng.stream_len = keep.size
    res = L.Result()
    if mp is not None:
        check(lib().rh_ransac_mp(pc._h, _p(pc.vertices, C.c_double), _p(pc.normals, C.c_double), C.byref(cp),
                                 C.byref(rng), mp._h, C.byref(res)))
    elif getattr(pc, "is_f32", False):   # a Float32 cloud: the loop in binary32, from the Float32 arrays as they are
        check(lib().rh_ransac_f32(pc._h, _p(pc.vertices32, C.c_float), _p(pc.normals32, C.c_float), C.byref(cp),
                                  C.byref(rng), C.byref(res)))
    else:
        check(lib().rh_ransac(pc._h, _p(pc.vertices, C.c_double), _p(pc.normals, C.c_double), C.byref(cp),
                              C.byref(rng), C.byref(res)))
    # the index lists stay where rh_ransac put them (one pinned block per run): every `inpoints` is a
    # zero-copy view whose base keeps the result alive; rh_result_free runs when the last view is gone
    owner = _ResultOwner(res)
    extracted = []
    for i in range(res.n_shapes):
        e = res.shapes[i]
        if e.n_inpoints > 0:
            raw = (C.c_int64 * e.n_inpoints).from_address(C.addressof(e.inpoints.contents))
            raw._owner = owner
            idx = np.frombuffer(raw, dtype=np.int64)
        else:
            idx = np.zeros(0, dtype=np.int64)
        es = ExtractedShape(shape_from_c(e.shape), idx)
        es.score_E, es.iteration, es.c_shape = e.score_E, e.iteration, L.Shape.from_buffer_copy(bytes(e.shape))
        extracted.append(es)
    stats = {"iterations": res.iterations, "candidates_scored": res.candidates_scored,
             "scored_left": res.scored_left, "seconds": res.seconds, "seconds_score": res.seconds_score,
             "seconds_extract": res.seconds_extract, "seconds_host": res.seconds_host,
             "seconds_to_last_extraction": res.seconds_to_last_extraction, "draws": rng.draws}
    del owner
    seconds = stats["seconds"]
    return (extracted, seconds, stats) if return_stats else (extracted, seconds)


# ------------------------------------------ parameter-space bitmap (dormant) ----
def largestconncomp(bimage, indmap=None, connectivity="default", device=0):
    """largestconncomp(bimage, indmap, conn) (parameterspacebitmap.jl:69-109).  bimage[x, y];
    connectivity "default" (4) or "eight".  Returns the indmap entries of the largest component
    (or 0-based column-major linear indices when indmap is None)."""
    conn8 = {"default": 0, "eight": 1, 4: 0, 8: 1, False: 0, True: 1}[connectivity]
    bm = np.asarray(bimage, dtype=np.uint8)
    xs, ys = bm.shape
    flat = np.ascontiguousarray(bm.reshape(-1, order="F"))
    out = np.zeros(max(1, flat.size), dtype=np.int64)
    n = C.c_int64()
    check(lib().rh_largestconncomp(_p(flat, C.c_uint8), xs, ys, conn8, device, _p(out, C.c_int64), out.size, C.byref(n)))
    lin = out[: n.value].copy()
    if indmap is None:
        return lin
    res = []
    for li in lin:
        v = indmap[int(li) % xs][int(li) // xs]
        res.extend(v if isinstance(v, (list, tuple, np.ndarray)) else [v])
    return res


def bitmapparameters(parameters, compatibility, beta, idsource=None):  # parameterspacebitmap.jl:12-60
    prm = _f64(parameters).reshape(-1, 2)
    comp = np.ascontiguousarray(compatibility, dtype=np.uint8)
    n = prm.shape[0]
    assert n == comp.size and (idsource is None or len(idsource) == n), "Everything must have the same length."
    ids = None if idsource is None else np.ascontiguousarray(idsource, dtype=np.int64)
    xs, ys, bx, by = C.c_int32(), C.c_int32(), C.c_double(), C.c_double()
    idp = None if ids is None else _p(ids, C.c_int64)
    check(lib().rh_bitmapparameters(_p(prm, C.c_double), _p(comp, C.c_uint8), idp, n, beta, C.byref(xs), C.byref(ys),
                                    C.byref(bx), C.byref(by), None, None))
    bitmap = np.zeros(xs.value * ys.value, dtype=np.uint8)
    idxmap = np.zeros(xs.value * ys.value, dtype=np.int64)
    check(lib().rh_bitmapparameters(_p(prm, C.c_double), _p(comp, C.c_uint8), idp, n, beta, C.byref(xs), C.byref(ys),
                                    C.byref(bx), C.byref(by), _p(bitmap, C.c_uint8), _p(idxmap, C.c_int64)))
    shp = (xs.value, ys.value)
    return bitmap.reshape(shp, order="F").astype(bool), idxmap.reshape(shp, order="F"), (bx.value, by.value)
