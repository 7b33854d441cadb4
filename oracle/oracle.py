"""ctypes wrapper around oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference's algorithm (see
ransac_oracle.h).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

PLANE, SPHERE, CYLINDER, CONE = 0, 1, 2, 3
SCORE_INT64_WRAP, SCORE_F64 = 0, 1
S_LENGTHC, S_ALLCAND, S_NOFMINSET = 1, 2, 3


class Shape(C.Structure):
    _fields_ = [("kind", C.c_int32), ("outwards", C.c_int32), ("v", C.c_double * 10)]


class Params(C.Structure):
    _fields_ = [
        ("eps", C.c_double * 4),
        ("alpha", C.c_double * 4),
        ("cos_alpha", C.c_double * 4),
        ("collin_threshold", C.c_double),
        ("parallelthrdeg", C.c_double),
        ("cos_parallelthr", C.c_double),
        ("sphere_par", C.c_double),
        ("minconeopang", C.c_double),
        ("prob_det", C.c_double),
        ("tau", C.c_int64),
        ("itermax", C.c_int64),
        ("drawN", C.c_int32),
        ("minsubsetN", C.c_int32),
        ("extract_s", C.c_int32),
        ("terminate_s", C.c_int32),
        ("n_shape_types", C.c_int32),
        ("shape_types", C.c_int32 * 8),
        ("score_mode", C.c_int32),
        ("sphere_uses_enabled", C.c_int32),
        ("sampling_streams", C.c_int32),
        ("octree_sampling", C.c_int32),
        ("octree_max_depth", C.c_int32),
    ]


class CI(C.Structure):
    _fields_ = [("min", C.c_double), ("max", C.c_double), ("E", C.c_double)]


class Rng(C.Structure):
    _fields_ = [
        ("s", C.c_uint64 * 4),
        ("stream", C.POINTER(C.c_uint64)),
        ("stream_len", C.c_int64),
        ("stream_pos", C.c_int64),
        ("draws", C.c_int64),
    ]


class Extracted(C.Structure):
    _fields_ = [
        ("shape", Shape),
        ("n_inpoints", C.c_int64),
        ("inpoints", C.POINTER(C.c_int64)),
        ("score_E", C.c_double),
        ("iteration", C.c_int64),
    ]


class Result(C.Structure):
    _fields_ = [
        ("shapes", C.POINTER(Extracted)),
        ("n_shapes", C.c_int64),
        ("iterations", C.c_int64),
        ("candidates_scored", C.c_int64),
        ("scored_left", C.c_int64),
        ("seconds", C.c_double),
    ]


def build(force=False):
    """Compile liboracle.so with gcc (Makefile in this directory)."""
    return _build_target(_SO, force)


def _build_target(so, force=False):
    deps = [os.path.join(_HERE, f) for f in ("ransac_oracle.c", "orc_f32.c", "ransac_oracle.h", "orc_trig.h", "Makefile")]
    if not force and os.path.exists(so) and os.path.getmtime(so) >= max(os.path.getmtime(d) for d in deps):
        return so
    subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(so)], stdout=subprocess.DEVNULL)
    return so


VARIANTS = {0: "default", 1: "fma", 2: "div", 3: "scaled", 4: "pairwise", 5: "libm"}
_variant_libs = {}


def variant_lib(v):
    """liboracle_v<v>.so: the same restatement under another reading of the StaticArrays rounding order
    (ransac_oracle.c header).  Measurement tooling only -- parity tests always use lib()."""
    if v == 0:
        return lib()
    if v not in _variant_libs:
        L = _bind(C.CDLL(_build_target(os.path.join(_HERE, "liboracle_v%d.so" % v))))
        assert L.orc_variant() == v
        _variant_libs[v] = L
    return _variant_libs[v]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    _lib = _bind(C.CDLL(_SO))
    return _lib


def _bind(L):
    dp = C.POINTER(C.c_double)
    i64p = C.POINTER(C.c_int64)
    u64p = C.POINTER(C.c_uint64)
    u8p = C.POINTER(C.c_uint8)
    i32p = C.POINTER(C.c_int32)
    sp, pp = C.POINTER(Shape), C.POINTER(Params)
    sig = {
        "orc_default_params": (None, [pp]),
        "orc_params_finalize": (None, [pp]),
        "orc_shape_finalize": (None, [sp]),
        "orc_compatible": (C.c_int, [sp, dp, dp, C.c_double, C.c_double]),
        "orc_compat_values": (None, [sp, dp, dp, dp]),
        "orc_variant": (C.c_int, []),
        "orc32_shape_finalize": (None, [sp]),
        "orc32_compatible": (C.c_int, [sp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_double, C.c_double]),
        "orc32_cloud_create": (C.c_void_p, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int64, i64p, C.c_int64]),
        "orc32_cloud_destroy": (None, [C.c_void_p]),
        "orc32_cloud_enable_all": (None, [C.c_void_p]),
        "orc32_cloud_set_enabled": (None, [C.c_void_p, u64p, C.c_int64]),
        "orc32_cloud_get_enabled": (None, [C.c_void_p, u64p, C.c_int64]),
        "orc32_scorecandidate": (C.c_int64, [C.c_void_p, sp, pp, i64p, u64p]),
        "orc32_score_masks_mt": (None, [C.c_void_p, sp, C.c_int32, pp, i32p, u64p, C.c_int32]),
        "orc32_refit": (C.c_int64, [C.c_void_p, sp, pp, i64p, C.c_int64]),
        "orc32_invalidate": (None, [C.c_void_p, i64p, C.c_int64]),
        "orc32_fit": (C.c_int, [C.c_int, dp, dp, C.c_int, pp, sp]),
        "orc_cloud_set_f32": (None, [C.c_void_p, C.c_int]),
        "orc_score_masks_mt": (None, [C.c_void_p, sp, C.c_int32, pp, i32p, u64p, C.c_int32]),
        "orc_margin_census_mt": (None, [C.c_void_p, sp, C.c_int32, pp, dp, C.c_int, i64p, C.c_int32]),
        "orc_confidence_interval": (C.c_int, [C.c_double, C.c_double, C.POINTER(CI)]),
        "orc_notsoconfident": (CI, [C.c_double, C.c_double]),
        "orc_isoverlap": (C.c_int, [CI, CI]),
        "orc_estimatescore": (CI, [C.c_int64, C.c_int64, C.c_int64, C.c_int]),
        "orc_prob": (C.c_double, [C.c_double, C.c_int64, C.c_int64, C.c_int64]),
        "orc_cloud_create": (C.c_void_p, [dp, dp, C.c_int64, i64p, C.c_int64]),
        "orc_cloud_destroy": (None, [C.c_void_p]),
        "orc_cloud_set_enabled": (None, [C.c_void_p, u64p, C.c_int64]),
        "orc_cloud_get_enabled": (None, [C.c_void_p, u64p, C.c_int64]),
        "orc_cloud_enable_all": (None, [C.c_void_p]),
        "orc_cloud_count_enabled": (C.c_int64, [C.c_void_p]),
        "orc_scorecandidate": (C.c_int64, [C.c_void_p, sp, pp, i64p, u64p]),
        "orc_score_batch": (None, [C.c_void_p, sp, C.c_int32, pp, i32p, u64p]),
        "orc_score_batch_mt": (None, [C.c_void_p, sp, C.c_int32, pp, i32p, C.c_int32]),
        "orc_refit": (C.c_int64, [C.c_void_p, sp, pp, i64p, C.c_int64]),
        "orc_invalidate": (None, [C.c_void_p, i64p, C.c_int64]),
        "orc_refit_lsq": (C.c_int, [C.c_void_p, sp, pp, C.c_int, sp, i64p, dp, i32p]),
        "orc_select_enabled": (C.c_int64, [C.c_void_p, C.c_int64]),
        "orc_fit": (C.c_int, [C.c_int, dp, dp, C.c_int, pp, sp]),
        "orc_fit2pointsphere": (C.c_int, [dp, dp, pp, sp]),
        "orc_fit2pointcylinder": (C.c_int, [dp, dp, pp, sp]),
        "orc_fit3pointcone": (C.c_int, [dp, dp, sp]),
        "orc_rodriguesrad": (None, [dp, C.c_double, dp]),
        "orc_pluscrossprod": (None, [dp, C.c_double, dp]),
        "orc_rank": (C.c_int, [dp, C.c_int, C.c_int]),
        "orc_findAABB": (None, [dp, C.c_int64, C.c_int, dp, dp]),
        "orc_iswithinrectangle": (C.c_int, [dp, dp, dp]),
        "orc_octree_build": (C.c_void_p, [dp, C.c_int64]),
        "orc_octree_build_f32": (C.c_void_p, [C.POINTER(C.c_float), C.c_int64]),
        "orc_octree_destroy": (None, [C.c_void_p]),
        "orc_octree_depth": (C.c_int, [C.c_void_p]),
        "orc_octree_findleaf": (C.c_int, [C.c_void_p, dp, i32p, C.c_int]),
        "orc_octree_node_npoints": (C.c_int64, [C.c_void_p, C.c_int32]),
        "orc_octree_node_points": (i64p, [C.c_void_p, C.c_int32]),
        "orc_rng_seed": (None, [C.POINTER(Rng), C.c_uint64]),
        "orc_rng_next": (C.c_uint64, [C.POINTER(Rng)]),
        "orc_rng_range": (C.c_int64, [C.POINTER(Rng), C.c_int64]),
        "orc_ransac": (C.c_int, [C.c_void_p, dp, dp, pp, C.POINTER(Rng), C.c_int, C.POINTER(Result)]),
        "orc_result_free": (None, [C.POINTER(Result)]),
        "orc_largestconncomp": (C.c_int64, [u8p, C.c_int32, C.c_int32, C.c_int, i64p, C.c_int64]),
        "orc_bitmapparameters": (C.c_int, [dp, u8p, i64p, C.c_int64, C.c_double, i32p, i32p, dp, dp, u8p, i64p]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def default_params(**kw):
    p = Params()
    lib().orc_default_params(C.byref(p))
    for k, v in kw.items():
        if k == "shape_types":
            p.n_shape_types = len(v)
            for i, t in enumerate(v):
                p.shape_types[i] = t
        elif isinstance(v, (list, tuple, np.ndarray)):
            arr = getattr(p, k)
            for i, x in enumerate(v):
                arr[i] = x
        else:
            setattr(p, k, v)
    lib().orc_params_finalize(C.byref(p))
    return p


def make_shape(kind, outwards, v):
    s = Shape()
    s.kind = kind
    s.outwards = int(bool(outwards))
    for i, x in enumerate(v):
        s.v[i] = float(x)
    lib().orc_shape_finalize(C.byref(s))
    return s


def shapes_array(shapes):
    arr = (Shape * max(1, len(shapes)))()
    for i, s in enumerate(shapes):
        arr[i] = s
    return arr


class Cloud:
    """Mirror of the pieces of RANSACCloud (octree.jl:37-59) the path reads.  f32: a Float32 cloud (force_eltype = Float32,
    octree.jl:102-109) -- the values are rounded to binary32 and every per-point test and fit, the whole ransac() loop
    included, runs in binary32 (orc_f32.c)."""

    def __init__(self, xyz, nrm, subset1_1based, L=None, f32=False):
        self.L = L if L is not None else lib()
        if f32:
            xyz, nrm = np.asarray(xyz, dtype=np.float32), np.asarray(nrm, dtype=np.float32)
        self.xyz = _f64(xyz).reshape(-1, 3)
        self.nrm = _f64(nrm).reshape(-1, 3)
        self.subset1 = np.ascontiguousarray(subset1_1based, dtype=np.int64)
        self.n = self.xyz.shape[0]
        self.s = self.subset1.shape[0]
        self.h = self.L.orc_cloud_create(_dp(self.xyz), _dp(self.nrm), self.n,
                                        self.subset1.ctypes.data_as(C.POINTER(C.c_int64)), self.s)
        if f32:
            self.L.orc_cloud_set_f32(self.h, 1)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_cloud_destroy(self.h)
            self.h = None

    @property
    def nchunks(self):
        return (self.n + 63) // 64

    def set_enabled(self, chunks):
        chunks = np.ascontiguousarray(chunks, dtype=np.uint64)
        self.L.orc_cloud_set_enabled(self.h, chunks.ctypes.data_as(C.POINTER(C.c_uint64)), chunks.size)

    def get_enabled(self):
        out = np.zeros(self.nchunks, dtype=np.uint64)
        self.L.orc_cloud_get_enabled(self.h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size)
        return out

    def enable_all(self):
        self.L.orc_cloud_enable_all(self.h)

    def count_enabled(self):
        return self.L.orc_cloud_count_enabled(self.h)

    def scorecandidate(self, shape, params, want_mask=False):
        inp = np.zeros(max(1, self.s), dtype=np.int64)
        w = (self.s + 63) // 64
        mask = np.zeros(max(1, w), dtype=np.uint64) if want_mask else None
        cnt = self.L.orc_scorecandidate(self.h, C.byref(shape), C.byref(params),
                                       inp.ctypes.data_as(C.POINTER(C.c_int64)),
                                       mask.ctypes.data_as(C.POINTER(C.c_uint64)) if want_mask else None)
        return (cnt, inp[:cnt].copy(), mask[:w]) if want_mask else (cnt, inp[:cnt].copy())

    def score_batch(self, shapes, params, want_masks=False):
        b = len(shapes)
        arr = shapes if isinstance(shapes, C.Array) else shapes_array(shapes)
        counts = np.zeros(max(1, b), dtype=np.int32)
        w = (self.s + 63) // 64
        masks = np.zeros((max(1, b), max(1, w)), dtype=np.uint64) if want_masks else None
        self.L.orc_score_batch(self.h, arr, b, C.byref(params),
                              counts.ctypes.data_as(C.POINTER(C.c_int32)),
                              masks.ctypes.data_as(C.POINTER(C.c_uint64)) if want_masks else None)
        return (counts[:b], masks[:b, :w]) if want_masks else counts[:b]

    def score_masks_mt(self, shapes, params, nthreads):
        """Counts and masks, candidates spread over `nthreads` host threads."""
        b = len(shapes)
        arr = shapes if isinstance(shapes, C.Array) else shapes_array(shapes)
        counts = np.zeros(max(1, b), dtype=np.int32)
        w = (self.s + 63) // 64
        masks = np.zeros((max(1, b), max(1, w)), dtype=np.uint64)
        self.L.orc_score_masks_mt(self.h, arr, b, C.byref(params), counts.ctypes.data_as(C.POINTER(C.c_int32)),
                                  masks.ctypes.data_as(C.POINTER(C.c_uint64)), nthreads)
        return counts[:b], masks[:b, :w]

    def margin_census(self, shapes, params, edges, nthreads):
        """Tests whose distance / angle side lies within edges[k] of its threshold: (2, len(edges)) counts."""
        b = len(shapes)
        arr = shapes if isinstance(shapes, C.Array) else shapes_array(shapes)
        e = _f64(edges)
        hist = np.zeros((2, e.size), dtype=np.int64)
        self.L.orc_margin_census_mt(self.h, arr, b, C.byref(params), _dp(e), e.size,
                                    hist.ctypes.data_as(C.POINTER(C.c_int64)), nthreads)
        return hist

    def score_batch_mt(self, shapes, params, nthreads):
        """Counts only, candidates spread over `nthreads` host threads (OpenMP)."""
        b = len(shapes)
        arr = shapes if isinstance(shapes, C.Array) else shapes_array(shapes)
        counts = np.zeros(max(1, b), dtype=np.int32)
        self.L.orc_score_batch_mt(self.h, arr, b, C.byref(params), counts.ctypes.data_as(C.POINTER(C.c_int32)), nthreads)
        return counts[:b]

    def refit(self, shape, params):
        out = np.zeros(max(1, self.n), dtype=np.int64)
        cnt = self.L.orc_refit(self.h, C.byref(shape), C.byref(params),
                              out.ctypes.data_as(C.POINTER(C.c_int64)), self.n)
        return out[:cnt].copy()

    def refit_lsq(self, shape, params, max_iter=10):
        out, n, rms, it = Shape(), C.c_int64(), C.c_double(), C.c_int32()
        rc = self.L.orc_refit_lsq(self.h, C.byref(shape), C.byref(params), max_iter, C.byref(out), C.byref(n),
                                 C.byref(rms), C.byref(it))
        if rc:
            raise RuntimeError("orc_refit_lsq failed: %d" % rc)
        return out, n.value, rms.value, it.value

    def invalidate(self, idx_1based):
        idx = np.ascontiguousarray(idx_1based, dtype=np.int64)
        self.L.orc_invalidate(self.h, idx.ctypes.data_as(C.POINTER(C.c_int64)), idx.size)

    def select_enabled(self, k):
        return self.L.orc_select_enabled(self.h, int(k))

    def ransac(self, params, seed=1234, stream=None, octree_depth=1):
        rng = Rng()
        self.L.orc_rng_seed(C.byref(rng), seed)
        if stream is not None:
            stream = np.ascontiguousarray(stream, dtype=np.uint64)
            rng.stream = stream.ctypes.data_as(C.POINTER(C.c_uint64))
            rng.stream_len = stream.size
        res = Result()
        rc = self.L.orc_ransac(self.h, _dp(self.xyz), _dp(self.nrm), C.byref(params),
                              C.byref(rng), octree_depth, C.byref(res))
        out = {"rc": rc, "iterations": res.iterations, "candidates_scored": res.candidates_scored,
               "scored_left": res.scored_left, "seconds": res.seconds, "draws": rng.draws, "shapes": []}
        for i in range(res.n_shapes):
            e = res.shapes[i]
            sh = Shape.from_buffer_copy(bytes(e.shape))
            idx = np.ctypeslib.as_array(e.inpoints, shape=(max(1, e.n_inpoints),))[: e.n_inpoints].copy()
            out["shapes"].append({"shape": sh, "inpoints": idx, "score_E": e.score_E, "iteration": e.iteration})
        self.L.orc_result_free(C.byref(res))
        return out


class Cloud32:
    """The Float32 twin of Cloud (RANSACCloud(...; force_eltype = Float32), octree.jl:102-109): orc_f32.c."""

    def __init__(self, xyz, nrm, subset1_1based):
        self.L = lib()
        self.xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        self.nrm = np.ascontiguousarray(nrm, dtype=np.float32).reshape(-1, 3)
        self.subset1 = np.ascontiguousarray(subset1_1based, dtype=np.int64)
        self.n, self.s = self.xyz.shape[0], self.subset1.shape[0]
        fp = C.POINTER(C.c_float)
        self.h = self.L.orc32_cloud_create(self.xyz.ctypes.data_as(fp), self.nrm.ctypes.data_as(fp), self.n,
                                           self.subset1.ctypes.data_as(C.POINTER(C.c_int64)), self.s)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc32_cloud_destroy(self.h)
            self.h = None

    @property
    def nchunks(self):
        return (self.n + 63) // 64

    def set_enabled(self, chunks):
        chunks = np.ascontiguousarray(chunks, dtype=np.uint64)
        self.L.orc32_cloud_set_enabled(self.h, chunks.ctypes.data_as(C.POINTER(C.c_uint64)), chunks.size)

    def get_enabled(self):
        out = np.zeros(self.nchunks, dtype=np.uint64)
        self.L.orc32_cloud_get_enabled(self.h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size)
        return out

    def enable_all(self):
        self.L.orc32_cloud_enable_all(self.h)

    def score_batch(self, shapes, params, want_masks=False, nthreads=8):
        b = len(shapes)
        arr = shapes if isinstance(shapes, C.Array) else shapes_array(shapes)
        counts = np.zeros(max(1, b), dtype=np.int32)
        w = (self.s + 63) // 64
        masks = np.zeros((max(1, b), max(1, w)), dtype=np.uint64) if want_masks else None
        self.L.orc32_score_masks_mt(self.h, arr, b, C.byref(params), counts.ctypes.data_as(C.POINTER(C.c_int32)),
                                    masks.ctypes.data_as(C.POINTER(C.c_uint64)) if want_masks else None, nthreads)
        return (counts[:b], masks[:b, :w]) if want_masks else counts[:b]

    def refit(self, shape, params):
        out = np.zeros(max(1, self.n), dtype=np.int64)
        cnt = self.L.orc32_refit(self.h, C.byref(shape), C.byref(params), out.ctypes.data_as(C.POINTER(C.c_int64)), self.n)
        return out[:cnt].copy()

    def invalidate(self, idx_1based):
        idx = np.ascontiguousarray(idx_1based, dtype=np.int64)
        self.L.orc32_invalidate(self.h, idx.ctypes.data_as(C.POINTER(C.c_int64)), idx.size)


def make_shape32(kind, outwards, v):
    """A Float32 shape: fields rounded to binary32, the cone's cos / sin as binary32."""
    s = Shape()
    s.kind = kind
    s.outwards = int(bool(outwards))
    for i, x in enumerate(v):
        s.v[i] = float(x)
    lib().orc32_shape_finalize(C.byref(s))
    return s


def compatible32(shape, p, n, eps, cos_alpha):
    p, n = np.ascontiguousarray(p, dtype=np.float32), np.ascontiguousarray(n, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    return bool(lib().orc32_compatible(C.byref(shape), p.ctypes.data_as(fp), n.ctypes.data_as(fp), eps, cos_alpha))


def fit32(kind, p, n, params):
    """fit on Float32 points (orc_f32.c); cones never fit"""
    p, n = _f64(np.asarray(p, dtype=np.float32)).reshape(-1, 3), _f64(np.asarray(n, dtype=np.float32)).reshape(-1, 3)
    out = Shape()
    ok = lib().orc32_fit(kind, _dp(p), _dp(n), p.shape[0], C.byref(params), C.byref(out))
    return out if ok else None


def fit(kind, p, n, params):
    p, n = _f64(p).reshape(-1, 3), _f64(n).reshape(-1, 3)
    out = Shape()
    ok = lib().orc_fit(kind, _dp(p), _dp(n), p.shape[0], C.byref(params), C.byref(out))
    return out if ok else None


def compatible(shape, p, n, eps, cos_alpha):
    p, n = _f64(p), _f64(n)
    return bool(lib().orc_compatible(C.byref(shape), _dp(p), _dp(n), eps, cos_alpha))


def estimatescore(S1, P, sigma, mode=SCORE_INT64_WRAP):
    ci = lib().orc_estimatescore(S1, P, sigma, mode)
    return ci.min, ci.max, ci.E


def prob(n, s, N, k):
    return lib().orc_prob(float(n), int(s), int(N), int(k))


def largestconncomp(bitmap_xy, conn8=False):
    """bitmap_xy: 2-D bool array indexed [x, y] like a Julia BitMatrix.
    Returns 0-based column-major linear indices of the largest component."""
    bm = np.asfortranarray(np.asarray(bitmap_xy, dtype=np.uint8))
    xs, ys = bm.shape
    flat = np.ascontiguousarray(bm.reshape(-1, order="F"))
    out = np.zeros(max(1, flat.size), dtype=np.int64)
    k = lib().orc_largestconncomp(flat.ctypes.data_as(C.POINTER(C.c_uint8)), xs, ys, int(conn8),
                                  out.ctypes.data_as(C.POINTER(C.c_int64)), out.size)
    return out[:k].copy()


def bitmapparameters(params2d, compat, beta, idsource=None):
    prm = _f64(params2d).reshape(-1, 2)
    n = prm.shape[0]
    comp = np.ascontiguousarray(compat, dtype=np.uint8)
    ids = None if idsource is None else np.ascontiguousarray(idsource, dtype=np.int64)
    xs, ys = C.c_int32(), C.c_int32()
    bx, by = C.c_double(), C.c_double()
    idp = None if ids is None else ids.ctypes.data_as(C.POINTER(C.c_int64))
    rc = lib().orc_bitmapparameters(_dp(prm), comp.ctypes.data_as(C.POINTER(C.c_uint8)), idp, n, beta,
                                    C.byref(xs), C.byref(ys), C.byref(bx), C.byref(by), None, None)
    if rc:
        raise AssertionError("max-min should be positive")
    bitmap = np.zeros(xs.value * ys.value, dtype=np.uint8)
    idxmap = np.zeros(xs.value * ys.value, dtype=np.int64)
    lib().orc_bitmapparameters(_dp(prm), comp.ctypes.data_as(C.POINTER(C.c_uint8)), idp, n, beta,
                               C.byref(xs), C.byref(ys), C.byref(bx), C.byref(by),
                               bitmap.ctypes.data_as(C.POINTER(C.c_uint8)),
                               idxmap.ctypes.data_as(C.POINTER(C.c_int64)))
    shape = (xs.value, ys.value)
    return bitmap.reshape(shape, order="F").astype(bool), idxmap.reshape(shape, order="F"), (bx.value, by.value)


class Octree:
    def __init__(self, xyz):
        if np.asarray(xyz).dtype == np.float32:   # a Float32 cloud's tree
            self.xyz32 = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
            self.xyz = self.xyz32.astype(np.float64)
            self.h = lib().orc_octree_build_f32(self.xyz32.ctypes.data_as(C.POINTER(C.c_float)), self.xyz32.shape[0])
            return
        self.xyz = _f64(xyz).reshape(-1, 3)
        self.h = lib().orc_octree_build(_dp(self.xyz), self.xyz.shape[0])

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_octree_destroy(self.h)
            self.h = None

    def depth(self):
        return lib().orc_octree_depth(self.h)

    def findleaf(self, p):
        p = _f64(p)
        path = (C.c_int32 * 64)()
        d = lib().orc_octree_findleaf(self.h, _dp(p), path, 64)
        return d, list(path[:d])

    def node_points(self, node):
        n = lib().orc_octree_node_npoints(self.h, node)
        ptr = lib().orc_octree_node_points(self.h, node)
        return np.ctypeslib.as_array(ptr, shape=(max(1, n),))[:n].copy()
