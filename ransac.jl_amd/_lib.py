"""ctypes binding of libransac_hip.so (include/ransac_hip.h).  Loading fails loudly when the
HIP library has not been built: there is no CPU fallback in the product path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libransac_hip.so")
# the -DRH_DIAG build of the same sources (build.py): the A/B switches of the experiments, RH_* environment variables and
# the rh_dbg_* audits live there and only there.  RH_LIB_VARIANT=diag makes it this process's library (tests marked `diag`,
# tools/fuzz_*.py, the profiling tools); the default is the product library, which reads no environment variable.
SO_PATH_DIAG = os.path.join(_HERE, "libransac_hip_diag.so")
OPTION_UNSET = -(1 << 63)
SCORE_PATH = {"auto": 0, "brute": 1, "groups": 2}
REFIT_PATH = {"auto": 0, "scan": 1, "culled": 2}

PLANE, SPHERE, CYLINDER, CONE = 0, 1, 2, 3
KIND_NAMES = {PLANE: "plane", SPHERE: "sphere", CYLINDER: "cylinder", CONE: "cone"}
SCORE_INT64_WRAP, SCORE_F64 = 0, 1
S_LENGTHC, S_ALLCAND, S_NOFMINSET = 1, 2, 3
RH_OK, RH_E_INVALID, RH_E_NODEVICE, RH_E_NOMEM, RH_E_CAPACITY, RH_E_INTERNAL = 0, -1, -2, -3, -4, -5


class Shape(C.Structure):
    _fields_ = [("kind", C.c_int32), ("outwards", C.c_int32), ("v", C.c_double * 10)]


class Params(C.Structure):
    _fields_ = [
        ("eps", C.c_double * 4), ("alpha", C.c_double * 4), ("cos_alpha", C.c_double * 4),
        ("collin_threshold", C.c_double), ("parallelthrdeg", C.c_double), ("cos_parallelthr", C.c_double),
        ("sphere_par", C.c_double), ("minconeopang", C.c_double), ("prob_det", C.c_double),
        ("tau", C.c_int64), ("itermax", C.c_int64),
        ("drawN", C.c_int32), ("minsubsetN", C.c_int32), ("extract_s", C.c_int32), ("terminate_s", C.c_int32),
        ("n_shape_types", C.c_int32), ("shape_types", C.c_int32 * 8),
        ("score_mode", C.c_int32), ("sphere_uses_enabled", C.c_int32), ("sampling_streams", C.c_int32),
        ("octree_sampling", C.c_int32), ("octree_max_depth", C.c_int32),
    ]


class Rng(C.Structure):
    _fields_ = [("s", C.c_uint64 * 4), ("stream", C.POINTER(C.c_uint64)), ("stream_len", C.c_int64),
                ("stream_pos", C.c_int64), ("draws", C.c_int64)]


class Extracted(C.Structure):
    _fields_ = [("shape", Shape), ("n_inpoints", C.c_int64), ("inpoints", C.POINTER(C.c_int64)),
                ("score_E", C.c_double), ("iteration", C.c_int64)]


class Result(C.Structure):
    _fields_ = [("shapes", C.POINTER(Extracted)), ("n_shapes", C.c_int64), ("iterations", C.c_int64),
                ("candidates_scored", C.c_int64), ("scored_left", C.c_int64), ("seconds", C.c_double),
                ("seconds_score", C.c_double), ("seconds_extract", C.c_double), ("seconds_host", C.c_double),
                ("seconds_to_last_extraction", C.c_double), ("arena", C.c_void_p)]


class RansacHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libransac_hip error %d: %s" % (code, msg))
        self.code = code


_dp = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)
_sp, _pp, _vp = C.POINTER(Shape), C.POINTER(Params), C.c_void_p

# every symbol include/ransac_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "rh_version": (C.c_int, []),
    "rh_last_error": (C.c_char_p, []),
    "rh_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rh_default_params": (None, [_pp]),
    "rh_params_finalize": (None, [_pp]),
    "rh_shape_finalize": (None, [_sp]),
    "rh_cloud_create": (C.c_int, [_dp, _dp, C.c_int64, _i64p, C.c_int64, C.c_int, C.POINTER(_vp)]),
    "rh_cloud_destroy": (C.c_int, [_vp]),
    "rh_cloud_info": (C.c_int, [_vp, _i64p, _i64p, C.POINTER(C.c_int)]),
    "rh_cloud_set_enabled": (C.c_int, [_vp, _u64p, C.c_int64]),
    "rh_cloud_get_enabled": (C.c_int, [_vp, _u64p, C.c_int64]),
    "rh_cloud_enable_all": (C.c_int, [_vp]),
    "rh_cloud_count_enabled": (C.c_int, [_vp, _i64p]),
    "rh_score_batch": (C.c_int, [_vp, _sp, C.c_int32, _pp, _i32p, _u64p]),
    "rh_score_batch_dev": (C.c_int, [_vp, _vp, C.c_int32, _pp, _vp, _vp]),
    "rh_cloud_set_stream": (C.c_int, [_vp, _vp, C.c_int]),
    "rh_score_batch_dev_timed": (C.c_int, [_vp, _vp, C.c_int32, _pp, _vp, _vp, C.POINTER(C.c_float)]),
    "rh_refit": (C.c_int, [_vp, _sp, _pp, _i64p, C.c_int64, _i64p]),
    "rh_refit_lsq": (C.c_int, [_vp, _sp, _pp, C.c_int32, _sp, _i64p, _dp, _i32p]),
    "rh_invalidate": (C.c_int, [_vp, _i64p, C.c_int64]),
    "rh_select_enabled": (C.c_int, [_vp, _i64p, C.c_int32, _i64p]),
    "rh_sample_sets": (C.c_int, [_vp, C.c_int32, C.POINTER(Rng), C.c_int32, _i64p, _i32p, _i32p]),
    "rh_fit_sets": (C.c_int, [_dp, _dp, _i64p, _i32p, C.c_int32, C.c_int32, _pp, C.c_int32, _sp, _i32p, C.c_int32, _i32p]),
    "rh_fit": (C.c_int, [C.c_int, _dp, _dp, C.c_int32, _pp, _sp, _i32p]),
    "rh_fit_f32": (C.c_int, [C.c_int, _dp, _dp, C.c_int32, _pp, _sp, _i32p]),
    "rh_estimatescore": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int32, _dp, _dp, _dp]),
    "rh_prob": (C.c_double, [C.c_double, C.c_int64, C.c_int64, C.c_int64]),
    "rh_rng_seed": (None, [C.POINTER(Rng), C.c_uint64]),
    "rh_rng_range": (C.c_int64, [C.POINTER(Rng), C.c_int64]),
    "rh_ransac": (C.c_int, [_vp, _dp, _dp, _pp, C.POINTER(Rng), C.POINTER(Result)]),
    "rh_ransac_f32": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), _pp, C.POINTER(Rng), C.POINTER(Result)]),
    "rh_result_free": (None, [C.POINTER(Result)]),
    "rh_largestconncomp": (C.c_int, [_u8p, C.c_int32, C.c_int32, C.c_int32, C.c_int, _i64p, C.c_int64, _i64p]),
    "rh_bitmapparameters": (C.c_int, [_dp, _u8p, _i64p, C.c_int64, C.c_double, _i32p, _i32p, _dp, _dp, _u8p, _i64p]),
    "rh_last_refit_ms": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "rh_timer_start": (C.c_int, [_vp]),
    "rh_timer_stop": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "rh_cloud_sync": (C.c_int, [_vp]),
    "rh_dev_alloc": (C.c_int, [_vp, C.c_int64, C.POINTER(_vp)]),
    "rh_cloud_create_f32": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_int64), C.c_int64, C.c_int, C.POINTER(_vp)]),
    "rh_shape_finalize_f32": (None, [C.POINTER(Shape)]),
    "rh_mp_open": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_int64, C.POINTER(_vp)]),
    "rh_mp_close": (C.c_int, [_vp]),
    "rh_mp_allgather": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "rh_ransac_mp": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(Params), C.POINTER(Rng), _vp, C.POINTER(Result)]),
    "rh_dev_free": (C.c_int, [_vp, _vp]),
    "rh_dev_upload": (C.c_int, [_vp, _vp, _vp, C.c_int64]),
    "rh_dev_download": (C.c_int, [_vp, _vp, _vp, C.c_int64]),
    "rh_octree_build": (C.c_int, [_dp, C.c_int64, C.POINTER(C.c_void_p)]),
    "rh_octree_build_f32": (C.c_int, [C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_void_p)]),
    "rh_octree_destroy": (C.c_int, [_vp]),
    "rh_octree_info": (C.c_int, [_vp, _i32p, _i32p, _i32p]),
    "rh_octree_findleaf": (C.c_int, [_vp, _dp, _i32p]),
    "rh_octree_getnthcell": (C.c_int, [_vp, C.c_int32, C.c_int32, _i32p]),
    "rh_octree_node_info": (C.c_int, [_vp, C.c_int32, _dp, _dp, _i32p, _i32p, _i32p, C.POINTER(C.c_int64)]),
    "rh_octree_node_points": (C.c_int, [_vp, C.c_int32, C.POINTER(C.c_int64), C.c_int64]),
    "rh_octree_cell_enabled": (C.c_int, [_vp, _vp, C.c_int32, C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_int64)]),
    "rh_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int64]),
    "rh_get_option": (C.c_int, [_vp, C.c_char_p, _i64p, _i32p]),
    "rh_score_launch_info": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "rh_last_list_launch_ms": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "rh_build_variant": (C.c_int, []),
    "rh_cloud_create_ms": (C.c_int, [_vp, _dp]),
    "rh_comm_unique_id": (C.c_int, [_vp]),
    "rh_comm_create": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp, C.POINTER(_vp)]),
    "rh_comm_destroy": (C.c_int, [_vp]),
    "rh_score_batch_allreduce_dev": (C.c_int, [_vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _pp, _vp]),
    "rh_comm_fence": (C.c_int, [_vp, _vp]),
    "rh_comm_sync": (C.c_int, [_vp]),
}

# include/ransac_hip_diag.h: exported by the diag build only
DIAG_SIGNATURES = {
    "rh_dbg_cls_audit": (C.c_int, [_vp, _sp, C.c_int32, _pp, _dp]),
    "rh_dbg_cls_soundness": (C.c_int, [_vp, _sp, C.c_int32, _pp, C.POINTER(C.c_uint64)]),
    "rh_dbg_oct_search_selftest": (C.c_int, [C.c_int64, C.c_uint64, C.c_int64, C.POINTER(C.c_int64)]),
    "rh_dbg_s4_stats": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_uint64)]),
}

_libs = {}


def variant():
    """which build this process uses by default: "product" (libransac_hip.so) unless RH_LIB_VARIANT=diag"""
    v = os.environ.get("RH_LIB_VARIANT", "product")
    if v not in ("product", "diag"):
        raise RuntimeError("RH_LIB_VARIANT=%r: expected 'product' or 'diag'" % v)
    return v


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so (SONAME
    libamdhip64.so.7, the name libransac_hip.so needs) and loads it by file name: if the system copy
    under /opt/rocm is mapped first, a later `import torch` maps a SECOND runtime that finds no device
    ("no ROCm-capable device is detected").  When torch is installed but not imported yet, map its copy
    first so both sides bind to the same one (RH_SYSTEM_HIP=1 skips this)."""
    import sys
    if "torch" in sys.modules or os.environ.get("RH_SYSTEM_HIP"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except (ImportError, OSError, ValueError):
        pass


def lib(which=None):
    """The loaded library (which = None: this process's default variant, see variant()).  Raises if the shared object is
    missing (run `python -c 'import __graft_entry__ as g; g.build()'` or `python ransac.jl_amd/build.py`)."""
    which = which or variant()
    L = _libs.get(which)
    if L is None:
        path = SO_PATH if which == "product" else SO_PATH_DIAG
        if os.environ.get("RH_LIB_PATH"):   # kernel A/B runs (tools/): another build of this variant of the library
            path = os.environ["RH_LIB_PATH"]
        if not os.path.exists(path):
            raise RuntimeError(
                "%s is missing: the HIP extension has not been built and this package has no CPU "
                "fallback (build it with ransac.jl_amd/build.py)" % path)
        _share_torch_hip_runtime()
        L = C.CDLL(path)
        sigs = dict(SIGNATURES)
        if which == "diag":
            sigs.update(DIAG_SIGNATURES)
        for name, (res, args) in sigs.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        if L.rh_build_variant() != (1 if which == "diag" else 0):
            raise RuntimeError("%s is not the %s build" % (path, which))
        _libs[which] = L
    return L


def check(rc):
    if rc != RH_OK:
        raise RansacHipError(rc, lib().rh_last_error().decode("utf-8", "replace"))
    return rc
