"""ctypes binding of libposehip.so (the C ABI declared in include/posehip.h).

There is no CPU fallback: if the HIP library is missing or a call fails, an
exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libposehip.so")

PH_OK = 0
NO_MATCH_IDX = -1
NO_MATCH_DIST = 2147483647


class PosehipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"posehip status {status}: {message}")
        self.status = status


class DMatchC(C.Structure):
    _fields_ = [("queryIdx", C.c_int32), ("trainIdx", C.c_int32), ("imgIdx", C.c_int32),
                ("distance", C.c_float)]


_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_dmp = C.POINTER(DMatchC)
_intp = C.POINTER(C.c_int)

# name -> (restype, argtypes); every symbol include/posehip.h declares.
SIGNATURES = {
    "ph_init": (C.c_int, [C.c_int]),
    "ph_shutdown": (None, []),
    "ph_last_error_string": (C.c_char_p, []),
    "ph_device": (C.c_int, []),
    "ph_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "ph_ctx_destroy": (C.c_int, [C.c_void_p]),
    "ph_ctx_make_current": (C.c_int, [C.c_void_p]),
    "ph_ctx_current": (C.c_void_p, []),
    "ph_ctx_device": (C.c_int, [C.c_void_p]),
    "ph_knn_hamming256": (C.c_int, [_u8p, C.c_int, C.c_size_t, _u8p, C.c_int, C.c_size_t, C.c_int,
                                    _i32p, _i32p]),
    "ph_match_hamming256": (C.c_int, [_u8p, C.c_int, C.c_size_t, _u8p, C.c_int, C.c_size_t, _dmp,
                                      _intp]),
    "ph_match_crosscheck_hamming256": (C.c_int, [_u8p, C.c_int, C.c_size_t, _u8p, C.c_int,
                                                 C.c_size_t, _dmp, _intp]),
    "ph_knn2_ratio_hamming256": (C.c_int, [_u8p, C.c_int, C.c_size_t, _u8p, C.c_int, C.c_size_t,
                                           C.c_float, _dmp, _intp]),
    "ph_min_dist_filter": (C.c_int, [_dmp, C.c_int, _dmp, _intp]),
    "ph_ratio_filter": (C.c_int, [_i32p, _i32p, C.c_int, C.c_float, _dmp, _intp]),
    "ph_db_upload": (C.c_int, [_u8p, C.c_int64, C.c_size_t, C.POINTER(C.c_void_p)]),
    "ph_db_free": (C.c_int, [C.c_void_p]),
    "ph_db_rows": (C.c_int64, [C.c_void_p]),
    "ph_db_device_ptr": (C.c_void_p, [C.c_void_p]),
    "ph_db_knn_hamming256": (C.c_int, [C.c_void_p, _u8p, C.c_int64, C.c_size_t, C.c_int, _i32p,
                                       _i32p, _u8p, C.c_float]),
    "ph_db_knn_hamming256_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    "ph_knn_hamming256_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int,
                                        C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                        C.c_void_p]),
    "ph_knn_hamming256_noalloc_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                                C.c_int, C.c_int32, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_float, C.c_void_p]),
    "ph_knn_hamming256_frames_dev": (C.c_int, [C.c_void_p, _i32p, C.c_void_p, _i32p, C.c_int,
                                               C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_float, C.c_void_p]),
    "ph_match_crosscheck_hamming256_frames_dev": (C.c_int, [C.c_void_p, _i32p, C.c_void_p, _i32p,
                                                            C.c_int, C.c_void_p, C.c_void_p,
                                                            C.c_void_p]),
    "ph_knn_merge_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    "ph_match_crosscheck_hamming256_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "ph_match_crosscheck_hamming256_shard_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p,
                                                           C.c_int64, C.c_int32, C.c_void_p,
                                                           C.c_void_p, C.c_void_p]),
    "ph_match_crosscheck_unpack_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ph_stream_synchronize": (C.c_int, [C.c_void_p]),
    "ph_solvepnp_ransac_epnp": (C.c_int, [_f32p, _f32p, C.c_int, _f64p, C.c_int, C.c_float,
                                          C.c_double, C.c_uint64, _f64p, _f64p, _i32p, _intp,
                                          _intp, _intp]),
    "ph_solvepnp_ransac_epnp_batched": (C.c_int, [_f32p, _f32p, _i32p, C.c_int, _f64p, C.c_int,
                                                  C.c_float, C.c_double, C.c_uint64, _f64p, _f64p,
                                                  _i32p, _i32p, _i32p, _i32p]),
    "ph_solvepnp_epnp": (C.c_int, [_f32p, _f32p, C.c_int, _f64p, _f64p, _f64p]),
    "ph_ransac_epnp_hypotheses_batched_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                        _f64p, C.c_void_p, C.c_int, C.c_float,
                                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "ph_ransac_subsets": (C.c_int, [C.c_uint64, C.c_int, C.c_int, _i32p]),
    "ph_ransac_epnp_hypotheses_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _f64p,
                                                C.c_void_p, C.c_int, C.c_float, C.c_void_p,
                                                C.c_void_p, C.c_void_p]),
    "ph_set_profiling": (C.c_int, [C.c_int]),
    "ph_knn_last_prepass_timing": (C.c_int, [C.POINTER(C.c_float)]),
    "ph_knn_set_path": (C.c_int, [C.c_int]),
    "ph_knn_last_path": (C.c_int, []),
    "ph_knn_last_timing": (C.c_int, [_f32p, _f32p]),
    "ph_knn_last_timing_overhead": (C.c_int, [_f32p]),
    "ph_valu_popcount_peak": (C.c_int, [C.c_int, _f64p]),
    "ph_valu_fp64_peak": (C.c_int, [C.c_int, _f64p, _f64p]),
    "ph_mfma_fp4_peak": (C.c_int, [C.c_int, _f64p]),
    "ph_ransac_set_solve_path": (C.c_int, [C.c_int]),
    "ph_ransac_last_timing": (C.c_int, [_f32p, _f32p, _f32p]),
}

_lib = None


def load():
    """Loads libposehip.so and declares the prototypes.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch ships its own HIP runtime; when both live in one process the runtime
    # must be loaded once, so let torch load it first if torch is installed.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C poseestimation_amd/csrc` "
            "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != PH_OK:
        raise PosehipError(status, load().ph_last_error_string().decode("utf-8", "replace"))


class Context:
    """A binding of the library to one GPU beside the default one (ph_ctx_create): its own
    stream, workspace and lock.  No counterpart in cv2 or the reference, which know one
    device; worker threads that share a GPU, or that each drive their own, create one each.

        ctx = Context(device)
        with ctx:                # ph_ctx_make_current for this thread, restored on exit
            bf.match(q, t)       # every call of this thread now runs in ctx
        ctx.close()              # ph_ctx_destroy (waits for calls in flight)
    """

    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(load().ph_ctx_create(int(device), C.byref(self._h)))
        self._before = []

    @property
    def handle(self):
        return self._h

    @property
    def device(self):
        return load().ph_ctx_device(self._h)

    def __enter__(self):
        if not self._h:
            raise PosehipError(1, "this context has been closed")
        self._before.append(load().ph_ctx_current())
        check(load().ph_ctx_make_current(self._h))
        return self

    def __exit__(self, *exc):
        load().ph_ctx_make_current(self._before.pop())
        return False

    def close(self):
        if self._h:
            check(load().ph_ctx_destroy(self._h))
            self._h = C.c_void_p()


def init(device=0):
    check(load().ph_init(int(device)))


def shutdown():
    load().ph_shutdown()
