"""Host logic of libposehip.so that needs no GPU: the cv::RNG sample stream, the Lowe ratio
filter and the reference's min-distance filter, against the oracle; and the refusal of
every compute entry point before ph_init (no CPU fallback)."""
import ctypes as C
import os

import numpy as np

import datagen
import oracle_bind as orc
from poseestimation_amd import _lib


def test_ransac_subsets_follow_cv_rng():
    lib = _lib.load()
    for n, h, seed in [(5, 40, orc.CV_RNG_DEFAULT), (6, 300, orc.CV_RNG_DEFAULT), (2000, 4096, orc.CV_RNG_DEFAULT),
                       (500, 100, 12345), (64, 64, 0)]:
        got = np.zeros((h, 5), np.int32)
        _lib.check(lib.ph_ransac_subsets(seed, n, h, got.ctypes.data_as(C.POINTER(C.c_int32))))
        want = orc.ransac_subsets(n, h, rng_state=seed)
        assert np.array_equal(got, want)
        assert all(len(set(r)) == 5 for r in got.tolist()) and got.min() >= 0 and got.max() < n
    assert lib.ph_ransac_subsets(1, 4, 1, got.ctypes.data_as(C.POINTER(C.c_int32))) == 1   # < 5 points


def test_ratio_and_min_dist_filters():
    lib = _lib.load()
    q = datagen.codebook_descriptors(400, 1, words=40)
    t = datagen.codebook_descriptors(300, 2, words=40)
    idx, dist = orc.knn(q, t, 2)
    idx[7, 1], dist[7, 1] = -1, 2147483647      # a row with one neighbour only is never kept
    i32 = C.POINTER(C.c_int32)
    for ratio in (0.5, 0.75, 1.0, 1.5):
        out = (_lib.DMatchC * 400)()
        n = C.c_int(-1)
        _lib.check(lib.ph_ratio_filter(idx.ctypes.data_as(i32), dist.ctypes.data_as(i32), 400, ratio, out,
                                       C.byref(n)))
        got = [(m.queryIdx, m.trainIdx, m.imgIdx, m.distance) for m in out[:n.value]]
        assert got == [m for m in orc.ratio_filter(idx, dist, ratio) if m[0] != 7 or idx[7, 1] >= 0]
        assert all(m[0] != 7 for m in got)
    matches = orc.match(q, t)
    buf = (_lib.DMatchC * len(matches))(*[_lib.DMatchC(*m) for m in matches])
    out = (_lib.DMatchC * len(matches))()
    n = C.c_int(-1)
    _lib.check(lib.ph_min_dist_filter(buf, len(matches), out, C.byref(n)))
    assert [(m.queryIdx, m.trainIdx, m.imgIdx, m.distance) for m in out[:n.value]] == orc.min_dist_filter(matches)
    _lib.check(lib.ph_min_dist_filter(buf, 0, out, C.byref(n)))
    assert n.value == 0


def test_compute_entry_points_refuse_without_init():
    """Before ph_init (and here there is no GPU to bind to) every compute entry point must
    return PH_ERR_NOT_INITIALIZED: the product has no CPU fallback."""
    lib = _lib.load()
    if lib.ph_device() >= 0:      # a GPU suite in the same process has initialised the library
        return
    q = datagen.random_descriptors(8, 1)
    u8, i32 = C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
    out = np.zeros((8, 2), np.int32)
    assert lib.ph_knn_hamming256(q.ctypes.data_as(u8), 8, 32, q.ctypes.data_as(u8), 8, 32, 2,
                                 out.ctypes.data_as(i32), out.ctypes.data_as(i32)) == 4
    h = C.c_void_p()
    assert lib.ph_db_upload(q.ctypes.data_as(u8), 8, 32, C.byref(h)) == 4 and not h.value
    off = np.array([0, 8], np.int32)
    assert lib.ph_knn_hamming256_frames_dev(1 << 12, off.ctypes.data_as(i32), 1 << 12, off.ctypes.data_as(i32),
                                            1, 2, 1 << 12, 1 << 12, None, 0.0, None) == 4
    obj, img = np.zeros((8, 3), np.float32), np.zeros((8, 2), np.float32)
    K = np.ascontiguousarray(datagen.CAMERA_K).reshape(9)
    r = np.zeros(3)
    fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
    found = C.c_int(0)
    assert lib.ph_solvepnp_ransac_epnp(obj.ctypes.data_as(fp), img.ctypes.data_as(fp), 8, K.ctypes.data_as(dp),
                                       100, 8.0, 0.99, 1, r.ctypes.data_as(dp), r.ctypes.data_as(dp), None,
                                       None, C.byref(found), None) == 4
    assert b"ph_init" in lib.ph_last_error_string()


def test_bench_refuses_more_ranks_than_gpus():
    """`bench.py --gpus N` started bare launches its own ranks -- and exits non-zero, before
    any GPU call, when the node has fewer than N GPUs (here: none)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"],
                         capture_output=True, text=True, env=env, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert out.returncode == 2 and "--gpus 2" in out.stderr
