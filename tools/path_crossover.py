"""Popcount kernel against matrix-core kernel over problem sizes (device-resident inputs,
whole call: pre-pass + distance kernel + merge), to place the size rule that picks one.
    python tools/path_crossover.py
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poseestimation_amd import _lib  # noqa: E402

_lib.init(0)
lib = _lib.load()
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
sptr = C.c_void_p(stream.cuda_stream)
i32p = C.POINTER(C.c_int32)


def time_call(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def single(nq, nt, k=2):
    q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device=dev)
    t = torch.randint(0, 256, (nt, 32), dtype=torch.uint8, device=dev)
    idx = torch.empty((nq, k), dtype=torch.int32, device=dev)
    dist = torch.empty((nq, k), dtype=torch.int32, device=dev)
    out = {}
    for name, mode in (("popcount", 1), ("matrix", 2)):
        _lib.check(lib.ph_knn_set_path(mode))
        out[name] = time_call(lambda: _lib.check(lib.ph_knn_hamming256_dev(
            q.data_ptr(), nq, t.data_ptr(), nt, k, 0, idx.data_ptr(), dist.data_ptr(), None,
            C.c_float(0), sptr)))
    _lib.check(lib.ph_knn_set_path(0))
    print(f"single {nq:>7} x {nt:>7} k={k}: popcount {out['popcount']:8.1f} us  matrix {out['matrix']:8.1f} us"
          f"  -> {'matrix' if out['matrix'] < out['popcount'] else 'popcount'} "
          f"({out['popcount'] / out['matrix']:.2f}x)", flush=True)


def frames(F, nq, nt, k=2):
    q = torch.randint(0, 256, (F * nq, 32), dtype=torch.uint8, device=dev)
    t = torch.randint(0, 256, (F * nt, 32), dtype=torch.uint8, device=dev)
    idx = torch.empty((F * nq, k), dtype=torch.int32, device=dev)
    dist = torch.empty((F * nq, k), dtype=torch.int32, device=dev)
    qo = (np.arange(F + 1, dtype=np.int64) * nq).astype(np.int32)
    to = (np.arange(F + 1, dtype=np.int64) * nt).astype(np.int32)
    out = {}
    for name, mode in (("popcount", 1), ("matrix", 2)):
        _lib.check(lib.ph_knn_set_path(mode))
        out[name] = time_call(lambda: _lib.check(lib.ph_knn_hamming256_frames_dev(
            q.data_ptr(), qo.ctypes.data_as(i32p), t.data_ptr(), to.ctypes.data_as(i32p), F, k,
            idx.data_ptr(), dist.data_ptr(), None, C.c_float(0), sptr)))
    _lib.check(lib.ph_knn_set_path(0))
    print(f"frames {F:>4} x ({nq:>6} x {nt:>6}) k={k}: popcount {out['popcount']:8.1f} us  matrix "
          f"{out['matrix']:8.1f} us  -> {'matrix' if out['matrix'] < out['popcount'] else 'popcount'} "
          f"({out['popcount'] / out['matrix']:.2f}x)", flush=True)


for n in (500, 1000, 1500, 2000, 3000, 4000, 6000, 10000, 20000):
    single(n, n)
for nq, nt in ((512, 20000), (512, 100000), (20000, 512), (100000, 512), (2000, 50000), (50000, 2000),
               (200000, 60000)):
    single(nq, nt)
single(10000, 10000, k=1)
for F, nq, nt in ((64, 500, 500), (256, 500, 500), (16, 1000, 1000), (16, 2000, 2000), (64, 2000, 2000),
                  (8, 4000, 4000), (16, 10000, 10000), (4, 10000, 10000), (64, 128, 4000), (64, 4000, 128)):
    frames(F, nq, nt)


def resident(nq, nt, k=2):
    """A resident train set (its FP4 image is kept with the handle) against the plain call."""
    q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device=dev)
    t_host = np.random.default_rng(1).integers(0, 256, size=(nt, 32), dtype=np.uint8)
    t = torch.from_numpy(t_host).to(dev)
    idx = torch.empty((nq, k), dtype=torch.int32, device=dev)
    dist = torch.empty((nq, k), dtype=torch.int32, device=dev)
    h = C.c_void_p()
    _lib.check(lib.ph_db_upload(t_host.ctypes.data_as(C.POINTER(C.c_uint8)), nt, 32, C.byref(h)))
    plain = time_call(lambda: _lib.check(lib.ph_knn_hamming256_dev(
        q.data_ptr(), nq, t.data_ptr(), nt, k, 0, idx.data_ptr(), dist.data_ptr(), None, C.c_float(0), sptr)))
    res = time_call(lambda: _lib.check(lib.ph_db_knn_hamming256_dev(
        h, q.data_ptr(), nq, k, idx.data_ptr(), dist.data_ptr(), None, C.c_float(0), sptr)))
    _lib.check(lib.ph_db_free(h))
    print(f"resident {nq:>7} x {nt:>8}: plain call {plain:8.1f} us, resident set {res:8.1f} us "
          f"({plain / res:.2f}x)", flush=True)


for nq, nt in ((2000, 50000), (10000, 10000), (2000, 1000000), (100000, 50000)):
    resident(nq, nt)
