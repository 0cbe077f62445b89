"""GPU test of the allocation-free k-NN entry point (ph_knn_hamming256_noalloc_dev): same
results as ph_knn_hamming256_dev, bit for bit, on two calls in a row that share the
library's fixed scratch array, the second with a smaller input; no device memory is taken
by the calls; sizes beyond the array are refused."""
import numpy as np
import pytest

import datagen
import oracle_bind as orc

pytestmark = pytest.mark.gpu

PH_ERR_TOO_LARGE = 5
SCRATCH_BYTES = 32 << 20


def _dev(arr):
    import torch
    return torch.from_numpy(np.ascontiguousarray(arr)).to("cuda:0")


def _call(posehip, name, d_q, d_t, k, base, ratio):
    import torch
    nq, nt = d_q.shape[0], d_t.shape[0]
    idx = torch.full((nq, k), 777, dtype=torch.int32, device="cuda:0")
    dist = torch.full((nq, k), 777, dtype=torch.int32, device="cuda:0")
    keep = torch.full((nq,), 9, dtype=torch.uint8, device="cuda:0") if ratio is not None else None
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    st = getattr(posehip.load(), name)(
        d_q.data_ptr() if nq else None, nq, d_t.data_ptr() if nt else None, nt, k, base,
        idx.data_ptr(), dist.data_ptr(), keep.data_ptr() if keep is not None else None,
        ratio if ratio is not None else 0.0, None)
    posehip.check(st)
    posehip.check(posehip.load().ph_stream_synchronize(None))
    free_after = torch.cuda.mem_get_info()[0]
    return idx, dist, keep, free_before - free_after


# (path mode, first call, second smaller call): popcount kernel on a small pair, the matrix
# cores by size on a large pair, and the matrix cores forced onto ragged small shapes
CASES = [
    (1, (1500, 3001), (257, 999)),
    (0, (9000, 12000), (6100, 5003)),
    (2, (700, 2051), (129, 300)),
]


@pytest.mark.parametrize("mode,first,second", CASES)
@pytest.mark.parametrize("k", [1, 2])
def test_noalloc_equals_allocating_entry_point(posehip, mode, first, second, k):
    import torch
    lib = posehip.load()
    posehip.check(lib.ph_knn_set_path(mode))
    try:
        for call, (nq, nt) in enumerate((first, second)):
            q = datagen.codebook_descriptors(nq, 300 + nq, words=64) if mode == 1 else \
                datagen.random_descriptors(nq, 300 + nq)
            t = datagen.codebook_descriptors(nt, 400 + nt, words=64) if mode == 1 else \
                datagen.random_descriptors(nt, 400 + nt)
            d_q, d_t = _dev(q), _dev(t)
            ratio = 0.8 if k == 2 else None
            base = 1000 * call
            ri, rd, rk, _ = _call(posehip, "ph_knn_hamming256_dev", d_q, d_t, k, base, ratio)
            path_ref = lib.ph_knn_last_path()
            gi, gd, gk, taken = _call(posehip, "ph_knn_hamming256_noalloc_dev", d_q, d_t, k, base,
                                      ratio)
            assert lib.ph_knn_last_path() == path_ref
            assert torch.equal(gi, ri) and torch.equal(gd, rd)
            if k == 2:
                assert torch.equal(gk, rk)
            assert taken == 0, "the allocation-free entry point took device memory"
            if nq * nt <= 40_000_000:  # and against the oracle where that takes seconds
                oi, od = orc.knn(q, t, k, threads=8)
                assert np.array_equal(gi.cpu().numpy(), oi + base)
                assert np.array_equal(gd.cpu().numpy(), od)
    finally:
        lib.ph_knn_set_path(0)


def test_noalloc_degenerate_and_limits(posehip):
    import torch
    lib = posehip.load()
    q, t = datagen.random_descriptors(70, 1), datagen.random_descriptors(1, 2)
    e = np.zeros((0, 32), np.uint8)
    # no train rows: every slot "no match"; a single train row: second slot empty
    gi, gd, gk, _ = _call(posehip, "ph_knn_hamming256_noalloc_dev", _dev(q), _dev(e), 2, 0, 0.75)
    assert (gi == -1).all() and (gd == 2147483647).all() and (gk == 0).all()
    gi, gd, _, _ = _call(posehip, "ph_knn_hamming256_noalloc_dev", _dev(q), _dev(t), 2, 0, None)
    ri, rd = orc.knn(q, t, 2)
    assert np.array_equal(gi.cpu().numpy(), ri) and np.array_equal(gd.cpu().numpy(), rd)
    _call(posehip, "ph_knn_hamming256_noalloc_dev", _dev(e), _dev(t), 1, 0, None)
    # beyond the fixed scratch: refused with a message naming the need, nothing launched.
    # 300k train rows need a 38 MB image; 4.5M queries need 36 MB of keys even in one split.
    out = torch.empty((4_500_000, 2), dtype=torch.int32, device="cuda:0")
    big_t = torch.zeros((300_000, 32), dtype=torch.uint8, device="cuda:0")
    big_q = torch.zeros((4_500_000, 32), dtype=torch.uint8, device="cuda:0")
    for mode, d_q, d_t in ((2, big_q[:1000], big_t), (1, big_q, big_t[:8192])):
        posehip.check(lib.ph_knn_set_path(mode))
        try:
            st = lib.ph_knn_hamming256_noalloc_dev(d_q.data_ptr(), d_q.shape[0], d_t.data_ptr(),
                                                   d_t.shape[0], 2, 0, out.data_ptr(),
                                                   out.data_ptr(), None, 0.0, None)
        finally:
            lib.ph_knn_set_path(0)
        assert st == PH_ERR_TOO_LARGE
        msg = lib.ph_last_error_string().decode()
        assert "scratch" in msg and str(SCRATCH_BYTES) in msg
    st = lib.ph_knn_hamming256_noalloc_dev(big_q.data_ptr(), 10, big_t.data_ptr(), 10, 3, 0,
                                           out.data_ptr(), out.data_ptr(), None, 0.0, None)
    assert st == 1  # PH_ERR_INVALID_ARG: k


@pytest.mark.parametrize("mode,nq,nt", [(1, 3000, 5000), (2, 20000, 30000)])
def test_noalloc_under_graph_capture(posehip, mode, nq, nt):
    """One of the callers the entry point is for: the call is captured into a HIP graph
    (nothing may allocate or synchronise under capture), the graph is replayed on new
    queries in the same buffers, and the replay equals the allocating entry point."""
    import ctypes as C
    import torch
    lib = posehip.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    d_q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device=dev, generator=g)
    d_t = torch.randint(0, 256, (nt, 32), dtype=torch.uint8, device=dev, generator=g)
    idx = torch.zeros((nq, 2), dtype=torch.int32, device=dev)
    dist = torch.zeros_like(idx)
    keep = torch.zeros(nq, dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream()
    posehip.check(lib.ph_knn_set_path(mode))
    try:
        def call(stream_handle):
            return lib.ph_knn_hamming256_noalloc_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 2, 0,
                                                     idx.data_ptr(), dist.data_ptr(), keep.data_ptr(),
                                                     0.8, C.c_void_p(stream_handle))
        posehip.check(call(s.cuda_stream))      # outside capture first: code objects get loaded
        s.synchronize()
        idx.zero_()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            status = call(torch.cuda.current_stream().cuda_stream)
        posehip.check(status)
        torch.cuda.synchronize()
        assert int(idx.abs().sum().item()) == 0          # captured, not run
        d_q.copy_(torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device=dev, generator=g))
        graph.replay()
        torch.cuda.synchronize()
        ri, rd, rk, _ = _call(posehip, "ph_knn_hamming256_dev", d_q, d_t, 2, 0, 0.8)
        assert torch.equal(idx, ri) and torch.equal(dist, rd) and torch.equal(keep, rk)
    finally:
        lib.ph_knn_set_path(0)


def test_noalloc_call_does_not_leave_stale_profiling_events(posehip):
    """With profiling on, a k-NN call records events around its kernels; the allocation-free
    entry point records none, so after it the timing query must say so instead of pairing
    the earlier call's events with the later call's kernel."""
    import ctypes as C
    lib = posehip.load()
    d_q, d_t = _dev(datagen.random_descriptors(600, 1)), _dev(datagen.random_descriptors(900, 2))
    a, b = C.c_float(-1), C.c_float(-1)
    posehip.check(lib.ph_set_profiling(1))
    try:
        posehip.check(lib.ph_knn_set_path(1))
        _call(posehip, "ph_knn_hamming256_dev", d_q, d_t, 2, 0, None)          # popcount kernel, profiled
        posehip.check(lib.ph_knn_last_timing(C.byref(a), C.byref(b)))
        assert a.value > 0
        posehip.check(lib.ph_knn_set_path(2))
        _call(posehip, "ph_knn_hamming256_noalloc_dev", d_q, d_t, 2, 0, None)  # matrix cores, no events
        assert lib.ph_knn_last_path() == 1
        assert lib.ph_knn_last_timing(C.byref(a), C.byref(b)) == 1
        assert b"no profiled" in lib.ph_last_error_string()
        assert lib.ph_knn_last_prepass_timing(C.byref(a)) == 1
    finally:
        lib.ph_knn_set_path(0)
        lib.ph_set_profiling(0)


def test_noalloc_makes_no_allocating_hip_call():
    """The same two-calls-in-a-row check with every allocator entry point of the HIP runtime
    interposed and COUNTED (tests/noalloc_probe.cpp: the executable defines hipMalloc, hipFree,
    hipHostMalloc, ... itself, so the library's calls land in it): zero allocating or freeing
    calls while the allocation-free entry point runs, on either distance kernel, and results
    equal to the allocating entry point's.  A free-memory delta cannot see pooled or sub-page
    allocations; a call count can.  Runs in a process of its own (no torch allocator)."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "noalloc_probe")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["ok"] is True
    assert line["allocating_calls_of_the_control"] > 0      # the interposition is live
    assert line["noalloc_calls"] == 6
    assert line["allocations_during_them"] == 0 and line["frees_during_them"] == 0
    assert line["equal_to_allocating_entry_point"] is True
