"""GPU suite: contexts (ph_ctx_*), the C ABI's way to more than one GPU -- or more than one
concurrent caller -- in one process.  One GPU is what a test box has, so the contexts here
share device 0; what is checked is that they share nothing else: two threads, each bound to
its own context, issue matcher and pose calls of changing sizes at the same time, and every
result equals the oracle's."""
import ctypes as C
import threading

import numpy as np
import pytest

import datagen
import oracle_bind as orc

pytestmark = pytest.mark.gpu

u8p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
f32p, f64p = C.POINTER(C.c_float), C.POINTER(C.c_double)


def _knn_host(posehip, q, t, k):
    idx = np.full((q.shape[0], k), 12345, np.int32)
    dist = np.full((q.shape[0], k), 12345, np.int32)
    posehip.check(posehip.load().ph_knn_hamming256(q.ctypes.data_as(u8p), q.shape[0], 32,
                                                   t.ctypes.data_as(u8p), t.shape[0], 32, k,
                                                   idx.ctypes.data_as(i32p), dist.ctypes.data_as(i32p)))
    return idx, dist


def _crosscheck_host(posehip, q, t):
    out = (posehip.DMatchC * q.shape[0])()
    n = C.c_int(-1)
    posehip.check(posehip.load().ph_match_crosscheck_hamming256(
        q.ctypes.data_as(u8p), q.shape[0], 32, t.ctypes.data_as(u8p), t.shape[0], 32, out, C.byref(n)))
    return [(m.queryIdx, m.trainIdx, m.imgIdx, m.distance) for m in out[:n.value]]


def _ransac_host(posehip, obj, img, iters):
    K = np.ascontiguousarray(datagen.CAMERA_K, np.float64).reshape(9)
    rvec, tvec = np.zeros(3), np.zeros(3)
    inl = np.zeros(obj.shape[0], np.int32)
    n_inl, found, it = C.c_int(0), C.c_int(0), C.c_int(0)
    posehip.check(posehip.load().ph_solvepnp_ransac_epnp(
        obj.ctypes.data_as(f32p), img.ctypes.data_as(f32p), obj.shape[0], K.ctypes.data_as(f64p), iters,
        8.0, 0.99, orc.CV_RNG_DEFAULT, rvec.ctypes.data_as(f64p), tvec.ctypes.data_as(f64p),
        inl.ctypes.data_as(i32p), C.byref(n_inl), C.byref(found), C.byref(it)))
    return found.value, rvec, tvec, inl[:n_inl.value].tolist(), it.value


def _workload(seed):
    """A worker's calls: sizes change from call to call (the workspace is regrown and reused),
    small ones take the popcount kernel, large ones the matrix cores."""
    rng = np.random.default_rng(seed)
    jobs = []
    for i in range(10):
        nq, nt = int(rng.integers(200, 4000)), int(rng.integers(200, 4000))
        q = datagen.random_descriptors(nq, seed * 100 + i)
        t = np.concatenate([datagen.perturbed_copy(q, seed * 100 + 50 + i)[:min(nq, nt) // 2],
                            datagen.random_descriptors(nt - min(nq, nt) // 2, seed * 100 + 70 + i)])
        jobs.append(("knn", q, t))
        jobs.append(("cc", q, t))
    jobs.append(("knn", datagen.random_descriptors(9000, seed + 900), datagen.random_descriptors(8000, seed + 901)))
    for i in range(3):
        obj, img, _, _, _ = datagen.pnp_scene(n=500 + 37 * i, seed=seed * 10 + i)
        jobs.append(("pnp", obj, img))
    return jobs


def test_two_contexts_on_one_gpu_called_from_two_threads(posehip):
    lib = posehip.load()
    ctxs = [C.c_void_p(), C.c_void_p()]
    for c in ctxs:
        posehip.check(lib.ph_ctx_create(0, C.byref(c)))
        assert c.value and lib.ph_ctx_device(c) == 0
    assert ctxs[0].value != ctxs[1].value
    results, errors = [None, None], []
    start = threading.Barrier(2)

    def worker(w):
        try:
            posehip.check(lib.ph_ctx_make_current(ctxs[w]))
            assert lib.ph_ctx_current() == ctxs[w].value and lib.ph_device() == 0
            jobs, out = _workload(11 + w), []
            start.wait()
            for rep in range(3):                  # the same jobs three times: later rounds run
                for kind, a, b in jobs:           # against the other thread's different sizes
                    if kind == "knn":
                        out.append(_knn_host(posehip, a, b, 2))
                    elif kind == "cc":
                        out.append(_crosscheck_host(posehip, a, b))
                    else:
                        out.append(_ransac_host(posehip, a, b, 100))
            results[w] = (jobs, out)
            posehip.check(lib.ph_ctx_make_current(None))
        except Exception as e:                    # surfaces in the main thread below
            errors.append((w, repr(e)))

    threads = [threading.Thread(target=worker, args=(w,)) for w in range(2)]
    for t in threads:
        t.start()
    # meanwhile the main thread keeps using the default context
    q, t0 = datagen.random_descriptors(700, 1), datagen.random_descriptors(900, 2)
    for _ in range(20):
        assert lib.ph_ctx_current() is None
        mi, md = _knn_host(posehip, q, t0, 2)
    for t in threads:
        t.join(timeout=100)
    assert not errors, errors
    ri, rd = orc.knn(q, t0, 2)
    assert np.array_equal(mi, ri) and np.array_equal(md, rd)
    for w in range(2):
        jobs, out = results[w]
        assert len(out) == 3 * len(jobs)
        want = []
        for kind, a, b in jobs:
            if kind == "knn":
                want.append(orc.knn(a, b, 2, threads=8))
            elif kind == "cc":
                want.append(orc.match(a, b, cross_check=True))
            else:
                want.append(orc.solvepnp_ransac(a, b, datagen.CAMERA_K, iterations=100))
        for i, got in enumerate(out):
            kind, a, b = jobs[i % len(jobs)]
            ref = want[i % len(jobs)]
            if kind == "knn":
                assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), (w, i)
            elif kind == "cc":
                assert got == ref and len(got) > 0, (w, i)
            else:
                found, rvec, tvec, inl, it = got
                ok, r_rvec, r_tvec, r_inl, r_it = ref
                assert found == int(ok) and inl == r_inl.tolist() and it == r_it, (w, i)
                assert np.abs(rvec - r_rvec).max() < 1e-9 and np.abs(tvec - r_tvec).max() < 1e-9
    for c in ctxs:
        posehip.check(lib.ph_ctx_destroy(c))
        assert lib.ph_ctx_device(c) == -1
    assert lib.ph_ctx_destroy(ctxs[0]) == 1                      # dead handle: refused, no crash


def test_context_lifetime_and_shared_train_set(posehip):
    """A resident train set uploaded through one context serves another context on the same
    GPU; a context outlives ph_shutdown of the default one; a thread whose current context
    has been destroyed is told so."""
    import torch
    lib = posehip.load()
    a, b = C.c_void_p(), C.c_void_p()
    posehip.check(lib.ph_ctx_create(0, C.byref(a)))
    posehip.check(lib.ph_ctx_create(0, C.byref(b)))
    t = datagen.random_descriptors(30_000, 5)
    q = datagen.random_descriptors(3000, 6)
    q[0] = t[29_999]
    posehip.check(lib.ph_ctx_make_current(a))
    h = C.c_void_p()
    posehip.check(lib.ph_db_upload(t.ctypes.data_as(u8p), t.shape[0], 32, C.byref(h)))
    posehip.check(lib.ph_ctx_make_current(b))
    idx = np.zeros((3000, 2), np.int32)
    dist = np.zeros((3000, 2), np.int32)
    posehip.check(lib.ph_db_knn_hamming256(h, q.ctypes.data_as(u8p), 3000, 32, 2, idx.ctypes.data_as(i32p),
                                           dist.ctypes.data_as(i32p), None, 0.0))
    ri, rd = orc.knn(q, t, 2, threads=8)
    assert np.array_equal(idx, ri) and np.array_equal(dist, rd) and idx[0, 0] == 29_999
    # device entry point on b's own stream (NULL) with torch-owned buffers
    d_q = torch.from_numpy(q).to("cuda:0")
    d_i = torch.empty((3000, 2), dtype=torch.int32, device="cuda:0")
    d_d = torch.empty_like(d_i)
    torch.cuda.synchronize()
    posehip.check(lib.ph_db_knn_hamming256_dev(h, d_q.data_ptr(), 3000, 2, d_i.data_ptr(), d_d.data_ptr(),
                                               None, 0.0, None))
    posehip.check(lib.ph_stream_synchronize(None))
    assert np.array_equal(d_i.cpu().numpy(), ri)
    # destroying the owner releases the train set
    posehip.check(lib.ph_ctx_destroy(a))
    assert lib.ph_db_rows(h) == -1
    assert lib.ph_db_knn_hamming256(h, q.ctypes.data_as(u8p), 3000, 32, 2, idx.ctypes.data_as(i32p),
                                    dist.ctypes.data_as(i32p), None, 0.0) == 1
    # b survives a shutdown of the default context ...
    lib.ph_shutdown()
    gi, gd = _knn_host(posehip, q[:500], t[:700], 2)
    oi, od = orc.knn(q[:500], t[:700], 2)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    # ... and once b is gone, this thread's calls are refused until it picks another context
    posehip.check(lib.ph_ctx_destroy(b))
    assert lib.ph_ctx_current() is None          # destroy un-binds the destroying thread
    assert lib.ph_stream_synchronize(None) == 4  # no default context either: not initialised
    posehip.init(0)
    gi, gd = _knn_host(posehip, q[:500], t[:700], 2)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    # a thread keeps a dead context current: refused with a message, no crash
    c = C.c_void_p()
    posehip.check(lib.ph_ctx_create(0, C.byref(c)))
    seen = {}

    def holder():
        posehip.check(lib.ph_ctx_make_current(c))
        ready.set()
        gone.wait(timeout=60)
        seen["status"] = lib.ph_stream_synchronize(None)
        seen["msg"] = lib.ph_last_error_string()
        lib.ph_ctx_make_current(None)

    ready, gone = threading.Event(), threading.Event()
    th = threading.Thread(target=holder)
    th.start()
    ready.wait(timeout=60)
    posehip.check(lib.ph_ctx_destroy(c))
    gone.set()
    th.join(timeout=60)
    assert seen["status"] == 1 and b"destroyed" in seen["msg"]
    assert lib.ph_ctx_make_current(c) == 1       # cannot bind a dead handle
    assert lib.ph_ctx_create(99, C.byref(c)) == 1 and not c.value


def test_python_context_wrapper(posehip):
    """poseestimation_amd.Context: the mirror's calls run in the context while it is entered,
    in the default context again afterwards, and a closed context cannot be entered."""
    import poseestimation_amd as pa
    lib = posehip.load()
    q, t = datagen.random_descriptors(400, 21), datagen.perturbed_copy(datagen.random_descriptors(400, 21), 22)
    want = orc.match(q, t, cross_check=True)
    ctx = pa.Context(0)
    assert ctx.device == 0 and lib.ph_ctx_current() is None
    with ctx:
        assert lib.ph_ctx_current() == ctx.handle.value
        got = pa.BFMatcher(pa.NORM_HAMMING, crossCheck=True).match(q, t)
        with pa.Context(0) as inner:          # nesting restores the outer binding
            assert lib.ph_ctx_current() == inner.handle.value
        assert lib.ph_ctx_current() == ctx.handle.value
        inner.close()
    assert lib.ph_ctx_current() is None
    assert [(m.queryIdx, m.trainIdx, m.imgIdx, m.distance) for m in got] == want
    ctx.close()
    ctx.close()                               # idempotent
    with pytest.raises(posehip.PosehipError):
        ctx.__enter__()


def test_settings_and_profiling_are_per_context(posehip):
    """The k-NN path setting and the profiling switch belong to a context: forcing the matrix
    cores and recording events in one context changes nothing in another or in the default."""
    lib = posehip.load()
    q, t = datagen.random_descriptors(600, 31), datagen.random_descriptors(700, 32)
    want = orc.knn(q, t, 2)
    a, b = C.c_void_p(), C.c_void_p()
    posehip.check(lib.ph_ctx_create(0, C.byref(a)))
    posehip.check(lib.ph_ctx_create(0, C.byref(b)))
    ms = C.c_float(-1)
    try:
        posehip.check(lib.ph_ctx_make_current(a))
        posehip.check(lib.ph_knn_set_path(2))
        posehip.check(lib.ph_set_profiling(1))
        gi, gd = _knn_host(posehip, q, t, 2)
        assert lib.ph_knn_last_path() == 1                      # matrix cores, as forced in a
        posehip.check(lib.ph_knn_last_timing(C.byref(ms), None))
        assert ms.value > 0
        assert np.array_equal(gi, want[0]) and np.array_equal(gd, want[1])
        posehip.check(lib.ph_ctx_make_current(b))
        gi, gd = _knn_host(posehip, q, t, 2)
        assert lib.ph_knn_last_path() == 0                      # b picks by size: popcount kernel
        assert lib.ph_knn_last_timing(C.byref(ms), None) == 1   # b never enabled profiling
        assert np.array_equal(gi, want[0]) and np.array_equal(gd, want[1])
        posehip.check(lib.ph_ctx_make_current(None))
        gi, gd = _knn_host(posehip, q, t, 2)
        assert lib.ph_knn_last_path() == 0 and lib.ph_knn_last_timing(C.byref(ms), None) == 1
        posehip.check(lib.ph_ctx_make_current(a))
        assert lib.ph_knn_last_path() == 1                      # a's own record is untouched
    finally:
        lib.ph_ctx_make_current(None)
        lib.ph_ctx_destroy(a)
        lib.ph_ctx_destroy(b)


def test_contexts_created_used_and_destroyed_concurrently(posehip):
    """Four threads each create a context, run a few calls in it and destroy it, twenty times
    over, while the main thread keeps calling through the default context: the registry and
    the per-context locks keep every result right and nothing is left behind."""
    lib = posehip.load()
    q, t = datagen.random_descriptors(300, 41), datagen.random_descriptors(350, 42)
    want = orc.knn(q, t, 2)
    want_cc = orc.match(q, t[:300], cross_check=True)
    errors, done = [], []

    def worker(w):
        try:
            for rep in range(20):
                c = C.c_void_p()
                posehip.check(lib.ph_ctx_create(0, C.byref(c)))
                posehip.check(lib.ph_ctx_make_current(c))
                gi, gd = _knn_host(posehip, q, t, 2)
                assert np.array_equal(gi, want[0]) and np.array_equal(gd, want[1])
                assert _crosscheck_host(posehip, q, t[:300]) == want_cc
                if rep % 2:        # destroy while current, or after handing the thread back
                    posehip.check(lib.ph_ctx_make_current(None))
                posehip.check(lib.ph_ctx_destroy(c))
                assert lib.ph_ctx_current() is None and lib.ph_ctx_device(c) == -1
            done.append(w)
        except Exception as e:
            errors.append((w, repr(e)))

    threads = [threading.Thread(target=worker, args=(w,)) for w in range(4)]
    for th in threads:
        th.start()
    while any(th.is_alive() for th in threads):
        gi, gd = _knn_host(posehip, q, t, 2)
        assert np.array_equal(gi, want[0]) and np.array_equal(gd, want[1])
    for th in threads:
        th.join(timeout=100)
    assert not errors, errors
    assert sorted(done) == [0, 1, 2, 3]


def test_context_stress_program():
    """tests/context_stress.cpp, a C++ host on the C ABI alone: four threads create, bind, use
    and destroy ten contexts each (own and shared resident train sets, matcher and pose calls,
    per-context settings) while two more threads call through the default context and poll
    the registry; every result must equal the default context's.  The same program runs under
    ASan + UBSan with the library's host code instrumented (tests/tools/build_sanitized.sh)."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "context_stress")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe, "4", "10"], capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line == {"threads": 4, "contexts_per_thread": 10, "failures": 0}
