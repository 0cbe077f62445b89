"""GPU suite: the HIP matcher against the CPU oracle, bit-exact, through the C ABI."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import datagen
import oracle_bind as orc
from test_oracle_matcher import KAT, padded_rows, rows

pytestmark = pytest.mark.gpu

INT_MAX = 2147483647


@pytest.fixture(params=["by_size", "matrix_cores"], autouse=True)
def knn_path(request, posehip):
    """Every test runs twice: with the library choosing the distance kernel by problem size
    (small problems take the popcount kernel, large ones the matrix cores) and with the
    matrix-core kernel forced, so that the small, ragged and degenerate cases reach it."""
    mode = {"by_size": 0, "popcount": 1, "matrix_cores": 2}[request.param]
    posehip.check(posehip.load().ph_knn_set_path(mode))
    yield request.param
    posehip.load().ph_knn_set_path(0)


# large problems go to the matrix cores by size: these keep the popcount kernel's long
# streams (lazy key update, several key ranges) covered as well
all_paths = pytest.mark.parametrize("knn_path", ["by_size", "matrix_cores", "popcount"],
                                    indirect=True)


def gpu_knn(lib, q, t, k):
    nq, nt = q.shape[0], t.shape[0]
    idx = np.full((nq, k), 12345, np.int32)
    dist = np.full((nq, k), 12345, np.int32)
    u8, i32 = C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
    lib.check(lib.load().ph_knn_hamming256(q.ctypes.data_as(u8), nq, q.strides[0] if nq else 32,
                                           t.ctypes.data_as(u8), nt, t.strides[0] if nt else 32,
                                           k, idx.ctypes.data_as(i32), dist.ctypes.data_as(i32)))
    return idx, dist


def gpu_match(lib, q, t, cross_check=False, ratio=None):
    nq, nt = q.shape[0], t.shape[0]
    out = (lib.DMatchC * max(nq, 1))()
    n = C.c_int(-1)
    u8 = C.POINTER(C.c_uint8)
    args = (q.ctypes.data_as(u8), nq, q.strides[0] if nq else 32, t.ctypes.data_as(u8), nt,
            t.strides[0] if nt else 32)
    if ratio is not None:
        st = lib.load().ph_knn2_ratio_hamming256(*args, float(ratio), out, C.byref(n))
    elif cross_check:
        st = lib.load().ph_match_crosscheck_hamming256(*args, out, C.byref(n))
    else:
        st = lib.load().ph_match_hamming256(*args, out, C.byref(n))
    lib.check(st)
    return [(m.queryIdx, m.trainIdx, m.imgIdx, m.distance) for m in out[:n.value]]


@pytest.mark.parametrize("case", KAT["cases"], ids=[c["name"] for c in KAT["cases"]])
def test_known_answers(posehip, case):
    q, t = rows(case["query"]), rows(case["train"])
    idx, dist = gpu_knn(posehip, q, t, 2)
    assert idx.tolist() == case["knn2_idx"]
    assert dist.tolist() == case["knn2_dist"]
    assert gpu_match(posehip, q, t) == [(a, b, 0, float(d)) for a, b, d in case["match"]]
    assert gpu_match(posehip, q, t, cross_check=True) == [(a, b, 0, float(d))
                                                          for a, b, d in case["crosscheck"]]


def test_opencv_crosscheck_regression_vector(posehip):
    """The vector of OpenCV's own regression test for the cross-check
    (Features2d_DMatch.issue_11855): exactly one consistent pair; host and device entry."""
    import torch
    reg = KAT["opencv_regression"]
    q, t = padded_rows(reg["sources"]), padded_rows(reg["targets"])
    want = [(a, b, 0, float(d)) for a, b, d in reg["crosscheck"]]
    assert gpu_match(posehip, q, t, cross_check=True) == want
    assert gpu_match(posehip, q, t) == [(a, b, 0, float(d)) for a, b, d in reg["match_without_crosscheck"]]
    d_q, d_t = torch.from_numpy(q).to("cuda:0"), torch.from_numpy(t).to("cuda:0")
    idx = torch.full((2,), 777, dtype=torch.int32, device="cuda:0")
    dist = torch.full((2,), 777, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    posehip.check(posehip.load().ph_match_crosscheck_hamming256_dev(
        d_q.data_ptr(), 2, d_t.data_ptr(), 2, idx.data_ptr(), dist.data_ptr(), None))
    posehip.check(posehip.load().ph_stream_synchronize(None))
    assert idx.cpu().tolist() == [-1, 0] and dist.cpu().tolist() == [INT_MAX, 0]


SHAPES = [(1, 1), (1, 2), (2, 1), (63, 65), (64, 64), (257, 255), (500, 500), (1000, 3),
          (3, 1000), (500, 4099), (2049, 2051)]


@pytest.mark.parametrize("nq,nt", SHAPES)
@pytest.mark.parametrize("k", [1, 2])
def test_random_parity(posehip, nq, nt, k):
    q = datagen.random_descriptors(nq, 11 + nq)
    t = datagen.random_descriptors(nt, 13 + nt)
    idx, dist = gpu_knn(posehip, q, t, k)
    ridx, rdist = orc.knn(q, t, k)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(dist, rdist)


@pytest.mark.parametrize("nq,nt", [(300, 700), (1000, 1000)])
def test_tie_stress_parity(posehip, nq, nt):
    q = datagen.codebook_descriptors(nq, 21)
    t = datagen.codebook_descriptors(nt, 22)
    idx, dist = gpu_knn(posehip, q, t, 2)
    ridx, rdist = orc.knn(q, t, 2)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(dist, rdist)


def test_all_identical_rows(posehip):
    q = np.tile(datagen.random_descriptors(1, 5), (130, 1))
    t = np.tile(q[:1], (777, 1))
    idx, dist = gpu_knn(posehip, q, t, 2)
    assert (idx == np.array([0, 1])).all() and (dist == 0).all()


@pytest.mark.parametrize("nq,nt,seed", [(500, 500, 1), (500, 431, 2), (77, 500, 3), (1500, 1200, 4)])
def test_match_and_crosscheck_parity(posehip, nq, nt, seed):
    q = datagen.random_descriptors(nq, seed)
    t = datagen.perturbed_copy(q, seed + 50)
    if nt <= nq:
        t = t[:nt]
    else:
        t = np.concatenate([t, datagen.random_descriptors(nt - nq, seed + 60)])
    assert gpu_match(posehip, q, t) == orc.match(q, t)
    got = gpu_match(posehip, q, t, cross_check=True)
    want = orc.match(q, t, cross_check=True)
    assert got == want
    assert 0 < len(got) <= nq


def test_crosscheck_tie_stress(posehip):
    q = datagen.codebook_descriptors(300, 31, words=8)
    t = datagen.codebook_descriptors(280, 32, words=8)
    assert gpu_match(posehip, q, t, cross_check=True) == orc.match(q, t, cross_check=True)


@pytest.mark.parametrize("ratio", [0.75, 0.8, 1.0])
def test_ratio_parity(posehip, ratio):
    q = datagen.random_descriptors(800, 41)
    t = np.concatenate([datagen.perturbed_copy(q, 42)[:500], datagen.random_descriptors(900, 43)])
    got = gpu_match(posehip, q, t, ratio=ratio)
    ridx, rdist = orc.knn(q, t, 2)
    assert got == orc.ratio_filter(ridx, rdist, ratio)
    if ratio < 1.0:
        assert 0 < len(got) < 800


def test_empty_and_degenerate(posehip):
    q = datagen.random_descriptors(5, 1)
    e = np.zeros((0, 32), np.uint8)
    assert gpu_match(posehip, q, e) == []
    assert gpu_match(posehip, e, q) == []
    assert gpu_match(posehip, q, e, cross_check=True) == []
    assert gpu_match(posehip, q, e, ratio=0.75) == []
    idx, dist = gpu_knn(posehip, q, e, 2)
    assert (idx == -1).all() and (dist == INT_MAX).all()


def test_strided_rows(posehip):
    """cv::Mat rows may be padded (step > 32)."""
    big = datagen.random_descriptors(400, 77).reshape(200, 64)
    q = big[:, :32]           # stride 64
    t = datagen.random_descriptors(300, 78)
    assert q.strides[0] == 64
    idx, dist = gpu_knn(posehip, q, t, 2)
    ridx, rdist = orc.knn(np.ascontiguousarray(q), t, 2)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)


def test_invalid_arguments(posehip):
    lib = posehip.load()
    q = datagen.random_descriptors(4, 1)
    u8, i32 = C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
    out = np.zeros(16, np.int32)
    st = lib.ph_knn_hamming256(q.ctypes.data_as(u8), 4, 32, q.ctypes.data_as(u8), 4, 32, 3,
                               out.ctypes.data_as(i32), out.ctypes.data_as(i32))
    assert st == 1 and b"k must be" in lib.ph_last_error_string()
    st = lib.ph_knn_hamming256(q.ctypes.data_as(u8), 4, 16, q.ctypes.data_as(u8), 4, 32, 1,
                               out.ctypes.data_as(i32), out.ctypes.data_as(i32))
    assert st == 1
    st = lib.ph_knn_hamming256(None, 4, 32, q.ctypes.data_as(u8), 4, 32, 1,
                               out.ctypes.data_as(i32), out.ctypes.data_as(i32))
    assert st == 1


def test_python_mirror_matches_reference_script_flow(posehip):
    """The flow of src/orb_feature_matching/matcher.py:18-24 through the Python mirror."""
    import poseestimation_amd as pa
    des1 = datagen.random_descriptors(500, 91)
    des2 = datagen.perturbed_copy(des1, 92)
    bf = pa.BFMatcher(pa.NORM_HAMMING, crossCheck=True)
    matches = bf.match(des1, des2)
    matches = sorted(matches, key=lambda x: x.distance)
    want = orc.sort_by_distance(orc.match(des1, des2, cross_check=True))
    assert [(m.queryIdx, m.trainIdx, m.imgIdx, m.distance) for m in matches] == want
    # and the C++ flow of src/cereal_box.cpp:344-356
    plain = pa.BFMatcher(pa.NORM_HAMMING).match(des1, des2)
    good = pa.min_dist_filter(plain)
    wgood = orc.min_dist_filter(orc.match(des1, des2))
    assert [(m.queryIdx, m.trainIdx, m.imgIdx, m.distance) for m in good] == wgood
    rows2 = pa.BFMatcher().knnMatch(des1, des2, k=2)
    ridx, rdist = orc.knn(des1, des2, 2)
    assert [[m.trainIdx for m in r] for r in rows2] == ridx.tolist()
    assert [[m.distance for m in r] for r in rows2] == rdist.astype(float).tolist()


# ---- device-buffer entry points ------------------------------------------------
def _dev(arr):
    import torch
    return torch.from_numpy(np.ascontiguousarray(arr)).to("cuda:0")


def gpu_knn_dev(lib, d_q, d_t, k, base=0, ratio=None):
    import torch
    nq, nt = d_q.shape[0], d_t.shape[0]
    idx = torch.full((nq, k), 777, dtype=torch.int32, device="cuda:0")
    dist = torch.full((nq, k), 777, dtype=torch.int32, device="cuda:0")
    keep = torch.full((nq,), 9, dtype=torch.uint8, device="cuda:0") if ratio is not None else None
    torch.cuda.synchronize()
    lib.check(lib.load().ph_knn_hamming256_dev(
        d_q.data_ptr() if nq else None, nq, d_t.data_ptr() if nt else None, nt, k, base,
        idx.data_ptr(), dist.data_ptr(), keep.data_ptr() if keep is not None else None,
        ratio if ratio is not None else 0.0, None))
    lib.check(lib.load().ph_stream_synchronize(None))
    return idx, dist, keep


@all_paths
def test_full_size_config_bit_exact(posehip):
    """BASELINE configs[1] at full size: 10k x 10k, k=2 + ratio, every row checked."""
    q = datagen.random_descriptors(10000, 2)
    t = datagen.random_descriptors(10000, 102)
    idx, dist, keep = gpu_knn_dev(posehip, _dev(q), _dev(t), 2, ratio=0.75)
    ridx, rdist = orc.knn(q, t, 2, threads=8)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(dist.cpu().numpy(), rdist)
    kept = [m[0] for m in orc.ratio_filter(ridx, rdist, 0.75)]
    assert np.nonzero(keep.cpu().numpy())[0].tolist() == kept


@all_paths
def test_long_train_stream_sampled(posehip):
    """200k queries x 60k train rows (the lazy key-update path, several rows per wave
    slice): 512 sampled query rows checked against the oracle, plus properties that
    hold for every row."""
    q = datagen.random_descriptors(200_000, 7)
    t = np.concatenate([datagen.random_descriptors(59_000, 8), q[:1000]])  # exact hits exist
    idx, dist, _ = gpu_knn_dev(posehip, _dev(q), _dev(t), 2)
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    rows = np.random.default_rng(0).choice(200_000, 512, replace=False)
    rows[:8] = np.arange(8)  # rows with an exact copy in the train set
    ridx, rdist = orc.knn(q[rows], t, 2, threads=8)
    assert np.array_equal(idx[rows], ridx) and np.array_equal(dist[rows], rdist)
    assert (dist[:1000, 0] == 0).all() and (idx[:1000, 0] == 59_000 + np.arange(1000)).all()
    assert (dist[:, 0] <= dist[:, 1]).all() and (idx[:, 0] != idx[:, 1]).all()
    assert idx.min() >= 0 and idx.max() < 60_000
    # reported distances are the true distances of the reported rows
    sel = np.random.default_rng(1).choice(200_000, 4096, replace=False)
    for r in range(2):
        true = np.unpackbits(q[sel] ^ t[idx[sel, r]], axis=1).sum(axis=1)
        assert np.array_equal(true, dist[sel, r])


@pytest.mark.parametrize("n_shards,k", [(2, 2), (3, 1), (8, 2)])
def test_shard_merge(posehip, n_shards, k):
    """Train set split into contiguous shards, searched shard by shard with global
    indices, merged on the GPU: equals the unsharded search (lowest index on ties)."""
    import torch
    from poseestimation_amd.distributed import block_bounds
    q = datagen.codebook_descriptors(700, 51, words=32)
    t = datagen.codebook_descriptors(2000, 52, words=32)
    d_q = _dev(q)
    ci, cd = [], []
    for s in range(n_shards):
        lo, hi = block_bounds(2000, n_shards, s)
        i, d, _ = gpu_knn_dev(posehip, d_q, _dev(t[lo:hi]), k, base=lo)
        ci.append(i)
        cd.append(d)
    all_i, all_d = torch.stack(ci).contiguous(), torch.stack(cd).contiguous()
    out_i = torch.empty((700, k), dtype=torch.int32, device="cuda:0")
    out_d = torch.empty((700, k), dtype=torch.int32, device="cuda:0")
    keep = torch.empty(700, dtype=torch.uint8, device="cuda:0") if k == 2 else None
    torch.cuda.synchronize()
    posehip.check(posehip.load().ph_knn_merge_dev(
        all_i.data_ptr(), all_d.data_ptr(), n_shards, 700, k, out_i.data_ptr(), out_d.data_ptr(),
        keep.data_ptr() if keep is not None else None, 0.8, None))
    posehip.check(posehip.load().ph_stream_synchronize(None))
    ridx, rdist = orc.knn(q, t, k)
    assert np.array_equal(out_i.cpu().numpy(), ridx)
    assert np.array_equal(out_d.cpu().numpy(), rdist)
    if k == 2:
        kept = [m[0] for m in orc.ratio_filter(ridx, rdist, 0.8)]
        assert np.nonzero(keep.cpu().numpy())[0].tolist() == kept


@pytest.mark.parametrize("n_shards,tie_stress", [(2, False), (3, True), (8, False), (8, True)])
def test_crosscheck_shards(posehip, n_shards, tie_stress):
    """Cross-check match with the train set in contiguous shards: every shard's candidates
    (ph_match_crosscheck_hamming256_shard_dev: the query's nearest row within the shard,
    global index, with the flag "the query is that row's nearest query" packed below it),
    folded by ph_knn_merge_dev(k=1) as after the all-gather of a train-sharded job and
    unpacked, equal the unsharded cross-check match of the oracle."""
    import torch
    from poseestimation_amd.distributed import block_bounds
    nq, nt = 1500, 2600
    if tie_stress:
        q = datagen.codebook_descriptors(nq, 55, words=16)
        t = datagen.codebook_descriptors(nt, 56, words=16)
    else:
        q = datagen.random_descriptors(nq, 55)
        t = np.concatenate([datagen.perturbed_copy(q, 57)[:1300], datagen.random_descriptors(1300, 58)])
    d_q, d_t = _dev(q), _dev(t)
    ci, cd = [], []
    for s in range(n_shards):
        lo, hi = block_bounds(nt, n_shards, s)
        i = torch.full((nq, 1), 777, dtype=torch.int32, device="cuda:0")
        d = torch.full((nq, 1), 777, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        posehip.check(posehip.load().ph_match_crosscheck_hamming256_shard_dev(
            d_q.data_ptr(), nq, d_t[lo:hi].data_ptr(), hi - lo, lo, i.data_ptr(), d.data_ptr(), None))
        posehip.check(posehip.load().ph_stream_synchronize(None))
        ci.append(i)
        cd.append(d)
        loc = i.cpu().numpy().ravel()
        assert ((loc == -1) | (((loc >> 1) >= lo) & ((loc >> 1) < hi))).all()
    all_i, all_d = torch.stack(ci).contiguous(), torch.stack(cd).contiguous()
    out_i = torch.empty((nq, 1), dtype=torch.int32, device="cuda:0")
    out_d = torch.empty((nq, 1), dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    posehip.check(posehip.load().ph_knn_merge_dev(all_i.data_ptr(), all_d.data_ptr(), n_shards, nq,
                                                  1, out_i.data_ptr(), out_d.data_ptr(), None, 0.0,
                                                  None))
    posehip.check(posehip.load().ph_match_crosscheck_unpack_dev(out_i.data_ptr(), out_d.data_ptr(),
                                                                nq, None))
    posehip.check(posehip.load().ph_stream_synchronize(None))
    idx, dist = out_i.cpu().numpy().ravel(), out_d.cpu().numpy().ravel()
    assert (dist[idx < 0] == 2147483647).all()
    got = [(int(i), int(idx[i]), 0, float(dist[i])) for i in range(nq) if idx[i] >= 0]
    want = orc.match(q, t, cross_check=True)
    assert got == want and len(got) > 0


def test_crosscheck_train_sharded_world1(posehip):
    """match_crosscheck_train_sharded end to end in a one-rank group on the GPU."""
    import torch
    import torch.distributed as dist
    from poseestimation_amd.distributed import match_crosscheck_train_sharded
    q = datagen.random_descriptors(900, 63)
    t = datagen.perturbed_copy(q, 64)[:700]
    created = False
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29618", rank=0, world_size=1)
        created = True
    try:
        with torch.cuda.stream(torch.cuda.Stream()):
            idx, dst = match_crosscheck_train_sharded(_dev(q), _dev(t), 0)
            torch.cuda.current_stream().synchronize()
    finally:
        if created:
            dist.destroy_process_group()
    idx, dst = idx.cpu().numpy(), dst.cpu().numpy()
    got = [(int(i), int(idx[i]), 0, float(dst[i])) for i in range(900) if idx[i] >= 0]
    assert got == orc.match(q, t, cross_check=True)


def test_train_sharded_world1(posehip):
    """knn_train_sharded end to end in a one-rank group on the GPU."""
    import torch
    import torch.distributed as dist
    from poseestimation_amd.distributed import knn_train_sharded
    q = datagen.random_descriptors(900, 61)
    t = datagen.random_descriptors(1300, 62)
    created = False
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29617", rank=0, world_size=1)
        created = True
    try:
        with torch.cuda.stream(torch.cuda.Stream()):
            idx, dst, keep = knn_train_sharded(_dev(q), _dev(t), 0, 2, ratio=0.75)
            torch.cuda.current_stream().synchronize()
    finally:
        if created:
            dist.destroy_process_group()
    ridx, rdist = orc.knn(q, t, 2)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(dst.cpu().numpy(), rdist)


def test_crosscheck_dev(posehip):
    import torch
    q = datagen.random_descriptors(3000, 71)
    t = datagen.perturbed_copy(q, 72)[:2500]
    d_q, d_t = _dev(q), _dev(t)
    idx = torch.empty(3000, dtype=torch.int32, device="cuda:0")
    dist = torch.empty(3000, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    posehip.check(posehip.load().ph_match_crosscheck_hamming256_dev(
        d_q.data_ptr(), 3000, d_t.data_ptr(), 2500, idx.data_ptr(), dist.data_ptr(), None))
    posehip.check(posehip.load().ph_stream_synchronize(None))
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    got = [(int(i), int(idx[i]), 0, float(dist[i])) for i in range(3000) if idx[i] >= 0]
    assert got == orc.match(q, t, cross_check=True)


@all_paths
def test_train_set_beyond_one_key_range(posehip):
    """More train rows than one 22-bit split index holds (4 194 304): the split cap and
    the cross-split tie-break (equal rows planted on both sides of the boundary)."""
    nt = (1 << 22) + 4099
    t = datagen.random_descriptors(nt, 81)
    q = datagen.random_descriptors(130, 82)
    # identical copies of query rows far apart: lowest index must win, second = next copy
    t[(1 << 22) + 7] = q[0]
    t[123] = q[0]
    t[(1 << 22) - 1] = q[1]
    t[(1 << 22)] = q[1]
    t[nt - 1] = q[2]
    idx, dist, _ = gpu_knn_dev(posehip, _dev(q), _dev(t), 2)
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    ridx, rdist = orc.knn(q, t, 2, threads=8)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    assert idx[0].tolist() == [123, (1 << 22) + 7] and dist[0].tolist() == [0, 0]
    assert idx[1].tolist() == [(1 << 22) - 1, 1 << 22]
    assert idx[2, 0] == nt - 1 and dist[2, 0] == 0


@pytest.mark.parametrize("knn_path,nq", [("matrix_cores", 1_000_000), ("popcount", 65_536)],
                         indirect=["knn_path"])
def test_ten_million_train_rows_unsharded_and_eight_shards(posehip, nq):
    """BASELINE configs[4]'s train set on one GPU: 10 000 000 train rows (1M queries on the
    matrix cores, 65 536 on the popcount kernel), once as one call and once as 8 contiguous
    shards searched with train_index_base and folded by ph_knn_merge_dev -- the single-GPU
    form of the train-sharded job.  96 sampled query rows are checked against the oracle
    on all 10M rows (exact copies planted on both sides of shard boundaries decide
    tie-breaks across shards); both forms must agree on every row."""
    import torch
    from poseestimation_amd.distributed import block_bounds
    nt, n_shards = 10_000_000, 8
    t = datagen.random_descriptors(nt, 5)
    q = datagen.random_descriptors(nq, 1005)
    cut = block_bounds(nt, n_shards, 3)[0]           # first row of shard 3
    rows = np.random.default_rng(3).choice(nq, 96, replace=False)
    t[cut - 1] = q[rows[0]]                          # equal rows either side of a boundary
    t[cut] = q[rows[0]]
    t[nt - 1] = q[rows[1]]                           # last row of the last shard
    t[17] = q[rows[2]]
    t[cut + 5] = q[rows[2]]                          # first copy wins, second is the runner-up
    d_q, d_t = _dev(q), _dev(t)
    idx, dist, keep = gpu_knn_dev(posehip, d_q, d_t, 2, ratio=0.75)
    ci, cd = [], []
    for s in range(n_shards):
        lo, hi = block_bounds(nt, n_shards, s)
        i, d, _ = gpu_knn_dev(posehip, d_q, d_t[lo:hi], 2, base=lo)
        ci.append(i)
        cd.append(d)
    all_i, all_d = torch.stack(ci).contiguous(), torch.stack(cd).contiguous()
    m_i, m_d = torch.empty_like(idx), torch.empty_like(dist)
    m_k = torch.empty_like(keep)
    torch.cuda.synchronize()
    posehip.check(posehip.load().ph_knn_merge_dev(
        all_i.data_ptr(), all_d.data_ptr(), n_shards, nq, 2, m_i.data_ptr(), m_d.data_ptr(),
        m_k.data_ptr(), 0.75, None))
    posehip.check(posehip.load().ph_stream_synchronize(None))
    assert torch.equal(m_i, idx) and torch.equal(m_d, dist) and torch.equal(m_k, keep)
    idx, dist, keep = idx.cpu().numpy(), dist.cpu().numpy(), keep.cpu().numpy()
    ridx, rdist = orc.knn(q[rows], t, 2, threads=8)
    assert np.array_equal(idx[rows], ridx) and np.array_equal(dist[rows], rdist)
    kept = {m[0] for m in orc.ratio_filter(ridx, rdist, 0.75)}
    assert {i for i in range(96) if keep[rows[i]]} == kept
    assert idx[rows[0]].tolist() == [cut - 1, cut] and dist[rows[0]].tolist() == [0, 0]
    assert idx[rows[1], 0] == nt - 1 and dist[rows[1], 0] == 0
    assert idx[rows[2]].tolist() == [17, cut + 5]
    # properties of every row: ordered pairs of distinct rows in range, true distances
    assert (dist[:, 0] <= dist[:, 1]).all() and (idx[:, 0] != idx[:, 1]).all()
    assert idx.min() >= 0 and idx.max() < nt
    sel = np.random.default_rng(4).choice(nq, 4096, replace=False)
    for r in range(2):
        true = np.unpackbits(q[sel] ^ t[idx[sel, r]], axis=1).sum(axis=1)
        assert np.array_equal(true, dist[sel, r])


@all_paths
@pytest.mark.parametrize("nq,nt", [(20_000_000, 100), (70_000_000, 5), (3, 3_000_000)])
def test_extreme_aspect_ratios(posehip, nq, nt):
    """Tens of millions of queries against a handful of train rows, and the reverse: rows
    generated on the GPU, the first and last 64 queries and 256 random ones against the
    oracle (k = 2 + ratio), and no output row left unwritten."""
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    d_q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device=dev, generator=g)
    d_t = torch.randint(0, 256, (nt, 32), dtype=torch.uint8, device=dev, generator=g)
    idx, dist, keep = gpu_knn_dev(posehip, d_q, d_t, 2, ratio=0.8)
    assert int((idx == 777).sum().item()) == 0 and int((keep == 9).sum().item()) == 0
    rows = torch.cat([torch.arange(0, min(nq, 64), device=dev),
                      torch.arange(max(nq - 64, 0), nq, device=dev),
                      torch.randint(0, nq, (256,), device=dev, generator=g)]).unique()
    ri, rd = orc.knn(d_q[rows].cpu().numpy(), d_t.cpu().numpy(), 2, threads=8)
    assert np.array_equal(idx[rows].cpu().numpy(), ri) and np.array_equal(dist[rows].cpu().numpy(), rd)
    want = (ri[:, 1] >= 0) & (rd[:, 0].astype(np.float32) < np.float32(0.8) * rd[:, 1].astype(np.float32))
    assert np.array_equal(keep[rows].cpu().numpy().astype(bool), want)


def test_index_base_and_overflow_guard(posehip):
    q = datagen.random_descriptors(70, 91)
    t = datagen.random_descriptors(300, 92)
    idx, dist, _ = gpu_knn_dev(posehip, _dev(q), _dev(t), 2, base=1_000_000)
    ridx, rdist = orc.knn(q, t, 2)
    assert np.array_equal(idx.cpu().numpy(), ridx + 1_000_000)
    assert np.array_equal(dist.cpu().numpy(), rdist)
    import torch
    out = torch.empty((70, 2), dtype=torch.int32, device="cuda:0")
    st = posehip.load().ph_knn_hamming256_dev(_dev(q).data_ptr(), 70, _dev(t).data_ptr(), 300, 2,
                                              2**31 - 100, out.data_ptr(), out.data_ptr(), None,
                                              0.0, None)
    assert st == 5  # PH_ERR_TOO_LARGE: base + nt would not fit an int32 index


@all_paths
def test_long_stream_k1_and_lazy_path(posehip):
    """k = 1 on a shape that takes the no-split, lazy-update plan (1.1M queries): sampled
    rows against the oracle; planted exact copies must be found at their lowest index."""
    nq, nt = 1_100_000, 9000
    q = datagen.random_descriptors(nq, 17)
    t = datagen.random_descriptors(nt, 18)
    t[5000] = q[10]
    t[7000] = q[10]
    t[8999] = q[nq - 1]
    for k in (1, 2):
        idx, dist, _ = gpu_knn_dev(posehip, _dev(q), _dev(t), k)
        idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
        rows = np.random.default_rng(k).choice(nq, 256, replace=False)
        rows[:2] = [10, nq - 1]
        ridx, rdist = orc.knn(q[rows], t, k, threads=8)
        assert np.array_equal(idx[rows], ridx) and np.array_equal(dist[rows], rdist)
        assert idx[10, 0] == 5000 and dist[10, 0] == 0 and idx[nq - 1, 0] == 8999
        if k == 2:
            assert idx[10, 1] == 7000 and dist[10, 1] == 0


def test_calls_from_two_threads(posehip):
    """The reference calls the matcher from a std::async worker (src/cereal_box.cpp:280):
    entry points must work from any thread and serialise internally."""
    import threading
    q = datagen.random_descriptors(600, 33)
    t = datagen.random_descriptors(900, 34)
    want = orc.knn(q, t, 2)
    errs = []

    def work():
        try:
            for _ in range(20):
                idx, dist = gpu_knn(posehip, q, t, 2)
                assert np.array_equal(idx, want[0]) and np.array_equal(dist, want[1])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=work) for _ in range(3)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errs, errs


def test_shutdown_and_reinit(posehip):
    lib = posehip.load()
    q = datagen.random_descriptors(100, 35)
    lib.ph_shutdown()
    assert lib.ph_device() == -1
    u8, i32 = C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
    out = np.zeros((100, 1), np.int32)
    st = lib.ph_knn_hamming256(q.ctypes.data_as(u8), 100, 32, q.ctypes.data_as(u8), 100, 32, 1,
                               out.ctypes.data_as(i32), out.ctypes.data_as(i32))
    assert st == 4  # PH_ERR_NOT_INITIALIZED: no silent fallback
    posehip.init(0)
    idx, dist = gpu_knn(posehip, q, q, 1)
    assert (idx[:, 0] == np.arange(100)).all() and (dist == 0).all()


# ---- batches of frames in one launch --------------------------------------------
def _frames_reference(q, qo, t, to, k):
    idx = np.full((q.shape[0], k), -1, np.int32)
    dist = np.full((q.shape[0], k), INT_MAX, np.int32)
    for f in range(len(qo) - 1):
        if qo[f + 1] > qo[f]:
            i, d = orc.knn(q[qo[f]:qo[f + 1]], t[to[f]:to[f + 1]], k)
            idx[qo[f]:qo[f + 1]], dist[qo[f]:qo[f + 1]] = i, d
    return idx, dist


@pytest.mark.parametrize("k", [1, 2])
def test_frames_ragged(posehip, k):
    """Ragged frames in one launch, each checked against the oracle run per frame:
    an empty frame, a frame without train rows, one train row with k = 2, sizes
    around the 64-query block and the 4-row group, odd train offsets."""
    import torch
    from poseestimation_amd import distributed as D
    nqs = [500, 0, 64, 65, 1, 300, 777, 129]
    nts = [431, 50, 1, 3, 1000, 0, 2051, 64]
    qo, to = np.concatenate([[0], np.cumsum(nqs)]), np.concatenate([[0], np.cumsum(nts)])
    q = datagen.codebook_descriptors(int(qo[-1]), 41, words=256)   # many equal distances
    t = datagen.codebook_descriptors(int(to[-1]), 42, words=256)
    with torch.cuda.stream(torch.cuda.Stream()):
        idx, dist, keep = D.knn_frames(_dev(q), qo, _dev(t), to, k, ratio=0.8 if k == 2 else None)
        torch.cuda.current_stream().synchronize()
    ridx, rdist = _frames_reference(q, qo, t, to, k)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(dist.cpu().numpy(), rdist)
    if k == 2:
        want = (ridx[:, 1] >= 0) & (rdist[:, 0].astype(np.float32)
                                    < np.float32(0.8) * rdist[:, 1].astype(np.float32))
        assert np.array_equal(keep.cpu().numpy().astype(bool), want)


def test_frames_video_stream(posehip):
    """Frame f against frame f-1 of ONE descriptor buffer (what the reference's loop
    does with descriptors / descriptorsOld, src/cereal_box.cpp:341-344), and a window
    that does not start at row 0."""
    import torch
    from poseestimation_amd import distributed as D
    counts = [500, 480, 500, 499, 350, 500]
    off = np.concatenate([[0], np.cumsum(counts)])
    base = datagen.random_descriptors(500, 43)
    frames = [datagen.perturbed_copy(base, 44 + f, max_flips=30)[:c]
              for f, c in enumerate(counts)]
    buf = np.concatenate(frames)
    d_buf = _dev(buf)
    qo, to = off[1:], off[:-1]      # frames 1..5 query, frames 0..4 train
    with torch.cuda.stream(torch.cuda.Stream()):
        idx, dist, _ = D.knn_frames(d_buf, qo, d_buf, to, 2)
        torch.cuda.current_stream().synchronize()
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    assert (idx[:off[1]] == -1).all() and (dist[:off[1]] == INT_MAX).all()   # frame 0: untouched
    for f in range(1, len(counts)):
        ri, rd = orc.knn(frames[f], frames[f - 1], 2)
        assert np.array_equal(idx[off[f]:off[f + 1]], ri)
        assert np.array_equal(dist[off[f]:off[f + 1]], rd)


@all_paths
def test_frames_equal_single_calls_at_bench_shape(posehip):
    """Four 10k x 10k frames in one launch == four single-frame calls (bit for bit)."""
    import torch
    from poseestimation_amd import distributed as D
    n, F = 10000, 4
    q = datagen.random_descriptors(n * F, 51)
    t = datagen.random_descriptors(n * F, 52)
    d_q, d_t = _dev(q), _dev(t)
    off = np.arange(F + 1) * n
    with torch.cuda.stream(torch.cuda.Stream()):
        idx, dist, keep = D.knn_frames(d_q, off, d_t, off, 2, ratio=0.75)
        torch.cuda.current_stream().synchronize()
    for f in range(F):
        i1, d1, k1 = gpu_knn_dev(posehip, d_q[f * n:(f + 1) * n], d_t[f * n:(f + 1) * n], 2,
                                 ratio=0.75)
        assert torch.equal(idx[f * n:(f + 1) * n], i1) and torch.equal(dist[f * n:(f + 1) * n], d1)
        assert torch.equal(keep[f * n:(f + 1) * n], k1)


def test_frames_invalid_arguments(posehip):
    lib = posehip.load()
    i32 = C.POINTER(C.c_int32)
    bad = np.array([0, 10, 5], np.int32)
    ok = np.array([0, 10, 20], np.int32)
    st = lib.ph_knn_hamming256_frames_dev(1 << 12, bad.ctypes.data_as(i32), 1 << 12,
                                          ok.ctypes.data_as(i32), 2, 2, 1 << 12, 1 << 12, None,
                                          0.0, None)
    assert st == 1 and b"ascending" in lib.ph_last_error_string()
    st = lib.ph_knn_hamming256_frames_dev(1 << 12, ok.ctypes.data_as(i32), 1 << 12,
                                          ok.ctypes.data_as(i32), 2, 3, 1 << 12, 1 << 12, None,
                                          0.0, None)
    assert st == 1
    st = lib.ph_knn_hamming256_frames_dev(None, None, None, None, 0, 2, None, None, None, 0.0,
                                          None)
    assert st == 0   # no frames: nothing to do


def test_calls_on_two_streams_are_ordered(posehip):
    """The workspace is shared, so a call on another stream must wait for the
    previous one: alternate two streams without any host synchronisation between."""
    import torch
    lib = posehip.load()
    qs = [datagen.random_descriptors(3000, 60 + i) for i in range(6)]
    t = datagen.random_descriptors(5000, 70)
    d_t = _dev(t)
    d_qs = [_dev(x) for x in qs]
    outs = [(torch.empty((3000, 2), dtype=torch.int32, device="cuda:0"),
             torch.empty((3000, 2), dtype=torch.int32, device="cuda:0")) for _ in qs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for i, d_q in enumerate(d_qs):
        sp = C.c_void_p(streams[i % 2].cuda_stream)
        posehip.check(lib.ph_knn_hamming256_dev(d_q.data_ptr(), 3000, d_t.data_ptr(), 5000, 2, 0,
                                                outs[i][0].data_ptr(), outs[i][1].data_ptr(),
                                                None, 0.0, sp))
    torch.cuda.synchronize()
    for i, x in enumerate(qs):
        ri, rd = orc.knn(x, t, 2, threads=8)
        assert np.array_equal(outs[i][0].cpu().numpy(), ri)
        assert np.array_equal(outs[i][1].cpu().numpy(), rd)


@pytest.mark.parametrize("tie_stress", [False, True])
def test_crosscheck_frames(posehip, tie_stress):
    """Cross-check match of ragged frames in one launch against the oracle per frame."""
    import torch
    from poseestimation_amd import distributed as D
    nqs = [500, 0, 64, 65, 1, 300, 777, 129]
    nts = [431, 50, 1, 3, 1000, 0, 2051, 64]
    qo, to = np.concatenate([[0], np.cumsum(nqs)]), np.concatenate([[0], np.cumsum(nts)])
    if tie_stress:
        q = datagen.codebook_descriptors(int(qo[-1]), 81, words=64)
        t = datagen.codebook_descriptors(int(to[-1]), 82, words=64)
    else:
        t = datagen.random_descriptors(int(to[-1]), 83)
        q = datagen.random_descriptors(int(qo[-1]), 84)
        q[:431] = datagen.perturbed_copy(t[:431], 85, max_flips=30)   # frame 0 has true matches
    with torch.cuda.stream(torch.cuda.Stream()):
        idx, dist = D.match_crosscheck_frames(_dev(q), qo, _dev(t), to)
        torch.cuda.current_stream().synchronize()
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    for f in range(len(nqs)):
        want_i = np.full(nqs[f], -1, np.int32)
        want_d = np.full(nqs[f], INT_MAX, np.int32)
        if nqs[f] and nts[f]:
            for qi, ti, _, d in orc.match(q[qo[f]:qo[f + 1]], t[to[f]:to[f + 1]], cross_check=True):
                want_i[qi], want_d[qi] = ti, int(d)
        assert np.array_equal(idx[qo[f]:qo[f + 1]], want_i), f
        assert np.array_equal(dist[qo[f]:qo[f + 1]], want_d), f


@pytest.mark.parametrize("F,nq,nt", [(8, 700, 900), (9, 129, 2051), (19, 64, 64)])
def test_frames_uniform_batches(posehip, F, nq, nt):
    """Eight or more equally sized frames take the XCD-contiguous workgroup order
    (also with workgroup counts that are not multiples of 8): every frame against the
    oracle, k = 2 + ratio and the cross-check."""
    import torch
    from poseestimation_amd import distributed as D
    q = datagen.codebook_descriptors(F * nq, 91, words=512)
    t = datagen.codebook_descriptors(F * nt, 92, words=512)
    qo, to = np.arange(F + 1) * nq, np.arange(F + 1) * nt
    with torch.cuda.stream(torch.cuda.Stream()):
        idx, dist, keep = D.knn_frames(_dev(q), qo, _dev(t), to, 2, ratio=0.8)
        cidx, cdist = D.match_crosscheck_frames(_dev(q), qo, _dev(t), to)
        torch.cuda.current_stream().synchronize()
    ridx, rdist = _frames_reference(q, qo, t, to, 2)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(dist.cpu().numpy(), rdist)
    cidx, cdist = cidx.cpu().numpy(), cdist.cpu().numpy()
    for f in range(F):
        want = np.full(nq, -1, np.int32)
        for qi, ti, _, d in orc.match(q[qo[f]:qo[f + 1]], t[to[f]:to[f + 1]], cross_check=True):
            want[qi] = ti
        assert np.array_equal(cidx[qo[f]:qo[f + 1]], want), f


def test_resident_train_collection(posehip):
    """DescriptorMatcher.add / train / match(query): three train images resident on the
    GPU; results equal the oracle on the concatenation with (imgIdx, trainIdx) mapped back,
    for match, knnMatch(k=2) and the cross-check; plus the raw handle API."""
    import poseestimation_amd as pa
    imgs = [datagen.codebook_descriptors(n, 120 + i, words=300) for i, n in enumerate([400, 1, 733])]
    q = datagen.codebook_descriptors(500, 130, words=300)
    cat = np.concatenate(imgs)
    start = np.array([0, 400, 401])

    def split(g):
        i = int(np.searchsorted(start, g, side="right") - 1)
        return i, int(g - start[i])

    bf = pa.BFMatcher(pa.NORM_HAMMING)
    assert bf.empty()
    bf.add(imgs)
    assert not bf.empty() and len(bf.getTrainDescriptors()) == 3
    ridx, rdist = orc.knn(q, cat, 2)
    for _ in range(2):      # second round: the set is already resident
        got = bf.match(q)
        assert [(m.queryIdx, m.imgIdx, m.trainIdx, m.distance) for m in got] == \
            [(i, *split(ridx[i, 0]), float(rdist[i, 0])) for i in range(500)]
    rows = bf.knnMatch(q, 2)
    assert [[(m.imgIdx, m.trainIdx, m.distance) for m in r] for r in rows] == \
        [[(*split(ridx[i, r]), float(rdist[i, r])) for r in range(2)] for i in range(500)]
    cc = pa.BFMatcher(pa.NORM_HAMMING, crossCheck=True)
    cc.add(imgs)
    want = orc.match(q, cat, cross_check=True)
    assert [(m.queryIdx, m.imgIdx, m.trainIdx, m.distance) for m in cc.match(q)] == \
        [(a, *split(b), float(d)) for a, b, _, d in want]
    bf.clear()
    assert bf.empty() and bf.match(q) == []
    # the handle API itself
    lib = posehip.load()
    h = C.c_void_p()
    u8, i32 = C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
    strided = np.zeros((cat.shape[0], 48), np.uint8)
    strided[:, :32] = cat
    posehip.check(lib.ph_db_upload(strided.ctypes.data_as(u8), cat.shape[0], 48, C.byref(h)))
    assert lib.ph_db_rows(h) == cat.shape[0] and lib.ph_db_device_ptr(h)
    idx, dist = np.zeros((500, 2), np.int32), np.zeros((500, 2), np.int32)
    keep = np.zeros(500, np.uint8)
    posehip.check(lib.ph_db_knn_hamming256(h, q.ctypes.data_as(u8), 500, 32, 2, idx.ctypes.data_as(i32),
                                           dist.ctypes.data_as(i32), keep.ctypes.data_as(u8), 0.8))
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    kept = [m[0] for m in orc.ratio_filter(ridx, rdist, 0.8)]
    assert np.nonzero(keep)[0].tolist() == kept
    out = (posehip.DMatchC * 500)()
    n = C.c_int(0)
    posehip.check(lib.ph_ratio_filter(idx.ctypes.data_as(i32), dist.ctypes.data_as(i32), 500, 0.8, out,
                                      C.byref(n)))
    assert [(m.queryIdx, m.trainIdx, m.imgIdx, m.distance) for m in out[:n.value]] == \
        orc.ratio_filter(ridx, rdist, 0.8)
    posehip.check(lib.ph_db_free(h))
    assert lib.ph_db_rows(h) == -1 and lib.ph_db_free(h) == 1    # dead handle: refused, no crash


@all_paths
def test_resident_database_device_queries(posehip):
    """A 50k-row database uploaded once, frames of 2000 device-resident queries matched
    against it (BASELINE configs[3] in small): every call equals ph_knn_hamming256_dev on
    the same buffers and the oracle on sampled rows; the second and third call reuse what
    the first one built for the handle; an empty database and a dead handle are handled."""
    import torch
    lib = posehip.load()
    u8 = C.POINTER(C.c_uint8)
    t = datagen.random_descriptors(50_000, 31)
    h = C.c_void_p()
    posehip.check(lib.ph_db_upload(t.ctypes.data_as(u8), t.shape[0], 32, C.byref(h)))
    d_t = _dev(t)
    for call, nq in enumerate([2000, 2000, 1]):
        q = datagen.random_descriptors(nq, 40 + call)
        q[0] = t[49_999 - call]  # an exact hit near the end of the database
        d_q = _dev(q)
        idx = torch.empty((nq, 2), dtype=torch.int32, device=d_q.device)
        dist = torch.empty_like(idx)
        keep = torch.empty((nq,), dtype=torch.uint8, device=d_q.device)
        posehip.check(lib.ph_db_knn_hamming256_dev(h, d_q.data_ptr(), nq, 2, idx.data_ptr(),
                                                   dist.data_ptr(), keep.data_ptr(),
                                                   C.c_float(0.75), None))
        posehip.check(lib.ph_stream_synchronize(None))
        ridx, rdist, rkeep = gpu_knn_dev(posehip, d_q, d_t, 2, ratio=0.75)
        assert torch.equal(idx, ridx) and torch.equal(dist, rdist) and torch.equal(keep, rkeep)
        rows = np.arange(min(nq, 64))
        oi, od = orc.knn(q[rows], t, 2, threads=8)
        assert np.array_equal(idx.cpu().numpy()[rows], oi) and np.array_equal(dist.cpu().numpy()[rows], od)
        assert idx[0, 0].item() == 49_999 - call and dist[0, 0].item() == 0
    posehip.check(lib.ph_db_free(h))
    assert lib.ph_db_knn_hamming256_dev(h, d_q.data_ptr(), 1, 2, idx.data_ptr(), dist.data_ptr(), None,
                                        C.c_float(0), None) != 0
    empty = C.c_void_p()
    posehip.check(lib.ph_db_upload(None, 0, 32, C.byref(empty)))
    posehip.check(lib.ph_db_knn_hamming256_dev(empty, d_q.data_ptr(), 1, 2, idx.data_ptr(),
                                               dist.data_ptr(), None, C.c_float(0), None))
    posehip.check(lib.ph_stream_synchronize(None))
    assert idx[0].tolist() == [-1, -1] and dist[0].tolist() == [INT_MAX, INT_MAX]
    posehip.check(lib.ph_db_free(empty))


@all_paths
def test_batched_frames_config_per_gpu_blocks(posehip):
    """BASELINE configs[3] as the frame-sharded job deals it: 4096 frames x 2000 descriptors
    against a 50k-row model database, 8 ranks, 512 frames per rank.  Each rank's block is ONE
    ph_db_knn_hamming256_dev call (1 024 000 query rows against the resident database, through
    distributed.knn_frame_block_resident); here the 8 blocks of distributed.frames_of_rank run
    one after the other on GPU 0.  Block 0: 512 sampled rows against the oracle; every block:
    64 sampled rows, planted exact hits, and the properties that hold for every row."""
    import torch
    from poseestimation_amd.distributed import frames_of_rank, knn_frame_block_resident
    lib = posehip.load()
    n_frames, per_frame, world, nt = 4096, 2000, 8, 50_000
    t = datagen.random_descriptors(nt, 300)
    h = C.c_void_p()
    posehip.check(lib.ph_db_upload(t.ctypes.data_as(C.POINTER(C.c_uint8)), nt, 32, C.byref(h)))
    try:
        covered = 0
        with torch.cuda.stream(torch.cuda.Stream()):
            for rank in range(world):
                frames = frames_of_rank(n_frames, world, rank)
                assert len(frames) == 512 and frames[0] == covered
                covered += len(frames)
                rows = len(frames) * per_frame
                # this block's frames, generated per block (the whole job is 262 MB of rows)
                q = datagen.random_descriptors(rows, 301 + rank)
                first = np.arange(len(frames)) * per_frame        # first row of every frame:
                hit = (np.asarray(frames) * 12 + 5) % nt          # an exact copy of a database row
                q[first] = t[hit]
                d_q = _dev(q)
                idx, dist, keep = knn_frame_block_resident(h, d_q, 2, ratio=0.75)
                torch.cuda.current_stream().synchronize()
                idx, dist, keep = idx.cpu().numpy(), dist.cpu().numpy(), keep.cpu().numpy()
                assert (dist[first, 0] == 0).all() and (idx[first, 0] == hit).all()
                assert keep[first].all()
                assert (dist[:, 0] <= dist[:, 1]).all() and (idx[:, 0] != idx[:, 1]).all()
                assert idx.min() >= 0 and idx.max() < nt
                n_sample = 512 if rank == 0 else 64
                pick = np.random.default_rng(rank).choice(rows, n_sample, replace=False)
                oi, od = orc.knn(q[pick], t, 2, threads=8)
                assert np.array_equal(idx[pick], oi) and np.array_equal(dist[pick], od)
                kept = {m[0] for m in orc.ratio_filter(oi, od, 0.75)}
                assert set(np.nonzero(keep[pick])[0].tolist()) == kept
                sel = np.random.default_rng(100 + rank).choice(rows, 4096, replace=False)
                for r in range(2):   # reported distances are those of the reported rows
                    true = np.unpackbits(q[sel] ^ t[idx[sel, r]], axis=1).sum(axis=1)
                    assert np.array_equal(true, dist[sel, r])
                del d_q
        assert covered == n_frames
    finally:
        posehip.check(lib.ph_db_free(h))
