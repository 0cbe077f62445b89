"""Randomised differential run of the matcher's device entry points against the CPU oracle:
every k-NN row, ratio verdict and cross-check winner must equal the oracle's, over random
shapes (around the kernels' block sizes and up to tens of thousands of rows), descriptor
families (iid, tiny codebooks = ties everywhere, perturbed copies = true matches), both
distance kernels, index bases, ragged frame batches, train shards + merge and the
allocation-free entry point.  Not part of the test suite:
    python tests/tools/fuzz_matcher_dev.py [seconds] [seed]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import datagen
import oracle_bind as orc
import poseestimation_amd as pa
from poseestimation_amd import distributed as D

pa._lib.init(0)
lib = pa._lib.load()
check = pa._lib.check
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
INT_MAX = 2147483647
side_stream = torch.cuda.Stream()
EDGES = [1, 2, 3, 4, 5, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1025]


def size(big):
    r = rng.random()
    if r < 0.45:
        return int(rng.choice(EDGES))
    if r < 0.85 or big <= 3000:
        return int(rng.integers(1, min(big, 3000)))
    return int(rng.integers(3000, big))


def descriptors(n, family):
    s = int(rng.integers(1 << 30))
    if family == 0:
        return datagen.random_descriptors(n, s)
    if family == 1:
        return datagen.codebook_descriptors(n, s, words=int(rng.choice([1, 2, 8, 64, 512])))
    # near-duplicates of a few base rows: distances cluster, many ties at small distances
    base = datagen.random_descriptors(int(rng.choice([1, 4, 32])), s)
    out = base[rng.integers(0, base.shape[0], size=n)].copy()
    flips = rng.integers(0, 256, size=(n, 3))
    on = rng.random((n, 3)) < 0.5
    for c in range(3):
        rows = np.nonzero(on[:, c])[0]
        out[rows, flips[rows, c] // 8] ^= (1 << (flips[rows, c] % 8)).astype(np.uint8)
    return out


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def knn_dev(fn, d_q, d_t, nq, nt, k, base, ratio):
    idx = torch.full((nq, k), 777, dtype=torch.int32, device=dev)
    dist = torch.full((nq, k), 777, dtype=torch.int32, device=dev)
    keep = torch.full((nq,), 9, dtype=torch.uint8, device=dev) if ratio is not None else None
    torch.cuda.synchronize()
    check(fn(d_q.data_ptr(), nq, d_t.data_ptr(), nt, k, base, idx.data_ptr(), dist.data_ptr(),
             keep.data_ptr() if keep is not None else None, ratio if ratio is not None else 0.0, None))
    check(lib.ph_stream_synchronize(None))
    return idx.cpu().numpy(), dist.cpu().numpy(), keep.cpu().numpy() if keep is not None else None


def want_keep(ridx, rdist, ratio):
    return (ridx[:, 1] >= 0) & (rdist[:, 0].astype(np.float32) < np.float32(ratio) * rdist[:, 1].astype(np.float32))


def crosscheck_rows(q, t):
    wi = np.full(q.shape[0], -1, np.int32)
    wd = np.full(q.shape[0], INT_MAX, np.int32)
    for qi, ti, _, d in orc.match(q, t, cross_check=True):
        wi[qi], wd[qi] = ti, int(d)
    return wi, wd


def case_single():
    nq, nt = size(40000), size(40000)
    fam = int(rng.integers(3))
    q, t = descriptors(nq, fam), descriptors(nt, fam)
    if fam == 0 and rng.random() < 0.5 and nt >= 2:       # plant exact and near copies
        m = min(nq, nt) // 2
        t[rng.permutation(nt)[:m]] = q[:m]
    k = int(rng.choice([1, 2]))
    base = int(rng.choice([0, 0, 1, 12345, 2_000_000_000 - nt]))
    ratio = float(rng.choice([0.5, 0.75, 0.8, 1.0])) if k == 2 and rng.random() < 0.6 else None
    d_q, d_t = to_dev(q), to_dev(t)
    fn = lib.ph_knn_hamming256_dev
    what = "knn_dev"
    if rng.random() < 0.25:
        fn, what = lib.ph_knn_hamming256_noalloc_dev, "knn_noalloc_dev"
    try:
        idx, dist, keep = knn_dev(fn, d_q, d_t, nq, nt, k, base, ratio)
    except RuntimeError as e:
        if what != "knn_noalloc_dev" or "scratch" not in str(e):
            raise
        idx, dist, keep = knn_dev(lib.ph_knn_hamming256_dev, d_q, d_t, nq, nt, k, base, ratio)
    ridx, rdist = orc.knn(q, t, k, threads=16)
    ridx = np.where(ridx >= 0, ridx + base, ridx)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist), (what, nq, nt, k, base, fam)
    if ratio is not None:
        assert np.array_equal(keep.astype(bool), want_keep(ridx, rdist, ratio)), (what, nq, nt, ratio, fam)
    if rng.random() < 0.5:
        ci = torch.full((nq,), 777, dtype=torch.int32, device=dev)
        cd = torch.full((nq,), 777, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        check(lib.ph_match_crosscheck_hamming256_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, ci.data_ptr(),
                                                     cd.data_ptr(), None))
        check(lib.ph_stream_synchronize(None))
        wi, wd = crosscheck_rows(q, t)
        assert np.array_equal(ci.cpu().numpy(), wi) and np.array_equal(cd.cpu().numpy(), wd), ("crosscheck", nq, nt, fam)
    return nq * nt


def case_frames():
    F = int(rng.choice([1, 2, 3, 8, 9, 17, 40]))
    uniform = rng.random() < 0.4
    if uniform:
        nqs, nts = [size(1500)] * F, [size(1500)] * F
    else:
        nqs = [0 if rng.random() < 0.1 else size(1500) for _ in range(F)]
        nts = [0 if rng.random() < 0.1 else size(1500) for _ in range(F)]
    qo, to = np.concatenate([[0], np.cumsum(nqs)]), np.concatenate([[0], np.cumsum(nts)])
    if qo[-1] == 0 or to[-1] == 0:
        return 0
    fam = int(rng.integers(3))
    q, t = descriptors(int(qo[-1]), fam), descriptors(int(to[-1]), fam)
    k = int(rng.choice([1, 2]))
    ratio = 0.8 if k == 2 and rng.random() < 0.5 else None
    d_q, d_t = to_dev(q), to_dev(t)
    torch.cuda.synchronize()
    with torch.cuda.stream(side_stream):
        idx, dist, keep = D.knn_frames(d_q, qo, d_t, to, k, ratio=ratio)
        cidx, cdist = D.match_crosscheck_frames(d_q, qo, d_t, to)
    torch.cuda.synchronize()
    idx, dist, cidx, cdist = idx.cpu().numpy(), dist.cpu().numpy(), cidx.cpu().numpy(), cdist.cpu().numpy()
    pairs = 0
    for f in range(F):
        a, b, c, d = int(qo[f]), int(qo[f + 1]), int(to[f]), int(to[f + 1])
        if b == a:
            continue
        if d == c:
            assert (idx[a:b] == -1).all() and (dist[a:b] == INT_MAX).all() and (cidx[a:b] == -1).all(), ("frames-empty", f)
            continue
        ri, rd = orc.knn(q[a:b], t[c:d], k, threads=16)
        assert np.array_equal(idx[a:b], ri) and np.array_equal(dist[a:b], rd), ("frames", F, f, nqs[f], nts[f], k, fam)
        if ratio is not None:
            assert np.array_equal(keep[a:b].cpu().numpy().astype(bool), want_keep(ri, rd, ratio)), ("frames-keep", f)
        wi, wd = crosscheck_rows(q[a:b], t[c:d])
        assert np.array_equal(cidx[a:b], wi) and np.array_equal(cdist[a:b], wd), ("frames-cc", F, f, nqs[f], nts[f], fam)
        pairs += (b - a) * (d - c)
    return 2 * pairs


def case_shards():
    nq, nt = size(6000), max(size(20000), 8)
    S = int(rng.choice([2, 3, 5, 8]))
    fam = int(rng.integers(3))
    q, t = descriptors(nq, fam), descriptors(nt, fam)
    k = int(rng.choice([1, 2]))
    d_q, d_t = to_dev(q), to_dev(t)
    ci, cd, xi, xd = [], [], [], []
    for s in range(S):
        lo, hi = D.block_bounds(nt, S, s)
        i = torch.full((nq, k), 777, dtype=torch.int32, device=dev)
        d = torch.full((nq, k), 777, dtype=torch.int32, device=dev)
        i1 = torch.full((nq, 1), 777, dtype=torch.int32, device=dev)
        d1 = torch.full((nq, 1), 777, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        check(lib.ph_knn_hamming256_dev(d_q.data_ptr(), nq, d_t[lo:hi].data_ptr() if hi > lo else None, hi - lo, k,
                                        lo, i.data_ptr(), d.data_ptr(), None, 0.0, None))
        check(lib.ph_match_crosscheck_hamming256_shard_dev(d_q.data_ptr(), nq,
                                                           d_t[lo:hi].data_ptr() if hi > lo else None, hi - lo, lo,
                                                           i1.data_ptr(), d1.data_ptr(), None))
        check(lib.ph_stream_synchronize(None))
        ci.append(i); cd.append(d); xi.append(i1); xd.append(d1)
    out_i = torch.empty((nq, k), dtype=torch.int32, device=dev)
    out_d = torch.empty((nq, k), dtype=torch.int32, device=dev)
    x_i = torch.empty((nq, 1), dtype=torch.int32, device=dev)
    x_d = torch.empty((nq, 1), dtype=torch.int32, device=dev)
    all_i, all_d = torch.stack(ci).contiguous(), torch.stack(cd).contiguous()
    axi, axd = torch.stack(xi).contiguous(), torch.stack(xd).contiguous()
    torch.cuda.synchronize()
    check(lib.ph_knn_merge_dev(all_i.data_ptr(), all_d.data_ptr(), S, nq, k, out_i.data_ptr(), out_d.data_ptr(),
                               None, 0.0, None))
    check(lib.ph_knn_merge_dev(axi.data_ptr(), axd.data_ptr(), S, nq, 1, x_i.data_ptr(), x_d.data_ptr(), None,
                               0.0, None))
    check(lib.ph_match_crosscheck_unpack_dev(x_i.data_ptr(), x_d.data_ptr(), nq, None))
    check(lib.ph_stream_synchronize(None))
    ri, rd = orc.knn(q, t, k, threads=16)
    assert np.array_equal(out_i.cpu().numpy(), ri) and np.array_equal(out_d.cpu().numpy(), rd), ("shards", nq, nt, S, k, fam)
    wi, wd = crosscheck_rows(q, t)
    assert np.array_equal(x_i.cpu().numpy().ravel(), wi) and np.array_equal(x_d.cpu().numpy().ravel(), wd), \
        ("shards-cc", nq, nt, S, fam)
    return 2 * nq * nt


t_end, last = time.time() + budget, time.time()
n_pairs, n_cases = 0, {"single": 0, "frames": 0, "shards": 0}
while time.time() < t_end:
    if time.time() - last > 20:
        print({"progress_pairs": n_pairs, "cases": n_cases}, flush=True)
        last = time.time()
    check(lib.ph_knn_set_path(int(rng.choice([0, 1, 2]))))
    kind = str(rng.choice(["single", "single", "frames", "shards"]))
    n_pairs += {"single": case_single, "frames": case_frames, "shards": case_shards}[kind]()
    n_cases[kind] += 1
check(lib.ph_knn_set_path(0))
print({"seed": seed, "cases": n_cases, "pairs": n_pairs, "ok": True})
