"""Randomised differential run of the matcher entry points against the CPU oracle
(shapes around the block, group and split boundaries, ragged and uniform frame batches,
tie-heavy data).  Not part of the test suite; run it after kernel changes:
    python tests/tools/fuzz_matcher.py [seconds] [seed]
POSEHIP_KNN_PATH=matrix (or popcount) in the environment pins the distance kernel; by default
the library picks by size, which at these shapes is mostly the popcount kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import datagen, oracle_bind as orc
import poseestimation_amd as pa
from poseestimation_amd import distributed as D
pa._lib.init(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
INT_MAX = 2147483647


def size():
    base = int(rng.choice([1, 2, 3, 4, 5, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 500, 511, 512, 513, 1000, 1023, 1024,
                           1025, 2047, 2048, 4099, 8191, 8193]))
    return max(0, base + int(rng.integers(-2, 3))) if rng.random() < 0.5 else int(rng.integers(0, 6000))


def descs(n, ties):
    seed = int(rng.integers(1 << 30))
    if n == 0:
        return np.zeros((0, 32), np.uint8)
    return datagen.codebook_descriptors(n, seed, words=int(rng.choice([2, 16, 300]))) if ties else datagen.random_descriptors(n, seed)


t_end, n_cases = time.time() + budget, 0
last = time.time()
while time.time() < t_end:
    if time.time() - last > 20:
        print({'progress': n_cases}, flush=True)
        last = time.time()
    ties = rng.random() < 0.5
    mode = rng.choice(["host", "frames", "uniform"])
    k = int(rng.choice([1, 2]))
    if mode == "host":
        nq, nt = size(), size()
        q, t = descs(nq, ties), descs(nt, ties)
        bf = pa.BFMatcher()
        idx, dist = bf.knnMatchArrays(q, t, k)
        if nq and nt:
            ri, rd = orc.knn(q, t, k, threads=8)
            assert np.array_equal(idx, ri) and np.array_equal(dist, rd), ("knn", nq, nt, k)
        got = pa.BFMatcher(pa.NORM_HAMMING, crossCheck=True).match(q, t)
        want = orc.match(q, t, cross_check=True) if nq and nt else []
        assert [(m.queryIdx, m.trainIdx, m.imgIdx, m.distance) for m in got] == want, ("cc", nq, nt)
    else:
        F = int(rng.integers(1, 20))
        if mode == "uniform":
            a, b = max(1, size() // 4), max(1, size() // 4)
            nqs, nts = [a] * F, [b] * F
        else:
            nqs, nts = [size() // 4 for _ in range(F)], [size() // 4 for _ in range(F)]
        qo, to = np.concatenate([[0], np.cumsum(nqs)]), np.concatenate([[0], np.cumsum(nts)])
        q, t = descs(int(qo[-1]), ties), descs(int(to[-1]), ties)
        if qo[-1] == 0 or to[-1] == 0:
            continue
        d_q, d_t = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
        idx, dist, _ = D.knn_frames(d_q, qo, d_t, to, k)
        cidx, cdist = D.match_crosscheck_frames(d_q, qo, d_t, to)
        stream.synchronize()
        idx, dist, cidx = idx.cpu().numpy(), dist.cpu().numpy(), cidx.cpu().numpy()
        for f in range(F):
            qs, ts = slice(qo[f], qo[f + 1]), slice(to[f], to[f + 1])
            if nqs[f] == 0:
                continue
            if nts[f] == 0:
                assert (idx[qs] == -1).all() and (dist[qs] == INT_MAX).all() and (cidx[qs] == -1).all()
                continue
            ri, rd = orc.knn(q[qs], t[ts], k, threads=4)
            assert np.array_equal(idx[qs], ri) and np.array_equal(dist[qs], rd), (mode, F, f, nqs[f], nts[f], k)
            want = np.full(nqs[f], -1, np.int32)
            for qi, ti, _, d in orc.match(q[qs], t[ts], cross_check=True):
                want[qi] = ti
            assert np.array_equal(cidx[qs], want), ("ccframes", mode, F, f, nqs[f], nts[f])
    n_cases += 1
print({"cases": n_cases, "seconds": budget, "ok": True})
