"""Randomised differential run of the k-NN entry point at LARGE shapes (hundreds to a
thousand tiles per split, several splits, tie-heavy data) against the CPU oracle on sampled
query rows; complements tests/tools/fuzz_matcher.py, whose shapes are small.
    python tests/tools/fuzz_matcher_big.py [seconds] [seed]
POSEHIP_KNN_PATH=matrix / popcount pins the distance kernel."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import datagen  # noqa: E402
import oracle_bind as orc  # noqa: E402
from poseestimation_amd import _lib  # noqa: E402

_lib.init(0)
lib = _lib.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
t_end, n_cases, last = time.time() + budget, 0, time.time()
paths = [0, 0]
while time.time() < t_end:
    if time.time() - last > 20:
        print({"progress": n_cases}, flush=True)
        last = time.time()
    nq = int(rng.choice([257, 511, 512, 513, 1000, 2049, 5000, 10000, 20001])) + int(rng.integers(0, 3))
    nt = int(rng.choice([1024, 5000, 10000, 32767, 32768, 32769, 65536, 65537, 70001, 131072 + 31]))
    nt += int(rng.integers(-2, 3))
    k = int(rng.choice([1, 2]))
    ties = rng.random() < 0.5
    seed = int(rng.integers(1 << 30))
    if ties:
        q = datagen.codebook_descriptors(nq, seed, words=int(rng.choice([2, 16, 300])))
        t = datagen.codebook_descriptors(nt, seed + 1, words=int(rng.choice([2, 16, 300])))
    else:
        q, t = datagen.random_descriptors(nq, seed), datagen.random_descriptors(nt, seed + 1)
    d_q, d_t = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
    idx = torch.empty((nq, k), dtype=torch.int32, device=dev)
    dist = torch.empty((nq, k), dtype=torch.int32, device=dev)
    base = int(rng.choice([0, 7, 1 << 20]))
    _lib.check(lib.ph_knn_hamming256_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, k, base, idx.data_ptr(),
                                         dist.data_ptr(), None, C.c_float(0), None))
    _lib.check(lib.ph_stream_synchronize(None))
    paths[lib.ph_knn_last_path()] += 1
    rows = np.unique(np.concatenate([[0, nq - 1], rng.integers(0, nq, 96)]))
    ri, rd = orc.knn(q[rows], t, k, threads=8)
    gi, gd = idx.cpu().numpy()[rows], dist.cpu().numpy()[rows]
    assert np.array_equal(gi, ri + base) and np.array_equal(gd, rd), ("knn", nq, nt, k, ties, seed, base)
    n_cases += 1
print({"cases": n_cases, "popcount_kernel": paths[0], "matrix_cores": paths[1], "seconds": budget, "ok": True})
