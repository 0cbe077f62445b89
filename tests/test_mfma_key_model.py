"""CPU model of the matrix-core matcher's arithmetic (poseestimation_amd/csrc/knn_mfma.hip),
checked against the oracle: bits as FP4 +-4, the accumulator as ordering key, running keys
aged once per pair of 32-row tiles, decoding back to (distance, row).  It pins the ALGORITHM
on the CPU; the kernel itself is checked bit for bit on the GPU (test_matcher_gpu.py)."""
import numpy as np
import pytest

import datagen
import oracle_bind as orc

NONE = np.float32(-65536.0)


def fp4_values(rows):
    """uint8[n,32] -> float32[n,256]: -4 for a set bit, +4 for a clear one (E2M1 0xE / 0x6)."""
    bits = np.unpackbits(rows, axis=1, bitorder="little")
    return np.where(bits == 1, np.float32(-4.0), np.float32(4.0))


def model_knn(q, t, k):
    qa, ta = fp4_values(q), fp4_values(t)
    nq, nt = q.shape[0], t.shape[0]
    n_tiles = (nt + 31) // 32
    m1 = np.full(nq, NONE, np.float32)
    m2 = np.full(nq, NONE, np.float32)
    for tile in range(n_tiles):
        rows = ta[32 * tile:32 * tile + 32]
        valid = rows.shape[0]
        # accumulator = C + 2^10 * dot, C = 2^22 + 31 - row (+ 32 in even tiles); float32 throughout
        dot = (qa @ rows.T).astype(np.float32)                       # exact: |dot| <= 4096
        c = np.float32(4194304.0) + np.float32(31.0) - np.arange(valid, dtype=np.float32)
        if tile % 2 == 0:
            c = c + np.float32(32.0)
            m1 = m1 + np.float32(64.0)                               # a new pair of tiles begins
            m2 = m2 + np.float32(64.0)
        acc = (np.float32(1024.0) * dot + c[None, :]).astype(np.float32)
        assert (acc == np.round(acc)).all() and acc.max() < 2 ** 24  # integers, exact in fp32
        both = np.concatenate([m1[:, None], m2[:, None], acc], axis=1)
        both.sort(axis=1)
        m1, m2 = both[:, -1].copy(), both[:, -2].copy()
    top = 64 * ((n_tiles - 1) >> 1) + 63
    idx = np.full((nq, k), -1, np.int32)
    dist = np.full((nq, k), 2147483647, np.int32)
    for r, m in enumerate((m1, m2)[:k]):
        ok = m >= 0
        u = m[ok].astype(np.int64)
        dist[ok, r] = 256 - (u >> 15)
        idx[ok, r] = top - (u & 32767)
    return idx, dist


@pytest.mark.parametrize("nq,nt", [(5, 1), (7, 31), (9, 32), (11, 33), (40, 64), (33, 65), (64, 1000)])
@pytest.mark.parametrize("k", [1, 2])
@pytest.mark.parametrize("ties", [False, True])
def test_key_model_equals_oracle(nq, nt, k, ties):
    gen = (lambda n, s: datagen.codebook_descriptors(n, s, words=3)) if ties else datagen.random_descriptors
    q, t = gen(nq, 50 + nq), gen(nt, 60 + nt)
    idx, dist = model_knn(q, t, k)
    ridx, rdist = orc.knn(q, t, k)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)


def test_key_model_extremes():
    """Distance 0 and distance 256 (complement rows), all rows identical: lowest index wins."""
    q = datagen.random_descriptors(4, 1)
    t = np.concatenate([np.repeat(q[:1], 70, axis=0), ~q[:1], q[1:2]])
    idx, dist = model_knn(q, t, 2)
    ridx, rdist = orc.knn(q, t, 2)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    assert idx[0].tolist() == [0, 1] and dist[0].tolist() == [0, 0]
    only = model_knn(q[:1], ~q[:1], 2)
    assert only[0].tolist() == [[0, -1]] and only[1].tolist() == [[256, 2147483647]]
