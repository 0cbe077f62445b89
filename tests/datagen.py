"""Seeded synthetic descriptor sets shared by the tests and the benchmark."""
import numpy as np


def random_descriptors(n, seed):
    """iid uniform bytes, uint8 [n, 32]."""
    return np.random.default_rng(seed).integers(0, 256, size=(n, 32), dtype=np.uint8)


def perturbed_copy(desc, seed, max_flips=40, shuffle=True):
    """A permuted copy of `desc` with 0..max_flips random bit flips per row, so
    that true correspondences exist (cross-check and the min-distance filter
    have something to keep)."""
    rng = np.random.default_rng(seed)
    out = desc.copy()
    n = out.shape[0]
    bits = np.unpackbits(out, axis=1)
    for i in range(n):
        k = int(rng.integers(0, max_flips + 1))
        pos = rng.choice(256, size=k, replace=False)
        bits[i, pos] ^= 1
    out = np.packbits(bits, axis=1)
    if shuffle:
        out = out[rng.permutation(n)]
    return np.ascontiguousarray(out)


def codebook_descriptors(n, seed, words=16):
    """Tie-stress set: every row drawn from a tiny codebook, so most distances
    are equal and the lowest-index rule decides nearly every match."""
    rng = np.random.default_rng(seed)
    book = rng.integers(0, 256, size=(words, 32), dtype=np.uint8)
    return np.ascontiguousarray(book[rng.integers(0, words, size=n)])


def numpy_knn(q, t, k):
    """Independent brute force in numpy: stable argsort gives lowest index on ties."""
    qb = np.unpackbits(q, axis=1).astype(np.int32)
    tb = np.unpackbits(t, axis=1).astype(np.int32)
    d = qb @ (1 - tb).T + (1 - qb) @ tb.T
    order = np.argsort(d, axis=1, kind="stable")[:, :k]
    idx = np.full((q.shape[0], k), -1, np.int32)
    dist = np.full((q.shape[0], k), 2147483647, np.int32)
    kk = order.shape[1]
    idx[:, :kk] = order
    dist[:, :kk] = np.take_along_axis(d, order, axis=1)
    return idx, dist
