"""CPU suite: the oracle's matcher restatement against hand-computed known answers
and against an independent numpy brute force.  (The reference ships no fixture
for this path, so these pin the restatement, not OpenCV itself.)"""
import json
import os

import numpy as np
import pytest

import datagen
import oracle_bind as orc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "matcher_kat.json")


def ones(n):
    """Descriptor with n leading one-bits."""
    bits = np.zeros(256, np.uint8)
    bits[:n] = 1
    return np.packbits(bits)


def rows(ns):
    return np.stack([ones(n) for n in ns]) if len(ns) else np.zeros((0, 32), np.uint8)


with open(GOLDEN) as f:
    KAT = json.load(f)


@pytest.mark.parametrize("case", KAT["cases"], ids=[c["name"] for c in KAT["cases"]])
def test_known_answers(case):
    q, t = rows(case["query"]), rows(case["train"])
    idx, dist = orc.knn(q, t, 2)
    assert idx.tolist() == case["knn2_idx"]
    assert dist.tolist() == case["knn2_dist"]
    idx1, dist1 = orc.knn(q, t, 1)
    assert idx1[:, 0].tolist() == [r[0] for r in case["knn2_idx"]]
    assert dist1[:, 0].tolist() == [r[0] for r in case["knn2_dist"]]
    assert orc.match(q, t) == [(a, b, 0, float(d)) for a, b, d in case["match"]]
    assert orc.match(q, t, cross_check=True) == [(a, b, 0, float(d))
                                                 for a, b, d in case["crosscheck"]]


def padded_rows(byte_rows):
    """Rows given as a few bytes, zero-padded to the 32 bytes of an ORB descriptor."""
    out = np.zeros((len(byte_rows), 32), np.uint8)
    for i, r in enumerate(byte_rows):
        out[i, :len(r)] = r
    return out


def test_opencv_crosscheck_regression_vector():
    """The vector of OpenCV's own regression test for the cross-check
    (Features2d_DMatch.issue_11855): one consistent pair, not two."""
    reg = KAT["opencv_regression"]
    q, t = padded_rows(reg["sources"]), padded_rows(reg["targets"])
    assert orc.match(q, t, cross_check=True) == [(a, b, 0, float(d)) for a, b, d in reg["crosscheck"]]
    assert orc.match(q, t) == [(a, b, 0, float(d)) for a, b, d in reg["match_without_crosscheck"]]


def test_empty_inputs():
    q = rows([1, 2, 3])
    e = np.zeros((0, 32), np.uint8)
    assert orc.match(q, e) == []
    assert orc.match(e, q) == []
    assert orc.match(q, e, cross_check=True) == []
    idx, dist = orc.knn(q, e, 2)
    assert (idx == -1).all() and (dist == 2147483647).all()


@pytest.mark.parametrize("case", KAT["min_dist_filter"])
def test_min_dist_filter(case):
    m = [(i, i, 0, float(d)) for i, d in enumerate(case["distances"])]
    assert orc.min_dist_filter(m) == [m[i] for i in case["kept"]]


def test_ratio_filter():
    c = KAT["ratio_0p75"]
    got = orc.ratio_filter(np.array(c["knn_idx"], np.int32), np.array(c["knn_dist"], np.int32),
                           0.75)
    assert got == [(q, t, 0, d) for q, t, d in c["kept"]]


def test_sort_is_stable():
    m = [(0, 9, 0, 5.0), (1, 8, 0, 3.0), (2, 7, 0, 5.0), (3, 6, 0, 3.0), (4, 5, 0, 1.0)]
    assert orc.sort_by_distance(m) == sorted(m, key=lambda x: x[3])


@pytest.mark.parametrize("nq,nt,seed", [(37, 53, 1), (200, 300, 2), (64, 1, 3), (1, 500, 4)])
def test_against_numpy_bruteforce(nq, nt, seed):
    q = datagen.random_descriptors(nq, seed)
    t = datagen.random_descriptors(nt, seed + 100)
    for k in (1, 2):
        idx, dist = orc.knn(q, t, k)
        ridx, rdist = datagen.numpy_knn(q, t, k)
        assert np.array_equal(idx, ridx)
        assert np.array_equal(dist, rdist)


def test_tie_stress_against_numpy():
    q = datagen.codebook_descriptors(150, 5)
    t = datagen.codebook_descriptors(400, 6)
    idx, dist = orc.knn(q, t, 2)
    ridx, rdist = datagen.numpy_knn(q, t, 2)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)


@pytest.mark.parametrize("tie_stress", [False, True])
def test_crosscheck_against_python_restatement(tie_stress):
    """Independent statement of the cross-check: the pairs (q, t) in which t is q's nearest
    train row and q is t's nearest query (numpy brute force both ways, lowest index on ties)."""
    if tie_stress:
        q, t = datagen.codebook_descriptors(120, 7, words=8), datagen.codebook_descriptors(90, 8, words=8)
    else:
        q = datagen.random_descriptors(120, 7)
        t = datagen.perturbed_copy(q, 8)[:90]
    tq_idx, _ = datagen.numpy_knn(t, q, 1)
    qt_idx, qt_dist = datagen.numpy_knn(q, t, 1)
    want = [(qq, int(qt_idx[qq, 0]), 0, float(qt_dist[qq, 0])) for qq in range(q.shape[0])
            if int(tq_idx[int(qt_idx[qq, 0]), 0]) == qq]
    got = orc.match(q, t, cross_check=True)
    assert got == want and len(got) > 0
    # a subset of the plain matches: cross-check only removes pairs
    assert set(got) <= set(orc.match(q, t))


def test_multithreaded_equals_single():
    q = datagen.random_descriptors(300, 9)
    t = datagen.random_descriptors(700, 10)
    a = orc.knn(q, t, 2)
    b = orc.knn(q, t, 2, threads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
