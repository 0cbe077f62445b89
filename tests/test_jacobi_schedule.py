"""Host-side model of the pair schedule the solve kernels use for the one-sided Jacobi SVD
(poseestimation_amd/csrc/ransac_epnp.hip: jacobi_sweeps_12_shared, jacobi_svd with a lane
group): the pairs of a sweep run in steps T = i + j - 1, the pairs of a step in rounds of G
lanes, and with the overlap the next sweep starts 12 steps behind the current one.

What makes the kernel's results equal to the cyclic order's (OpenCV JacobiSVDImpl_,
modules/core/src/lapack.cpp: for i < n-1, for j > i) bit for bit is that a pair reads and
writes only its own two rows and their norms: if every row sees its pairs in the same
sequence in both orders, every pair gets the same operands.  The tests check exactly that on
the schedule, and then on numbers with a float64 model of the pair arithmetic."""
import math

import numpy as np
import pytest


def cyclic(n):
    return [(i, j) for i in range(n - 1) for j in range(i + 1, n)]


def rounds(n, G, sweeps, overlap):
    """Rounds as the kernel runs them: lists of (sweep, i, j), one entry per lane."""
    last = 2 * n - 4
    period = n if overlap else last + 1
    out = []
    for s in range(sweeps + 1):
        for T0 in range(period):
            T1 = T0 + period
            lo0, lo1 = max(T0 - (n - 2), 0), T1 - (n - 2)
            c0 = (T0 >> 1) - lo0 + 1 if s < sweeps else 0
            c1 = (T1 >> 1) - lo1 + 1 if (overlap and s > 0 and T1 <= last) else 0
            for q0 in range(0, c0 + c1, G):
                rnd = []
                for g in range(G):
                    q = q0 + g
                    if q < c0 + c1:
                        old = q >= c0
                        i = lo1 + q - c0 if old else lo0 + q
                        j = (T1 if old else T0) + 1 - i
                        rnd.append((s - 1 if old else s, i, j))
                out.append(rnd)
    return out


def row_sequences(order, n):
    seq = {r: [] for r in range(n)}
    for s, i, j in order:
        seq[i].append((s, i, j))
        seq[j].append((s, i, j))
    return seq


@pytest.mark.parametrize("n", [4, 5, 12])
@pytest.mark.parametrize("G", [1, 2, 4, 6, 8, 64])
@pytest.mark.parametrize("overlap", [False, True])
def test_schedule_keeps_every_rows_pair_sequence(n, G, overlap):
    sweeps = 3
    rs = rounds(n, G, sweeps, overlap)
    flat = [p for rnd in rs for p in rnd]
    want = [(s, i, j) for s in range(sweeps) for (i, j) in cyclic(n)]
    assert sorted(flat) == sorted(want)  # every pair of every sweep exactly once
    for rnd in rs:  # the pairs of a round touch different rows
        rows = [r for (_, i, j) in rnd for r in (i, j)]
        assert len(rows) == len(set(rows))
        assert all(0 <= i < j < n for (_, i, j) in rnd)
    assert row_sequences(flat, n) == row_sequences(want, n)


def test_round_counts():
    per_sweep = lambda n, G, overlap: len([r for r in rounds(n, G, 1, overlap) if r])
    assert per_sweep(12, 1, False) == 66
    assert per_sweep(12, 2, False) == 39
    assert per_sweep(12, 4, False) == 26
    assert per_sweep(12, 6, False) == 21
    assert per_sweep(5, 2, False) == 7 and per_sweep(4, 2, False) == 5
    # steady state with the overlap: 12 steps per sweep once six lanes take a whole step
    many = len([r for r in rounds(12, 8, 10, True) if r])
    assert many == 12 * 10 + 9


def cv_hypot(a, b):
    """hypot as OpenCV's JacobiSVD computes it (lapack.cpp's own template), rounded one
    operation at a time."""
    a, b = abs(a), abs(b)
    if a > b:
        b /= a
        return a * math.sqrt(1 + b * b)
    if b > 0:
        a /= b
        return b * math.sqrt(1 + a * a)
    return 0.0


def pair(A, W, i, j, eps):
    """One pair on rows i, j of A (squared norms W); the arithmetic of the fixed-size pair
    in ransac_epnp.hip in float64, products and sums rounded one by one."""
    a, b, p = W[i], W[j], 0.0
    for k in range(A.shape[1]):
        p += A[i, k] * A[j, k]
    if abs(p) <= eps * math.sqrt(a * b):
        return False
    p *= 2.0
    beta = a - b
    gamma = cv_hypot(p, beta)
    if beta < 0:
        s = math.sqrt((gamma - beta) * 0.5 / gamma)
        c = p / (gamma * s * 2)
    else:
        c = math.sqrt((gamma + beta) / (gamma * 2))
        s = p / (gamma * c * 2)
    na = nb = 0.0
    for k in range(A.shape[1]):
        t0 = c * A[i, k] + s * A[j, k]
        t1 = -s * A[i, k] + c * A[j, k]
        A[i, k], A[j, k] = t0, t1
        na += t0 * t0
        nb += t1 * t1
    W[i], W[j] = na, nb
    return True


def sweeps_cyclic(A, max_sweeps=30):
    eps = np.finfo(np.float64).eps * 10
    W = np.array([float(np.dot(r, r)) for r in A])
    for it in range(max_sweeps):
        changed = False
        for i, j in cyclic(A.shape[0]):
            changed |= pair(A, W, i, j, eps)
        if not changed:
            return it + 1
    return max_sweeps


def sweeps_scheduled(A, G, overlap, max_sweeps=30):
    """The kernel's loop: a sweep without a rotation ends the run; with the overlap the
    first steps of its successor have been run by then."""
    n = A.shape[0]
    eps = np.finfo(np.float64).eps * 10
    W = np.array([float(np.dot(r, r)) for r in A])
    rotated = {}
    done_at = 2 * n - 4
    period = n if overlap else done_at + 1
    rs = iter(rounds(n, G, max_sweeps, overlap))
    for s in range(max_sweeps + 1):
        for T0 in range(period):
            T1 = T0 + period
            lo0, lo1 = max(T0 - (n - 2), 0), T1 - (n - 2)
            c0 = (T0 >> 1) - lo0 + 1 if s < max_sweeps else 0
            c1 = (T1 >> 1) - lo1 + 1 if (overlap and s > 0 and T1 <= done_at) else 0
            for _ in range(0, c0 + c1, G):
                for (sw, i, j) in next(rs):
                    if pair(A, W, i, j, eps):
                        rotated[sw] = True
            if overlap and s > 0 and T1 == done_at and (s == max_sweeps or not rotated.get(s - 1)):
                return s
        if not overlap and (s == max_sweeps - 1 or not rotated.get(s)):
            return s + 1
    raise AssertionError("unreachable")


@pytest.mark.parametrize("shape", [(12, 12), (5, 6), (4, 6)])
@pytest.mark.parametrize("G,overlap", [(2, False), (4, False), (8, True), (64, True), (2, True)])
def test_scheduled_sweeps_equal_cyclic_sweeps_bitwise(shape, G, overlap):
    rng = np.random.default_rng(shape[0] * 100 + G)
    for trial in range(4):
        M = rng.standard_normal(shape)
        if shape == (12, 12):
            if trial == 1:
                M = M @ M.T  # symmetric, like MtM
            if trial == 2:
                M[:, 5:] = 0  # rank deficient: zero norms, pairs that never rotate
        A0, A1 = M.copy(), M.copy()
        n0 = sweeps_cyclic(A0)
        n1 = sweeps_scheduled(A1, G, overlap)
        assert n0 == n1
        assert A0.tobytes() == A1.tobytes()


def table_rounds(name, G):
    """The precomputed rounds for 2 and 4 lanes per matrix, read from the kernel source."""
    import os
    import re
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       "poseestimation_amd", "csrc", "ransac_epnp.hip")
    text = open(src).read()
    body = re.search(name + r"\[(\d+)\] = \{(.*?)\};", text, re.S)
    words = [int(w.rstrip("u"), 16) for w in re.findall(r"0x[0-9A-Fa-f]+u", body.group(2))]
    assert len(words) == int(body.group(1))
    out = []
    for w in words:
        rnd = []
        for g in range(4):
            e = (w >> (8 * g)) & 0xFF
            if e != 0xFF:
                assert g < G  # lanes beyond the group never get a pair
                rnd.append((0, e >> 4, e & 15))
        out.append(rnd)
    return out


@pytest.mark.parametrize("name,G,n_rounds", [("JACOBI_ROUNDS_2", 2, 35), ("JACOBI_ROUNDS_4", 4, 23)])
def test_precomputed_rounds_keep_every_rows_pair_sequence(name, G, n_rounds):
    rs = table_rounds(name, G)
    assert len(rs) == n_rounds
    flat = [p for rnd in rs for p in rnd]
    want = [(0, i, j) for (i, j) in cyclic(12)]
    assert sorted(flat) == sorted(want)
    for rnd in rs:
        rows = [r for (_, i, j) in rnd for r in (i, j)]
        assert len(rows) == len(set(rows)) and all(0 <= i < j < 12 for (_, i, j) in rnd)
    assert row_sequences(flat, 12) == row_sequences(want, 12)


@pytest.mark.parametrize("name,G", [("JACOBI_ROUNDS_2", 2), ("JACOBI_ROUNDS_4", 4)])
def test_precomputed_rounds_equal_cyclic_sweeps_bitwise(name, G):
    rs = table_rounds(name, G)
    eps = np.finfo(np.float64).eps * 10
    rng = np.random.default_rng(G)
    for trial in range(4):
        M = rng.standard_normal((12, 12))
        if trial % 2:
            M = M @ M.T
        A0, A1 = M.copy(), M.copy()
        n0 = sweeps_cyclic(A0)
        W = np.array([float(np.dot(r, r)) for r in A1])
        n1 = 0
        for n1 in range(1, 31):
            rotated = False
            for rnd in rs:
                for (_, i, j) in rnd:
                    rotated |= pair(A1, W, i, j, eps)
            if not rotated:
                break
        assert n0 == n1 and A0.tobytes() == A1.tobytes()


def pair_by_two_lanes(A, W, i, j, eps):
    """jacobi_pair_lds_half in float64: both lanes read the rows as they stand, test and
    compute the rotation; lane `second` then forms cv * A[i] + cj * A[j] with (cv, cj) =
    (c, s) for row i and (-s, c) for row j, and its row's norm."""
    a, b, p = W[i], W[j], 0.0
    ri, rj = A[i].copy(), A[j].copy()
    for k in range(A.shape[1]):
        p += ri[k] * rj[k]
    if abs(p) <= eps * math.sqrt(a * b):
        return False
    p *= 2.0
    beta = a - b
    gamma = cv_hypot(p, beta)
    if beta < 0:
        s = math.sqrt((gamma - beta) * 0.5 / gamma)
        c = p / (gamma * s * 2)
    else:
        c = math.sqrt((gamma + beta) / (gamma * 2))
        s = p / (gamma * c * 2)
    for second in (False, True):
        cv, cj = (-s, c) if second else (c, s)
        norm2 = 0.0
        row = j if second else i
        for k in range(A.shape[1]):
            t = cv * ri[k] + cj * rj[k]
            A[row, k] = t
            norm2 += t * t
        W[row] = norm2
    return True


def test_pair_split_over_two_lanes_equals_the_pair_bitwise():
    eps = np.finfo(np.float64).eps * 10
    rng = np.random.default_rng(12)
    rotated = 0
    for trial in range(300):
        M = rng.standard_normal((12, 12)) * 10.0 ** rng.integers(-3, 4)
        if trial % 3 == 0:
            M = M @ M.T
        A0, A1 = M.copy(), M.copy()
        W0 = np.array([float(np.dot(r, r)) for r in A0])
        W1 = W0.copy()
        for i, j in cyclic(12):
            r0 = pair(A0, W0, i, j, eps)
            r1 = pair_by_two_lanes(A1, W1, i, j, eps)
            assert r0 == r1
            rotated += r0
        assert A0.tobytes() == A1.tobytes() and W0.tobytes() == W1.tobytes()
    assert rotated > 10000
