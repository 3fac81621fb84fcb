"""The per-task dense eigen-solver of the ground-state kernel (csrc/qd_eig.h), compiled for the CPU, against
numpy.linalg.eigh (what the reference calls on the whole 32x32 matrix, ground_state.py:149-162): hop-type blocks
(non-negative diagonal, non-positive couplings, connected), every block size 2..32, couplings from 1e-22 to 1e44,
near-degenerate lowest pairs.  No GPU needed."""
import ctypes

import numpy as np
import pytest

import helpers as H


def solve(A):
    h = H.hosttest()
    s = A.shape[0]
    packed = np.ascontiguousarray(A[np.tril_indices(s)], dtype=np.float64)
    lam = ctypes.c_double(); res = ctypes.c_double(); it = ctypes.c_int()
    x = np.zeros(s)
    rc = h.qdh_eig_lowest(s, H._p(packed, ctypes.c_double), ctypes.byref(lam), H._p(x, ctypes.c_double),
                          ctypes.byref(res), ctypes.byref(it))
    assert rc == 0
    return lam.value, x, res.value, it.value


def hop_block(rng, s, tscale, extra_edges=0.3):
    """connected block: random spanning tree + a few more edges, couplings -t*sqrt(k), diagonal O(1) >= 0"""
    A = np.zeros((s, s))
    for i in range(1, s):
        j = rng.integers(0, i)
        A[i, j] = A[j, i] = -tscale * rng.uniform(0.3, 3.0) * np.sqrt(rng.integers(1, 7))
    for _ in range(int(extra_edges * s)):
        i, j = rng.integers(0, s, 2)
        if i != j:
            A[i, j] = A[j, i] = -tscale * rng.uniform(0.3, 3.0) * np.sqrt(rng.integers(1, 7))
    A[np.diag_indices(s)] = rng.uniform(0, 4.0, s)
    A[rng.integers(0, s), rng.integers(0, s)] += 0.0
    return A


@pytest.mark.parametrize("s", list(range(2, 33)))
def test_lowest_pair_matches_eigh_over_scales(s):
    rng = np.random.default_rng(100 + s)
    worst = 0.0
    for tscale in (0.0 if s == 2 else 1e-22, 1e-8, 1e-3, 1.0, 30.0, 1e6, 1e14, 1e30, 1e44):
        for rep in range(6):
            A = hop_block(rng, s, tscale)
            w, V = np.linalg.eigh(A)
            hn = np.abs(A).sum(axis=1).max()
            lam, x, res, it = solve(A)
            assert abs(lam - w[0]) <= 4e-15 * hn, (s, tscale, lam, w[0])
            assert abs(np.linalg.norm(x) - 1) < 1e-14
            assert res <= 4e-15 * hn, (s, tscale, res / hn)
            gap = (w[1] - w[0]) / hn
            if gap > 1e-9:
                # eigenvector error of any backward-stable solver ~ eps / gap
                err = min(np.abs(x - V[:, 0]).max(), np.abs(x + V[:, 0]).max())
                assert err <= 2e-7 + 1e-15 / gap, (s, tscale, err, gap)
                worst = max(worst, err)
            assert it <= 64
    print(f"s={s}: worst eigenvector difference vs eigh {worst:.1e}")


@pytest.mark.parametrize("s", [2, 3, 4, 6, 8, 9, 12, 22])
def test_classical_limit_and_tiny_couplings(s):
    """couplings of exactly zero or far below the diagonal's spread: the lowest diagonal entry wins"""
    rng = np.random.default_rng(s)
    A = np.diag(rng.uniform(0, 3, s)); A[3 % s, 3 % s] = 0.0
    lam, x, res, it = solve(A)
    k = int(np.argmin(np.diag(A)))
    assert lam == pytest.approx(A[k, k], abs=1e-15) and abs(abs(x[k]) - 1) < 1e-12


@pytest.mark.parametrize("s,sep", [(4, 1e-5), (4, 1e-7), (6, 1e-6), (8, 1e-8), (8, 3e-9), (12, 1e-6), (20, 1e-7)])
def test_near_degenerate_lowest_pair_at_huge_coupling(s, sep):
    """The corner round 2 got wrong (plain Lanczos, no re-orthogonalisation): two weakly linked identical halves at
    tc ~ 1e14..1e20 give a lowest pair whose relative gap is `sep`; the vector must be the symmetric/positive one."""
    rng = np.random.default_rng(int(-np.log10(sep)) * 100 + s)
    h = s // 2
    for tc in (1e14, 1e20):
        B = hop_block(rng, h, tc)
        A = np.zeros((s, s)); A[:h, :h] = B; A[h:2 * h, h:2 * h] = B
        if s > 2 * h:
            A[s - 1, s - 1] = 5.0 * tc; A[s - 1, 0] = A[0, s - 1] = -tc
        hn = np.abs(A).sum(axis=1).max()
        A[0, h] = A[h, 0] = -sep * hn                      # weak link between the halves
        A[np.diag_indices(s)] += rng.uniform(0, 1.0, s)    # O(1) free-energy differences
        w, V = np.linalg.eigh(A)
        gap = (w[1] - w[0]) / hn
        lam, x, res, it = solve(A)
        assert abs(lam - w[0]) <= 4e-15 * hn
        assert res <= 2e-14 * hn
        err = min(np.abs(x - V[:, 0]).max(), np.abs(x + V[:, 0]).max())
        assert err <= 1e-6 * max(1.0, 1e-9 / gap) + 2e-15 / gap, (s, sep, tc, err, gap)


def test_laguerre_iteration_counts_are_bounded():
    rng = np.random.default_rng(5)
    its = []
    for s in (3, 4, 5, 8):
        for tscale in (1e-3, 1.0, 1e9, 1e20):
            for _ in range(50):
                its.append(solve(hop_block(rng, s, tscale))[3])
    its = np.array(its)
    print("Laguerre iterations: mean %.2f max %d" % (its.mean(), its.max()))
    assert its.mean() < 8 and its.max() <= 40


def test_small_eigenvector_entries_regression():
    """A pixel of the 6-dot `mid` scene: well separated ground state whose vector has an entry of 3e-7.  Inverse iteration
    on the top-down LDL^T (round 2's scheme) returned it wrong by 5e-10 (eigen residual 6e-10); the twisted factorisation
    gives every entry to working accuracy."""
    A = np.zeros((5, 5))
    A[np.diag_indices(5)] = [1.8211606185966067e-03, 9.0892916727111694e-02, 1.1252147134350532e-01,
                             1.9574221943041425e-01, 2.2405314073563976e-01]
    for i, j, v in ((0, 1, -6.2326547064007944e-03), (0, 4, -1.1348431758338040e+00), (1, 3, -8.7785459752451171e-05),
                    (2, 4, -5.1873123805475409e-03)):
        A[i, j] = A[j, i] = v
    lam, x, res, it = solve(A)
    w, V = np.linalg.eigh(A)
    v0 = V[:, 0] * np.sign(V[0, 0]) * np.sign(x[0])
    assert res <= 2e-15 and abs(lam - w[0]) <= 1e-15
    assert np.all(np.abs(x - v0) <= 1e-13 * np.abs(v0) + 1e-20), (x, v0)


@pytest.mark.parametrize("s", [3, 4, 5, 6, 7, 8, 10, 16, 32])
def test_mixed_coupling_scales_inside_one_block(s):
    """couplings of one block spread over many decades (tc_i = tc_base exp(-alpha_i vb_i) differs per barrier): the residual
    stays at round-off and small vector entries keep their relative accuracy"""
    rng = np.random.default_rng(7 * s)
    for rep in range(40):
        A = np.zeros((s, s))
        for i in range(1, s):
            j = rng.integers(0, i)
            A[i, j] = A[j, i] = -10.0 ** rng.uniform(-9, 2)
        A[np.diag_indices(s)] = rng.uniform(0, 1.0, s)
        hn = np.abs(A).sum(axis=1).max()
        lam, x, res, it = solve(A)
        w, V = np.linalg.eigh(A)
        assert abs(lam - w[0]) <= 4e-15 * hn and res <= 4e-15 * hn, (s, rep, res / hn)


def test_column_tail_far_below_the_pivot_regression():
    """A 10-state block of the random-action sweep (seed 1234, env 8, tc up to 2e45 next to couplings of 7e-5): after
    scaling, a Householder column had x0 ~ 1e-66 and a tail of ~1e-160; v0^2 underflowed and tau = 0 * (1 / denormal) = NaN.
    Negligible tails are dropped now."""
    rows = [[0.0], [-6.974298567106001e-05, 0.020594523099134676], [0.0, 0.0, 0.17739829741913127], [0.0, -0.025654117816680404, 0.0, 0.20331981604977045], [0.0, -1.9997155568046502e+45, 0.0, 0.0, 0.20989064725290518], [-5.121032850084054, 0.0, -1.9997155568046502e+45, 0.0, 0.0, 0.3314526208159805], [0.0, -5.121032850084054, 0.0, 0.0, 0.0, -6.974298567106001e-05, 0.3591142439254327], [0.0, 0.0, 0.0, -1.9997155568046502e+45, -0.025654117816680404, 0.0, 0.0, 0.3912569026942947], [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, -6.974298567106001e-05, 0.44611644657561556], [0.0, 0.0, -2971.837515200498, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.4715202777515515]]
    s = len(rows)
    A = np.zeros((s, s))
    for i, r in enumerate(rows):
        A[i, :i + 1] = r; A[:i + 1, i] = r
    hn = np.abs(A).sum(axis=1).max()
    lam, x, res, it = solve(A)
    w = np.linalg.eigvalsh(A)
    assert np.isfinite(res) and np.all(np.isfinite(x))
    assert abs(lam - w[0]) <= 4e-15 * hn and res <= 4e-15 * hn


@pytest.mark.parametrize("s", [3, 5, 8, 9, 12, 20])
def test_extreme_scale_mix_never_gives_nan(s):
    rng = np.random.default_rng(900 + s)
    for rep in range(200):
        A = np.zeros((s, s))
        for i in range(1, s):
            j = rng.integers(0, i)
            A[i, j] = A[j, i] = -10.0 ** rng.uniform(-30, 45) * (rng.random() < 0.9)
        for _ in range(s // 2):
            i, j = rng.integers(0, s, 2)
            if i != j: A[i, j] = A[j, i] = -10.0 ** rng.uniform(-30, 45)
        A[np.diag_indices(s)] = rng.uniform(0, 1.0, s) * 10.0 ** rng.integers(0, 5)
        hn = np.abs(A).sum(axis=1).max()
        lam, x, res, it = solve(A)
        assert np.isfinite(lam) and np.isfinite(res) and np.all(np.isfinite(x)), (s, rep)
        assert res <= 1e-14 * hn, (s, rep, res / hn)
