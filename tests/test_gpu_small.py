"""a11-a13 — the ncv x ncv restart kernels (one workgroup, LDS-resident) vs the oracle and vs the
identities the reference's own tests assert (test/QR.cpp:20-99, :117-134; test/Eigen.cpp:67-110)."""
import numpy as np
import pytest

import oracle as O
import spectra_amd as sa
from helpers import random_tridiag

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [2, 3, 17, 40, 64, 100, 128])
@pytest.mark.parametrize("shift", [1.2345, 0.6789])
def test_tridiag_qr(ctx, n, shift):
    T = random_tridiag(n, 123 + n)
    Q, QtHQ = sa.tridiag_qr(T, shift, ctx=ctx)
    I = np.eye(n)
    tol = 1e-12  # test/QR.cpp:22
    assert np.abs(Q.T @ Q - I).max() < tol and np.abs(Q @ Q.T - I).max() < tol
    R = Q.T @ (T - shift * I)
    assert np.abs(np.tril(R, -1)).max() < tol            # H - sI = QR with R upper triangular
    assert np.abs(QtHQ - Q.T @ T @ Q).max() < tol        # matrix_QtHQ
    assert np.abs(np.tril(QtHQ, -2)).max() == 0.0 and np.array_equal(QtHQ, QtHQ.T)
    # oracle agreement (same algorithm, device libm / FMA contraction differ by rounding only)
    _, D0, Q0 = O.tridiag_qr(T, shift)
    assert np.abs(Q - Q0).max() < 1e-13 * n and np.abs(QtHQ - D0).max() < 1e-13 * n


def test_tridiag_qr_deflation(ctx):
    T = np.diag([1.0, 2.0, 3.0, -1.0]) + np.diag([1e-20, 0.5, 0.25], -1) + np.diag([1e-20, 0.5, 0.25], 1)
    Q, QtHQ = sa.tridiag_qr(T, 0.3, ctx=ctx)
    _, D0, Q0 = O.tridiag_qr(T, 0.3)
    assert QtHQ[1, 0] == 0.0 == D0[1, 0]
    assert np.abs(Q - Q0).max() < 1e-14


@pytest.mark.parametrize("n", [1, 2, 3, 10, 40, 50, 64, 100, 128])
def test_tridiag_eigen(ctx, n):
    T = random_tridiag(n, 321 + n) if n > 1 else np.array([[0.7]])
    ev, U = sa.tridiag_eigen(T, ctx=ctx)
    assert np.abs(T @ U - U * ev).max() < 1e-12          # test/Eigen.cpp:84-86
    assert np.abs(U.T @ U - np.eye(n)).max() < 1e-12
    ev0, U0 = O.tridiag_eigen(T)
    assert np.abs(np.sort(ev) - np.sort(ev0)).max() < 1e-13
    assert np.abs(np.sort(ev) - np.linalg.eigvalsh(T)).max() < 1e-12


def test_tridiag_eigen_special_cases(ctx):
    ev, U = sa.tridiag_eigen(np.zeros((6, 6)), ctx=ctx)   # TridiagEigen.h:142-150
    assert np.all(ev == 0) and np.array_equal(U, np.eye(6))
    ev, U = sa.tridiag_eigen(np.diag([3.0, -1.0, 2.0]), ctx=ctx)
    assert np.array_equal(ev, [3.0, -1.0, 2.0]) and np.array_equal(U, np.eye(3))
    with pytest.raises(ValueError):
        sa.tridiag_eigen(np.zeros((129, 129)), ctx=ctx)


# ---- the sweeps of a restart as a skewed pipeline on the device (k_restart_pipelined) ------------------------------------------
@pytest.mark.parametrize("kind", ["dense", "graded", "zeros"])
@pytest.mark.parametrize("n,p", [(2, 1), (3, 2), (5, 4), (17, 9), (40, 18), (40, 26), (40, 39), (50, 20), (64, 63), (64, 1)])
def test_pipelined_restart_kernel_equals_the_host_routine_bit_for_bit(ctx, n, p, kind):
    # lane s of wave 0 = sweep s, waves 1-3 rotate Q: no fused multiply-add, glibc's hypot restated for the device
    # (SmallDensePipelined.h) — T and Q must be the host's bits, so that either side can run a restart's sweeps
    from test_host_small_pipelined import _case

    d, e, mu = _case(n, p, 1000 * n + p, kind)
    d0, e0, Q0, _ = sa.restart_sweeps(d, e, mu, "host-serial")
    d2, e2, Q2, _ = sa.restart_sweeps(d, e, mu, "device-pipelined", ctx=ctx)
    assert np.array_equal(d0, d2) and np.array_equal(e0, e2) and np.array_equal(Q0, Q2)


def test_pipelined_restart_kernel_many_random_shapes(ctx):
    from test_host_small_pipelined import _case

    rng = np.random.default_rng(11)
    for trial in range(120):
        n = int(rng.integers(2, 65))
        p = int(rng.integers(1, n))
        d, e, mu = _case(n, p, 5000 + trial, ["dense", "graded", "zeros"][trial % 3])
        a = sa.restart_sweeps(d, e, mu, "host-pipelined")
        b = sa.restart_sweeps(d, e, mu, "device-pipelined", ctx=ctx)
        assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3])), (trial, n, p)


def test_one_wavefront_restart_kernel_agrees_to_rounding(ctx):
    from test_host_small_pipelined import _case

    d, e, mu = _case(40, 18, 3, "dense")
    d0, e0, Q0, _ = sa.restart_sweeps(d, e, mu, "host-serial")
    d3, e3, Q3, _ = sa.restart_sweeps(d, e, mu, "device-wavefront", ctx=ctx)
    assert np.abs(d0 - d3).max() < 1e-12 and np.abs(e0 - e3).max() < 1e-12 and np.abs(Q0 - Q3).max() < 1e-12
