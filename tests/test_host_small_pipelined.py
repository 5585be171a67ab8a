"""The restart's shifted QR sweeps as a skewed pipeline (include/Spectra/internal/SmallDensePipelined.h, the routine a restart
runs on the host) against the reference's serial order (tridiag_shifted_qr per shift: UpperHessenbergQR.h:515-693 via
HermEigsBase.h:124-147) — bit for bit — and against the oracle.  No device needed (mispec_restart_sweeps, variants 0 and 1)."""
import numpy as np
import pytest

import oracle as O
import spectra_amd as sa


def _case(n, p, seed, kind):
    rng = np.random.default_rng(seed)
    d = 4.0 * (rng.random(n) - 0.5)
    if kind == "dense":
        e = 1.0 + (rng.random(n - 1) - 0.5)
    elif kind == "graded":  # tiny sub-diagonals: both deflation tests (:526-539, :684-692) fire
        e = (rng.random(n - 1) - 0.5) * 10.0 ** (-rng.integers(0, 18, n - 1).astype(float))
    else:  # exact zeros and a shift equal to a diagonal entry (the x == 0 / y == 0 branches of Givens.h:166-205)
        e = rng.random(n - 1) - 0.5
        e[rng.integers(0, n - 1, max(1, n // 5))] = 0.0
    mu = 3.0 * (rng.random(p) - 0.5)
    if kind == "zeros":
        mu[0] = d[0]
    return d, e, mu


@pytest.mark.parametrize("kind", ["dense", "graded", "zeros"])
@pytest.mark.parametrize("n,p", [(2, 1), (3, 1), (3, 2), (5, 4), (17, 9), (40, 18), (40, 26), (40, 39), (64, 63), (65, 30), (100, 50), (128, 127)])
def test_pipelined_sweeps_equal_the_serial_order_bit_for_bit(n, p, kind):
    d, e, mu = _case(n, p, 1000 * n + p, kind)
    d0, e0, Q0, _ = sa.restart_sweeps(d, e, mu, "host-serial")
    d1, e1, Q1, _ = sa.restart_sweeps(d, e, mu, "host-pipelined")
    assert np.array_equal(d0, d1) and np.array_equal(e0, e1) and np.array_equal(Q0, Q1)
    assert np.abs(Q1.T @ Q1 - np.eye(n)).max() < 1e-12  # test/QR.cpp:22


def test_pipelined_sweeps_many_random_shapes():
    rng = np.random.default_rng(7)
    for trial in range(300):
        n = int(rng.integers(2, 97))
        p = int(rng.integers(1, n))
        d, e, mu = _case(n, p, trial, ["dense", "graded", "zeros"][trial % 3])
        a = sa.restart_sweeps(d, e, mu, "host-serial")
        b = sa.restart_sweeps(d, e, mu, "host-pipelined")
        assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3])), (trial, n, p)


@pytest.mark.parametrize("n,p", [(6, 3), (40, 18)])
def test_pipelined_sweeps_against_the_oracle(n, p):
    # the oracle applies one shift at a time (oracle/spectra_oracle.hpp, restating UpperHessenbergQR.h): chain it
    d, e, mu = _case(n, p, 99 + n, "dense")
    T = np.diag(d) + np.diag(e, -1) + np.diag(e, 1)
    Qacc = np.eye(n)
    for s in mu:
        _, T, Q = O.tridiag_qr(T, float(s))
        Qacc = Qacc @ Q
    d1, e1, Q1, _ = sa.restart_sweeps(d, e, mu, "host-pipelined")
    assert np.abs(np.diag(T) - d1).max() < 1e-13 * n and np.abs(np.diag(T, -1) - e1).max() < 1e-13 * n
    assert np.abs(Qacc - Q1).max() < 1e-13 * n


def test_pipelined_sweeps_are_faster_than_the_serial_order():
    # the point of the routine: it sits on the critical path of every restart (m = 40, 18 shifts: ~20 against ~50 us)
    d, e, mu = _case(40, 18, 5, "dense")
    t_serial = min(sa.restart_sweeps(d, e, mu, "host-serial", reps=200)[3] for _ in range(3))
    t_pipe = min(sa.restart_sweeps(d, e, mu, "host-pipelined", reps=200)[3] for _ in range(3))
    assert t_pipe < 0.8 * t_serial, (t_pipe, t_serial)
