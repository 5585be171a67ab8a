"""The product's host-side ncv x ncv kernels of the general restart (include/Spectra/internal/SmallDenseGen.h,
exported through the C ABI; no GPU needed) vs the oracle and vs the reference's identities."""
import numpy as np
import pytest

import oracle as O
import spectra_amd as sa


def hessenberg(n, seed):
    return np.triu(np.random.default_rng(seed).uniform(-1, 1, (n, n)), -1)


@pytest.mark.parametrize("n", [3, 10, 40, 64, 100])
def test_host_kernels_match_oracle(n):
    H = hessenberg(n, 100 + n)
    Q, D = sa.hess_qr(H, 0.6789)
    Q0, D0 = O.hess_qr(H, 0.6789)
    assert np.abs(Q - Q0).max() < 1e-14 and np.abs(D - D0).max() < 1e-14
    assert np.abs(D - Q.T @ H @ Q).max() < 1e-12          # test/QR.cpp:59-61
    H2 = H.copy()
    if n > 3:
        H2[1, 0] = 0.0
    Q, D = sa.double_shift_qr(H2, 2.0, 3.0)
    Q0, D0 = O.double_shift_qr(H2, 2.0, 3.0)
    assert np.abs(Q - Q0).max() < 1e-14 and np.abs(D - D0).max() < 1e-14
    assert np.abs(D - Q.T @ H2 @ Q).max() < 1e-12         # test/QR.cpp:156-158
    T, U = sa.hess_schur(H)
    T0, U0 = O.hess_schur(H)
    assert np.abs(T - T0).max() < 1e-14 and np.abs(U - U0).max() < 1e-14
    assert np.abs(H @ U - U @ T).max() < 1e-12            # test/Schur.cpp
    ev, V = sa.hess_eigen(H)
    ev0, V0 = O.hess_eigen(H)
    assert np.abs(ev - ev0).max() < 1e-14 and np.abs(V - V0).max() < 1e-13
    assert np.abs(H @ V - V * ev).max() < 1e-12           # test/Eigen.cpp:42


def test_host_kernel_argument_checks():
    with pytest.raises(ValueError):
        sa.hess_qr(np.zeros((1, 1)), 0.0)
    ev, V = sa.hess_eigen(np.array([[2.0]]))
    assert ev[0] == 2.0 and V[0, 0] == 1.0


@pytest.mark.parametrize("n", [3, 4, 5, 10, 31, 40, 64, 96, 128])
def test_lane_parallel_source_on_one_host_lane(n):
    # internal/SmallDenseGenLanes.h is what the HIP kernel k_hess_restart compiles; with one lane on the host it must give
    # what the host classes (internal/SmallDenseGen.h) and the oracle give
    H = hessenberg(n, 300 + n)
    Q, D = sa.hess_qr_lanes_host(H, 0.6789)
    Q0, D0 = O.hess_qr(H, 0.6789)
    assert np.abs(Q - Q0).max() < 1e-14 and np.abs(D - D0).max() < 1e-14
    Q1, D1 = sa.hess_qr(H, 0.6789)
    assert np.array_equal(Q, Q1) and np.array_equal(D, D1)     # same operations in the same order
    for variant in range(3):
        H2 = H.copy()
        if variant == 1 and n > 4:
            H2[2, 1] = 0.0                                      # two unreduced blocks
        if variant == 2 and n > 6:
            H2[1, 0] = 0.0
            H2[4, 3] = 1e-300                                   # deflated by the absolute test
        Q, D = sa.double_shift_qr_lanes_host(H2, 2.0, 3.0)
        Q0, D0 = O.double_shift_qr(H2, 2.0, 3.0)
        assert np.abs(Q - Q0).max() < 1e-14 and np.abs(D - D0).max() < 1e-14
        Q1, D1 = sa.double_shift_qr(H2, 2.0, 3.0)
        assert np.array_equal(Q, Q1) and np.array_equal(D, D1)
        assert np.abs(Q.T @ Q - np.eye(n)).max() < 1e-13 and np.abs(D - Q.T @ H2 @ Q).max() < 1e-12
