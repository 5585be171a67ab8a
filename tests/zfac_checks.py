"""Checks of the complex-scalar factorisation behind the mispec_zfac / mispec_zdense entry points (include/mispec_extras.h), written
against a ctypes library object so that they can run twice: on libmispec.so on the GPU (tests/test_gpu_zfac.py) and, without a GPU,
on a test-only build of the same control flow (spectra_amd/csrc/zfac_flow.hpp) over a host backend (tests/cpp/zfac_host_capi.cpp,
tests/test_host_zfac.py).  What they assert is what the reference's test/Arnoldi.cpp:20-85 asserts: A V - V H = f e', V^H V = I,
the residual norm."""
import ctypes as C

import numpy as np

MISPEC_EINVAL = -1
op_fn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))


def ok(rc):
    assert rc == 0, rc


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def matrix(n, hermitian, seed):
    rng = np.random.default_rng(seed)
    A = rng.uniform(-1, 1, (n, n)) + 1j * rng.uniform(-1, 1, (n, n))
    if hermitian:
        A = A + A.conj().T
    return np.asfortranarray(A)


def check_identities(A, V, H, f, k, beta, tol):
    Vk, Hk = V[:, :k], H[:k, :k]
    R = A @ Vk - Vk @ Hk
    R[:, -1] -= f
    scale = max(1.0, np.abs(A).sum(axis=1).max())
    assert np.abs(R).max() <= tol * scale
    assert np.abs(Vk.conj().T @ Vk - np.eye(k)).max() <= tol
    assert abs(np.linalg.norm(f) - beta) <= tol * scale


def run_factorisation(lib, ctxh, A, m, hermitian, stored, callback):
    """stored: the matrix handed to the library (for a Hermitian operator only its lower triangle is valid)."""
    n = A.shape[0]
    fac = C.c_void_p()
    keep = []
    D = C.c_void_p()
    if callback:
        def op(user, x, y):
            xv = np.ctypeslib.as_array(x, shape=(2 * n,)).view(np.complex128)
            yv = np.ctypeslib.as_array(y, shape=(2 * n,)).view(np.complex128)
            yv[:] = A @ xv
            return 0
        cb = op_fn(op)
        keep.append(cb)
        ok(lib.mispec_zfac_create_op(ctxh, cb, None, n, m, int(hermitian), C.byref(fac)))
    else:
        uplo = b"L" if hermitian else b"\0"
        ok(lib.mispec_zdense_upload(ctxh, n, n, dp(stored), n, 0, uplo, C.byref(D)))
        assert lib.mispec_zdense_rows(D) == n and lib.mispec_zdense_cols(D) == n
        ok(lib.mispec_zfac_create_dense(ctxh, D, m, int(hermitian), C.byref(fac)))
    try:
        rng = np.random.default_rng(7)
        v0 = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
        ops = C.c_int64(0)
        ok(lib.mispec_zfac_init(fac, dp(v0), C.byref(ops)))
        assert ops.value == 2 and lib.mispec_zfac_subspace_dim(fac) == 1
        H = np.zeros((m, m), dtype=np.complex128, order="F")
        V = np.zeros((n, m), dtype=np.complex128, order="F")
        f = np.zeros(n, dtype=np.complex128)
        beta = C.c_double()

        def fetch():
            ok(lib.mispec_zfac_get_H(fac, dp(H)))
            ok(lib.mispec_zfac_get_V(fac, m, dp(V)))
            ok(lib.mispec_zfac_get_f(fac, dp(f)))
            ok(lib.mispec_zfac_f_norm(fac, C.byref(beta)))

        tol = 1e-12 if n <= 100 else 1e-11
        fetch()
        check_identities(A, V, H, f, 1, beta.value, tol)
        for frm, to in ((1, m // 2), (m // 2, m)):
            ok(lib.mispec_zfac_factorize(fac, frm, to, C.byref(ops)))
            assert lib.mispec_zfac_subspace_dim(fac) == to
            fetch()
            check_identities(A, V, H, f, to, beta.value, tol)
        assert ops.value == 2 + (m - 1)
        if hermitian:
            assert np.abs(np.triu(H, 2)).max() == 0.0 and np.abs(np.tril(H, -2)).max() == 0.0
            # the Ritz values of the projected matrix lie inside A's spectrum
            ritz = np.linalg.eigvalsh((H + H.conj().T) / 2)
            ev = np.linalg.eigvalsh(A)
            assert ev[0] - 1e-10 <= ritz[0] and ritz[-1] <= ev[-1] + 1e-10
        assert lib.mispec_zfac_factorize(fac, m + 1, m + 2, C.byref(ops)) == MISPEC_EINVAL
    finally:
        ok(lib.mispec_zfac_destroy(fac))
        if D:
            ok(lib.mispec_zdense_destroy(D))


def run_dense_case(lib, ctxh, n, m, hermitian):
    A = matrix(n, hermitian, 100 + n)
    stored = A.copy(order="F")
    if hermitian:  # only the lower triangle is read: poison the rest and the imaginary part of the diagonal
        stored[np.triu_indices(n, 1)] = 1e3 - 7e2j
        stored[np.diag_indices(n)] += 5j
    run_factorisation(lib, ctxh, A, m, hermitian, stored, callback=False)


def run_callback_case(lib, ctxh, hermitian):
    A = matrix(64, hermitian, 5)
    run_factorisation(lib, ctxh, A, 12, hermitian, A, callback=True)


def run_operator_checks(lib, ctxh):
    n = 37
    A = matrix(n, False, 3)
    D = C.c_void_p()
    ok(lib.mispec_zdense_upload(ctxh, n, n, dp(A), n, 0, b"\0", C.byref(D)))
    x = np.random.default_rng(1).uniform(-1, 1, n) + 1j * np.random.default_rng(2).uniform(-1, 1, n)
    try:
        y = np.zeros(n, dtype=np.complex128)
        ok(lib.mispec_zdense_gemv_host(D, dp(x), dp(y)))
        assert np.abs(y - A @ x).max() <= 1e-13 * n
        out = np.zeros(2)
        ok(lib.mispec_zdense_coeff(D, 5, 9, dp(out)))
        assert complex(out[0], out[1]) == A[5, 9]
        assert lib.mispec_zdense_coeff(D, n, 0, dp(out)) == MISPEC_EINVAL
    finally:
        ok(lib.mispec_zdense_destroy(D))
    # row-major input and an upper triangle
    Hm = matrix(n, True, 4)
    stored = np.ascontiguousarray(Hm)  # row-major
    stored[np.tril_indices(n, -1)] = -9.0
    ok(lib.mispec_zdense_upload(ctxh, n, n, dp(stored), n, 1, b"U", C.byref(D)))
    try:
        y = np.zeros(n, dtype=np.complex128)
        ok(lib.mispec_zdense_gemv_host(D, dp(x), dp(y)))
        assert np.abs(y - Hm @ x).max() <= 1e-13 * n
    finally:
        ok(lib.mispec_zdense_destroy(D))
