"""a4-a10 — the device-resident Lanczos/Arnoldi factorisation (K2-K10) vs the oracle and vs the identities
of test/Arnoldi.cpp:19-85 (A V - V H = f e', V'V = I)."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle as O
import spectra_amd as sa
from helpers import sparse_fixture

pytestmark = pytest.mark.gpu


class DenseOp:
    """A user operator with the reference's concept: rows(), cols(), perform_op on host arrays."""

    def __init__(self, A):
        self.A = A
        self.calls = 0

    def rows(self):
        return self.A.shape[0]

    def cols(self):
        return self.A.shape[1]

    def perform_op(self, x):
        self.calls += 1
        return self.A @ x


def check_identities(fac, A, k, tol=1e-12):
    V, H, f = fac.matrix_V()[:, :k], fac.matrix_H()[:k, :k], fac.vector_f()
    resid = A @ V - V @ H
    if k > 1:
        assert np.abs(resid[:, :k - 1]).max() < tol
    assert np.abs(resid[:, -1] - f).max() < tol
    assert np.abs(V.T @ V - np.eye(k)).max() < tol
    assert abs(np.linalg.norm(f) - fac.f_norm()) < tol


@pytest.mark.parametrize("symmetric", [True, False])
def test_arnoldi_cpp_identities_user_op(ctx, symmetric):
    # test/Arnoldi.cpp: n = 10, m = 6, dense operator given as a user OpType (host perform_op)
    n, m = 10, 6
    rng = np.random.default_rng(123)
    M = rng.uniform(-1, 1, (n, n))
    A = M + M.T if symmetric else M
    op = DenseOp(A)
    fac = sa.Factorization(op, m, symmetric, ctx=ctx)
    v0 = rng.uniform(-1, 1, n)
    fac.init(v0)
    assert fac.subspace_dim() == 1
    check_identities(fac, A, 1)
    fac.factorize_from(1, m // 2)
    assert fac.subspace_dim() == m // 2
    check_identities(fac, A, m // 2)
    fac.factorize_from(m // 2, m)
    assert fac.subspace_dim() == m
    check_identities(fac, A, m)
    assert fac.num_operations() == 2 + (m - 1) == op.calls
    # the oracle run on the same data produces the same H and V (to rounding)
    ofac = O.Factorization(O.Op.dense_sym(A) if symmetric else O.Op.dense_gen(A), m, symmetric)
    ofac.init(v0)
    ofac.factorize_from(1, m)
    V0, H0, f0 = ofac.matrices()
    assert np.abs(fac.matrix_H() - H0).max() < 1e-12
    assert np.abs(fac.matrix_V() - V0).max() < 1e-11
    with pytest.raises(ValueError):
        sa.Factorization(op, m, symmetric, ctx=ctx).init(np.zeros(n))       # Arnoldi.h:146-148
    f2 = sa.Factorization(op, m, symmetric, ctx=ctx)
    f2.init(v0)
    with pytest.raises(ValueError, match="larger than the current subspace dimension"):
        f2.factorize_from(3, 5)                                             # Lanczos.h:70-75


@pytest.mark.parametrize("n,prob,m", [(100, 0.1, 20), (1000, 0.01, 50), (1000, 0.01, 64), (1000, 0.01, 65), (1000, 0.01, 100),
                                      (1000, 0.01, 128), (1000, 0.01, 200), (1000, 0.01, 256)])  # > 64 columns: column panels
def test_lanczos_on_device_matrix_vs_oracle(ctx, n, prob, m):
    A, S = sparse_fixture(n, prob)
    op = sa.SparseSymMatProd(A, ctx=ctx)
    fac = sa.Factorization(op, m, True)
    fac.init_random(0)  # SimpleRandom(0) generated on the device == the reference's default start vector
    fac.factorize_from(1, m)
    check_identities(fac, S.toarray(), m, tol=1e-11)
    ofac = O.Factorization(O.Op.csr(n, n, S.indptr, S.indices, S.data), m, True)
    ofac.init(O.simple_random(n, 0))
    ofac.factorize_from(1, m)
    V0, H0, f0 = ofac.matrices()
    H = fac.matrix_H()
    assert np.abs(np.tril(np.triu(H, -1), 1) - H).max() == 0.0  # tridiagonal, zero elsewhere (Lanczos.h:85-86)
    # early columns agree to rounding; later ones drift by the usual Lanczos error growth, so compare the
    # projected matrices through their eigenvalues and the leading block entry-wise
    assert np.abs(H[:8, :8] - H0[:8, :8]).max() < 1e-10
    assert np.abs(np.linalg.eigvalsh(H) - np.linalg.eigvalsh(H0)).max() < 1e-9
    assert fac.num_operations() == ofac.num_operations() == m + 1


def test_start_vector_generated_on_device_is_the_reference_stream(ctx):
    n = 4099
    A = sp.identity(n, format="csr") * 2.0
    fac = sa.Factorization(sa.SparseGenMatProd(A, ctx=ctx), 4, True)
    fac.init_random(0)
    v = fac.matrix_V(1)[:, 0]
    r = O.simple_random(n, 0)          # v = A v0 / |A v0| = v0 / |v0|
    assert np.abs(v - r / np.linalg.norm(r)).max() < 1e-15


def test_expand_basis_on_zero_matrix(ctx):
    # test/Example4.cpp case 1: A = 0 -> every step takes the restart path (Arnoldi.h:66-115)
    n, m = 100, 6
    A = sp.csr_matrix((n, n))
    fac = sa.Factorization(sa.SparseGenMatProd(A, ctx=ctx), m, True)
    v0 = np.random.default_rng(1).uniform(-1, 1, n)
    fac.init(v0)
    assert fac.f_norm() == 0.0
    fac.factorize_from(1, m)
    V, H = fac.matrix_V(), fac.matrix_H()
    assert np.abs(V.T @ V - np.eye(m)).max() < 1e-12 and np.abs(H).max() == 0.0
    ofac = O.Factorization(O.Op.csr(n, n, A.indptr, A.indices, A.data), m, True)
    ofac.init(v0)
    ofac.factorize_from(1, m)
    assert fac.num_operations() == ofac.num_operations()


def test_restart_primitives_vs_oracle(ctx):
    # one implicit restart: device QR sweeps + compress_V == the oracle's TridiagQR/apply_YQ/compress_V
    n, m, k = 1000, 20, 12
    A, S = sparse_fixture(n, 0.01)
    fac = sa.Factorization(sa.SparseSymMatProd(A, ctx=ctx), m, True)
    fac.init_random(0)
    fac.factorize_from(1, m)
    ev, U = fac.tridiag_eigen()
    H = fac.matrix_H()
    assert np.abs(H @ U - U * ev).max() < 1e-12
    order = np.argsort(-np.abs(ev))
    shifts = ev[order][k:]
    V_before, f_before, beta_before = fac.matrix_V(), fac.vector_f(), fac.f_norm()
    fac.restart_sym(shifts)
    assert fac.subspace_dim() == k
    Sd = S.toarray()
    check_identities(fac, Sd, k, tol=1e-10)
    # explicit Q from the oracle's sweeps
    Hq, Q = H.copy(), np.eye(m)
    for mu in shifts:
        _, Hq, Qi = O.tridiag_qr(Hq, mu)
        Q = Q @ Qi
    # Only the leading k x k block, H(k,k-1) and the first k columns of Q are well determined: with exact
    # shifts the trailing block is the deflated part, whose entries are rounding noise amplified by the
    # sweeps (the reason the reference keeps deflate_H disabled, HermEigsBase.h:144-146).
    Hd = fac.matrix_H()
    assert np.abs(Hd[:k, :k] - Hq[:k, :k]).max() < 1e-11
    assert abs(Hd[k, k - 1] - Hq[k, k - 1]) < 1e-11
    Vn = V_before @ Q[:, :k + 1]
    assert np.abs(fac.matrix_V()[:, :k] - Vn[:, :k]).max() < 1e-10
    fk = f_before * Q[m - 1, k - 1] + Vn[:, k] * Hq[k, k - 1]
    assert np.abs(fac.vector_f() - fk).max() < 1e-10
    # host-side variant (caller supplies Q and the compressed H)
    fac2 = sa.Factorization(sa.SparseSymMatProd(A, ctx=ctx), m, True)
    fac2.init_random(0)
    fac2.factorize_from(1, m)
    fac2.compress_V(Q, Hq, k)
    assert np.abs(fac2.matrix_V()[:, :k] - fac.matrix_V()[:, :k]).max() < 1e-10
    # continue: back to an m-step factorisation
    fac.factorize_from(k, m)
    check_identities(fac, Sd, m, tol=1e-10)
    Y = np.random.default_rng(0).uniform(-1, 1, (m, 5))
    assert np.abs(fac.ritz_vectors(Y) - fac.matrix_V() @ Y).max() < 1e-12


def test_wide_basis_restart_and_limits(ctx):
    # ncv > 64: orthogonalisation in column panels, out-of-place V*Q, LDS-resident restart kernel
    n, m, k = 3000, 100, 45
    A, S = sparse_fixture(1000, 0.01)
    S3 = sp.block_diag([S, S * 0.5 + sp.identity(1000), S * 0.25 - sp.identity(1000)], format="csr")
    fac = sa.Factorization(sa.SparseGenMatProd(S3, ctx=ctx), m, True)
    fac.init_random(0)
    fac.factorize_from(1, m)
    check_identities(fac, S3.toarray(), m, tol=1e-11)
    ev, U = fac.tridiag_eigen()
    order = np.argsort(-np.abs(ev))
    fac.restart_sym(ev[order][k:])
    assert fac.subspace_dim() == k
    V, H, f = fac.matrix_V(k), fac.matrix_H()[:k, :k], fac.vector_f()
    Sd = S3.toarray()
    resid = Sd @ V - V @ H
    resid[:, k - 1] -= f
    assert np.abs(resid).max() < 1e-10 and np.abs(V.T @ V - np.eye(k)).max() < 1e-11   # A V = V H + f e_k'
    fac.factorize_from(k, m)
    check_identities(fac, Sd, m, tol=1e-10)
    with pytest.raises(ValueError, match="1024"):
        sa.Factorization(sa.SparseGenMatProd(S3, ctx=ctx), 1025, True)
    # m = 200 > 128: the restart sweeps run on the host (the LDS-resident kernel stops at 128 columns)
    m2, k2 = 200, 90
    fac2 = sa.Factorization(sa.SparseGenMatProd(S3, ctx=ctx), m2, True)
    fac2.init_random(0)
    fac2.factorize_from(1, m2)
    check_identities(fac2, Sd, m2, tol=1e-10)
    ev2, _ = fac2.tridiag_eigen()
    order2 = np.argsort(-np.abs(ev2))
    fac2.restart_sym(ev2[order2][k2:])
    assert fac2.subspace_dim() == k2
    V2, H2, f2 = fac2.matrix_V(k2), fac2.matrix_H()[:k2, :k2], fac2.vector_f()
    r2 = Sd @ V2 - V2 @ H2
    r2[:, k2 - 1] -= f2
    assert np.abs(r2).max() < 1e-10 and np.abs(V2.T @ V2 - np.eye(k2)).max() < 1e-11
    fac2.factorize_from(k2, m2)
    check_identities(fac2, Sd, m2, tol=1e-10)


def test_vq_on_matrix_cores_matches_fma_kernel():
    # V <- V Q has two implementations: plain f64 FMAs out of LDS (default: already HBM-bound) and
    # v_mfma_f64_16x16x4_f64 (MISPEC_VQ=mfma).  Same products, different summation order: agree to rounding.
    import os
    import subprocess
    import sys

    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import spectra_amd as sa; from helpers import sparse_fixture\n"
        "A, S = sparse_fixture(1000, 0.01)\n"
        "fac = sa.Factorization(sa.SparseSymMatProd(A), 50, True); fac.init_random(0); fac.factorize_from(1, 50)\n"
        "ev, U = fac.tridiag_eigen(); order = np.argsort(-np.abs(ev)); fac.restart_sym(ev[order][23:])\n"
        "sys.stdout.write(np.concatenate([fac.matrix_V(24).ravel(), fac.vector_f()]).tobytes().hex())\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    res = []
    for impl in ("fma", "mfma"):
        env = dict(os.environ, MISPEC_VQ=impl)
        r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        res.append(np.frombuffer(bytes.fromhex(r.stdout.strip()), dtype=np.float64))
    assert np.abs(res[0] - res[1]).max() <= 1e-13
