"""The complex-scalar factorisation on the device (csrc/zfac.hip behind mispec_zfac / mispec_zdense, include/mispec_extras.h) —
outside the hot path of SURVEY.md section 8; it serves the complex instantiations of the reference's LinAlg/Arnoldi.h and
LinAlg/Lanczos.h (test/Arnoldi.cpp:122-158).  Checked the way that program checks them (tests/zfac_checks.py): A V - V H = f e',
V^H V = I, the residual norm, for general and Hermitian matrices, a dense device operator and a host-pointer callback operator.
The same checks run without a GPU on a host build of the control flow (tests/test_host_zfac.py)."""
import os
import subprocess

import pytest

import spectra_amd as sa

import zfac_checks as Z

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,m", [(10, 6), (300, 40), (1500, 24)])
@pytest.mark.parametrize("hermitian", [False, True])
def test_complex_factorisation_dense_operator(ctx, n, m, hermitian):
    Z.run_dense_case(sa.lib(), ctx.h, n, m, hermitian)


@pytest.mark.parametrize("hermitian", [False, True])
def test_complex_factorisation_host_pointer_operator(ctx, hermitian):
    Z.run_callback_case(sa.lib(), ctx.h, hermitian)


def test_complex_dense_operator_product_and_coefficients(ctx):
    Z.run_operator_checks(sa.lib(), ctx.h)


def test_reference_arnoldi_program():
    # /root/reference/test/Arnoldi.cpp, unmodified, against include/Spectra (oracle/eigen_shim in Eigen's place): Arnoldi / Lanczos
    # over DenseGenMatProd / DenseSymMatProd<double> (the real device factorisation) and over DenseGenMatProd / DenseHermMatProd
    # <std::complex<double>> (this module's)
    exe = os.path.join(ROOT, "tests", "cpp", "_ref", "Arnoldi.bin")
    if not os.path.exists(exe):
        pytest.skip("tests/cpp/_ref/Arnoldi.bin not built (needs /root/reference at build time)")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "All tests passed" in r.stdout, r.stdout[-3000:]
