"""The complex-scalar factorisation on the device (csrc/zfac.hip behind mispec_zfac / mispec_zdense, include/mispec_extras.h) —
outside the hot path of SURVEY.md section 8; it serves the complex instantiations of the reference's LinAlg/Arnoldi.h and
LinAlg/Lanczos.h (test/Arnoldi.cpp:122-158).  Checked the way that program checks them (tests/zfac_checks.py): A V - V H = f e',
V^H V = I, the residual norm, for general and Hermitian matrices, a dense device operator and a host-pointer callback operator.
The same checks run without a GPU on a host build of the control flow (tests/test_host_zfac.py); the reference's own
test/Arnoldi.cpp runs in tests/test_gpu_reference_programs.py."""
import pytest

import spectra_amd as sa

import zfac_checks as Z

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,m", [(10, 6), (300, 40), (1500, 24)])
@pytest.mark.parametrize("hermitian", [False, True])
def test_complex_factorisation_dense_operator(ctx, n, m, hermitian):
    Z.run_dense_case(sa.lib(), ctx.h, n, m, hermitian)


@pytest.mark.parametrize("hermitian", [False, True])
def test_complex_factorisation_host_pointer_operator(ctx, hermitian):
    Z.run_callback_case(sa.lib(), ctx.h, hermitian)


def test_complex_dense_operator_product_and_coefficients(ctx):
    Z.run_operator_checks(sa.lib(), ctx.h)
