"""The complex-scalar factorisation's control flow (spectra_amd/csrc/zfac_flow.hpp — the source libmispec.so compiles over HIP
kernels) built over a host backend behind the same C entry points (tests/cpp/zfac_host_capi.cpp, test infrastructure) and put
through the checks the GPU module applies to libmispec.so (tests/zfac_checks.py, tests/test_gpu_zfac.py).  No GPU needed; the
product itself has no host path."""
import ctypes as C
import os
import subprocess

import pytest

import zfac_checks as Z

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hostlib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("zfac") / "libzfac_host.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "spectra_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "zfac_host_capi.cpp"), "-o", so])
    lib = C.CDLL(so)
    lib.mispec_zdense_rows.restype = C.c_int64
    lib.mispec_zdense_cols.restype = C.c_int64
    i64, vp = C.c_int64, C.c_void_p
    dp, vpp = C.POINTER(C.c_double), C.POINTER(C.c_void_p)
    lib.mispec_zdense_upload.argtypes = [vp, i64, i64, dp, i64, C.c_int, C.c_char, vpp]
    lib.mispec_zdense_coeff.argtypes = [vp, i64, i64, dp]
    lib.mispec_zdense_gemv_host.argtypes = [vp, dp, dp]
    for nm in ("mispec_zdense_destroy", "mispec_zdense_rows", "mispec_zdense_cols", "mispec_zfac_destroy", "mispec_zfac_subspace_dim"):
        getattr(lib, nm).argtypes = [vp]
    lib.mispec_zfac_create_dense.argtypes = [vp, vp, C.c_int, C.c_int, vpp]
    lib.mispec_zfac_create_op.argtypes = [vp, Z.op_fn, vp, i64, C.c_int, C.c_int, vpp]
    lib.mispec_zfac_init.argtypes = [vp, dp, C.POINTER(i64)]
    lib.mispec_zfac_factorize.argtypes = [vp, C.c_int, C.c_int, C.POINTER(i64)]
    lib.mispec_zfac_f_norm.argtypes = [vp, dp]
    lib.mispec_zfac_get_H.argtypes = [vp, dp]
    lib.mispec_zfac_get_V.argtypes = [vp, C.c_int, dp]
    lib.mispec_zfac_get_f.argtypes = [vp, dp]
    return lib


@pytest.mark.parametrize("n,m", [(10, 6), (300, 40)])
@pytest.mark.parametrize("hermitian", [False, True])
def test_flow_dense_operator(hostlib, n, m, hermitian):
    Z.run_dense_case(hostlib, None, n, m, hermitian)


@pytest.mark.parametrize("hermitian", [False, True])
def test_flow_host_pointer_operator(hostlib, hermitian):
    Z.run_callback_case(hostlib, None, hermitian)


def test_operator_image(hostlib):
    Z.run_operator_checks(hostlib, None)
