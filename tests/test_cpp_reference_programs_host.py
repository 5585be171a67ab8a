"""SURVEY.md 8f row 2, the part that needs no GPU: the reference's test programs for host-side classes — test/Schur.cpp
(UpperHessenbergSchur), test/Orthogonalization.cpp, test/Givens.cpp, test/QR.cpp (UpperHessenbergQR, TridiagQR, DoubleShiftQR)
and test/Eigen.cpp (UpperHessenbergEigen, TridiagEigen), real and complex cases — compiled UNMODIFIED against include/Spectra with
a stand-in in Eigen's place (tests/cpp/eigen_lite; oracle/eigen_shim for the three that instantiate complex scalars) (tests/cpp/build_reference_tests.sh, run by __graft_entry__.build() where /root/reference is present) and run
here.  The device-side programs are in tests/test_gpu_reference_programs.py."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["Schur", "Orthogonalization", "Givens", "QR", "Eigen"])
def test_reference_host_program(name):
    exe = os.path.join(ROOT, "tests", "cpp", "_ref", name + ".bin")
    if not os.path.exists(exe):
        pytest.skip("tests/cpp/_ref/%s.bin not built (needs /root/reference at build time)" % name)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "All tests passed" in r.stdout, r.stdout[-3000:]
