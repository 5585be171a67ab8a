"""The host-side LinAlg classes of the C++ header API, exercised like the reference's unit tests
(test/QR.cpp, test/Eigen.cpp, test/Schur.cpp) by tests/cpp/linalg_host.cpp — plain C++11, no GPU, no library."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_linalg_classes():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g

    exe = g.build_host_linalg_test()
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert out.returncode == 0 and "ALL PASSED" in out.stdout, out.stdout


def test_headers_are_cxx11():
    # the reference requires C++11 only; the drop-in program must at least parse in that mode
    src = os.path.join(ROOT, "tests", "cpp", "dropin_symeigs.cpp")
    out = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"), src],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0 and "warning" not in out.stdout, out.stdout


def test_complex_factorisation_flow_on_a_host_backend(tmp_path):
    # spectra_amd/csrc/zfac_flow.hpp (the control flow the library runs over HIP kernels, csrc/zfac.hip) instantiated with a plain
    # host backend: the checks of the reference's test/Arnoldi.cpp for complex general / Hermitian matrices, and the breakdown paths
    exe = str(tmp_path / "zfac_flow_host")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "spectra_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "zfac_flow_host.cpp"), "-o", exe])
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert out.returncode == 0 and "ALL PASSED" in out.stdout, out.stdout
