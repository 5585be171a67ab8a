"""The Eigen-facing constructors and result types of include/Spectra (the MISPEC_HAVE_EIGEN blocks) compile and link against
tests/cpp/eigen_lite, a small stand-in for Eigen's API (Eigen is not in this image — SURVEY.md 8f row 2).  The host-only
LinAlg program is also RUN with the Eigen-like container types; the programs that need a GPU are run by the GPU tests."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def test_eigen_facing_code_compiles_and_links():
    import __graft_entry__ as g

    g.build_eigen_api_checks()
    for exe in ("eigen_api.bin", "dropin_symeigs_eigenapi.bin", "linalg_host_eigenapi.bin"):
        assert os.path.exists(os.path.join(CPP, exe))
    out = subprocess.run([os.path.join(CPP, "eigen_api.bin")], stdout=subprocess.PIPE, text=True, timeout=60)
    assert out.returncode == 0 and "compiled" in out.stdout  # the no-GPU branch: loading the shared library works


def test_host_linalg_with_eigen_like_containers():
    import __graft_entry__ as g

    g.build_eigen_api_checks()
    out = subprocess.run([os.path.join(CPP, "linalg_host_eigenapi.bin")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0 and "ALL PASSED" in out.stdout, out.stdout
