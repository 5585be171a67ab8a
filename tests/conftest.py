import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build the library, the oracle and the C++ test
    programs once, exactly as the driver's build step does.  Never a fallback: if hipcc is missing the build fails loudly."""
    lib = os.path.join(ROOT, "spectra_amd", "libmispec.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g

        g.build()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "operator_only: a GPU test of an operator without a solver (modules that run every test in both "
                            "orthogonalisation modes run these once)")


@pytest.fixture(scope="session")
def ctx():
    """The default device context (session-wide).  Fails loudly without a GPU — no fallback."""
    import spectra_amd as sa

    return sa.default_context()
