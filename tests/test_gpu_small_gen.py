"""a19 on the device: the Hessenberg sweeps of the general restart as HIP kernels (spectra_amd/csrc/small.hip
k_hess_restart: one wavefront, H and Q in LDS; source internal/SmallDenseGenLanes.h) vs the oracle's restatement of
UpperHessenbergQR (LinAlg/UpperHessenbergQR.h:136-255) and DoubleShiftQR (LinAlg/DoubleShiftQR.h:334-467), with the
reference's own test identities (test/QR.cpp:38-98, :136-160), and a GenEigsSolver run with the restart on the device."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle as O
import spectra_amd as sa

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def hessenberg(n, seed):
    return np.triu(np.random.default_rng(seed).uniform(-1, 1, (n, n)), -1)


@pytest.mark.parametrize("n", [3, 4, 5, 10, 30, 31, 40, 63, 64, 65, 80, 96])
def test_device_sweeps_match_the_oracle(ctx, n):
    H = hessenberg(n, 500 + n)
    Q, D = sa.hess_qr_device(H, 0.6789, ctx)
    Q0, D0 = O.hess_qr(H, 0.6789)
    assert np.abs(Q - Q0).max() <= 1e-13 * n and np.abs(D - D0).max() <= 1e-13 * n
    assert np.abs(Q.T @ Q - np.eye(n)).max() <= 1e-12            # test/QR.cpp:52-55
    assert np.abs(D - Q.T @ H @ Q).max() <= 1e-12                # :59-61
    assert np.abs(np.tril(D, -2)).max() == 0.0                   # stays Hessenberg
    for variant in range(3):
        H2 = H.copy()
        if variant == 1 and n > 4:
            H2[2, 1] = 0.0
        if variant == 2 and n > 6:
            H2[1, 0] = 0.0
            H2[4, 3] = 1e-300
        Q, D = sa.double_shift_qr_device(H2, 2.0, 3.0, ctx)
        Q0, D0 = O.double_shift_qr(H2, 2.0, 3.0)
        assert np.abs(Q - Q0).max() <= 1e-13 * n and np.abs(D - D0).max() <= 1e-13 * n, (n, variant)
        assert np.abs(Q.T @ Q - np.eye(n)).max() <= 1e-12        # test/QR.cpp:148-151
        assert np.abs(D - Q.T @ H2 @ Q).max() <= 1e-12           # :156-158
    with pytest.raises(ValueError):
        sa.hess_qr_device(hessenberg(97, 1), 0.1, ctx)


def test_restart_on_the_device_gives_the_host_solve():
    # MISPEC_SMALL=device|host A/B: the same arithmetic (one source), so the same eigenvalues and operation counts
    code = (
        "import sys, time, numpy as np; sys.path.insert(0, %r); import spectra_amd as sa\n"
        "op = sa.SparseGenMatProd.synth_band(300000)\n"
        "for rep in range(2):\n"
        "    e = sa.GenEigsSolver(op, 10, 30); e.profile(1); t0 = time.perf_counter(); e.init(); n = e.compute(sa.SortRule.LargestMagn, 1000, 1e-11); dt = time.perf_counter() - t0\n"
        "p = e.get_profile()\n"
        "print(n, e.num_operations(), e.num_iterations(), dt, p['ms_small'], p['n_small'], ' '.join(repr(complex(x)) for x in e.eigenvalues()))\n"
    ) % ROOT
    outs = []
    for mode in ("host", "device"):
        env = dict(os.environ, MISPEC_SMALL=mode)
        r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout.split())
    assert outs[0][:3] == outs[1][:3] and int(outs[0][0]) == 10
    a, b = (np.array([complex(x) for x in o[6:]]) for o in outs)
    assert np.abs(a - b).max() <= 1e-10
    assert int(outs[1][5]) > 0 and int(outs[0][5]) == 0           # the kernel really ran in "device" mode, not in "host"
    print("general restart A/B: host %.4f s per solve, device %.4f s (kernel %.3f ms in %d launches)" %
          (float(outs[0][3]), float(outs[1][3]), float(outs[1][4]), int(outs[1][5])))
