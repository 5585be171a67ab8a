"""Times the kernels of one banded shift solve under rocprofv3 (run this script under the profiler): C5's matrix, 30 solves."""
import os, sys, numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spectra_amd as sa
from test_gpu_fullsize import banded_spd
n = 2_000_000
A = banded_spd(n, 3, seed=5)
op = sa.SparseSymShiftSolve(sp.tril(A).tocsc())
op.set_shift(0.0)
x = np.random.default_rng(0).uniform(-1, 1, n)
for _ in range(30):
    y = op.perform_op(x)
print(float(np.abs(A @ y - x).max()))
