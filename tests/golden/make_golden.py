"""Generates tests/golden/symeigs_golden.npz — known answers for the reference's own fixtures.

The reference (C++/Eigen) cannot be run in this image, so the golden values are
  * the full spectra of the reference's reproducible sparse fixtures (gen_sparse_data, test/SymEigs.cpp:25-42,
    lower triangle mirrored) from numpy.linalg.eigvalsh — an independent dense solver, which is also what the
    reference's regression tests compare against (test/Example1.cpp:39-41);
  * a CRC of the fixtures' triplets (so a libstdc++ change that alters the fixture is noticed);
  * the oracle's own counters on those fixtures at generation time (regression pin of the restatement);
  * spot values of the synthetic-matrix hash (bit-exact integers -> doubles).
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as O  # noqa: E402
from helpers import RULES_SYM, SPARSE_CASES, sparse_fixture  # noqa: E402

out = {}
for n, prob, k, m in SPARSE_CASES:
    A, S = sparse_fixture(n, prob)
    r, c, v = O.gen_sparse_data(n, prob)
    out[f"crc_{n}"] = np.array([zlib.crc32(r.tobytes()), zlib.crc32(c.tobytes()), zlib.crc32(v.tobytes())], dtype=np.int64)
    out[f"spectrum_{n}"] = np.linalg.eigvalsh(S.toarray())
    op = O.Op.csc_sym(n, A.indptr, A.indices, A.data, True)
    for rule in RULES_SYM:
        s = O.SymEigsSolver(op, k, m)
        s.init()
        nconv = s.compute(getattr(O, rule))
        out[f"oracle_{n}_{rule}"] = np.array([nconv, s.info(), s.num_iterations(), s.num_operations()], dtype=np.int64)
        out[f"oracle_evals_{n}_{rule}"] = s.eigenvalues()
pairs = [(0, 0), (0, 1), (5, 7), (123456, 123457), (9999999, 9999999), (17, 100017)]
out["synth_pairs"] = np.array(pairs, dtype=np.int64)
out["synth_values"] = np.array([O.lib().oracle_synth_value(O.SYNTH_SEED, a, b) for a, b in pairs])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "symeigs_golden.npz"), **out)
print("wrote symeigs_golden.npz with", len(out), "arrays")
