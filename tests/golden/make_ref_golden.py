"""Generates tests/golden/ref_pin_golden.npz from oracle/_ref — the REFERENCE'S OWN solver code (yixuan/spectra headers compiled
where they lie by oracle/build_ref.sh, oracle/eigen_shim standing in for Eigen) — run in the container that has /root/reference.

What is recorded, per case: [nconv, info, num_iterations, num_operations] and the eigenvalues the reference's code returned, for
  * SymEigsSolver + SparseSymMatProd on gen_sparse_data(n, prob) (test/SymEigs.cpp:25-42, :133-167) x the 5 selection rules;
  * GenEigsSolver + SparseGenMatProd on the same fixtures (test/GenEigs.cpp:148-174) x the 6 selection rules;
  * test/Example1.cpp (cycle Laplacian, 3 (k, m) pairs, tol 1e-15), test/Example2.cpp (3 literal matrices),
    the diag(1..10) example of SymEigsSolver.h:99-126.
tests/test_ref_pin.py compares the restatement (oracle/spectra_oracle*.hpp) with these vectors on every run, with or without
oracle/_ref present, and with oracle/_ref itself when it is.  The same script also rewrites the `oracle_*` entries of
symeigs_golden.npz from the reference's code (they used to be a regression pin of the restatement only).
Run from the repo root:  python tests/golden/make_ref_golden.py
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as O  # noqa: E402
from oracle import ref as R  # noqa: E402
from helpers import EXAMPLE2, RULES_SYM, SPARSE_CASES, cycle_laplacian, sparse_fixture  # noqa: E402

RULES_GEN = ["LargestMagn", "LargestReal", "LargestImag", "SmallestMagn", "SmallestReal", "SmallestImag"]

assert R.available(), "oracle/_ref is not built (needs /root/reference)"
out = {"describe": np.frombuffer(R.lib().ref_describe(), dtype=np.uint8)}


def rec(key, r):
    out[key] = np.array([r.nconv, r.info, r.num_iterations, r.num_operations], dtype=np.int64)
    ev = np.asarray(r.eigenvalues)
    out[key + "_evals"] = ev.view(np.float64) if np.iscomplexobj(ev) else ev


for n, prob, k, m in SPARSE_CASES:
    A, S = sparse_fixture(n, prob)
    for rule in RULES_SYM:
        rec(f"sym_{n}_{rule}", R.symeigs(R.Op.csc_sym(n, A.indptr, A.indices, A.data, True), k, m, selection=getattr(R, rule)))
    for rule in RULES_GEN:
        rec(f"gen_{n}_{rule}", R.geneigs(R.Op.csc(n, A.indptr, A.indices, A.data), k, m, selection=getattr(R, rule)))
M = cycle_laplacian(20)
for k, m in [(3, 6), (5, 12), (6, 12)]:
    rec(f"example1_{k}_{m}", R.symeigs(R.Op.dense_sym(M), k, m, selection=R.LargestMagn, tol=1e-15, sorting=R.SmallestAlge))
for i, M2 in enumerate(EXAMPLE2):
    rec(f"example2_{i}", R.symeigs(R.Op.dense_sym(M2), 1, 3, selection=R.LargestAlge))
D = sp.diags(np.arange(1.0, 11.0)).tocsc()
rec("doc_diag", R.symeigs(R.Op.csc_sym(10, D.indptr, D.indices, D.data, True), 3, 6, selection=R.LargestAlge))
# shift-and-invert drivers (SymEigsShiftSolver.h, GenEigsRealShiftSolver.h) on the reference's shift fixtures
# (test/SymEigsShift.cpp:148-185, test/GenEigsRealShift.cpp:146-180); the shift solve is scipy's sparse LU behind a callback
import scipy.sparse.linalg as spla  # noqa: E402

for n, prob, k, m, sigma in [(10, 0.5, 3, 6, 1.0), (100, 0.1, 10, 20, 10.0), (1000, 0.01, 20, 50, 100.0)]:
    A, S = sparse_fixture(n, prob)
    lu = spla.splu((S - sigma * sp.identity(n)).tocsc())
    for rule in RULES_SYM:
        rec(f"symshift_{n}_{rule}", R.symeigs_shift(R.Op.callback(n, lu.solve), k, m, sigma, selection=getattr(R, rule)))
for n, prob, k, m, sigma in [(10, 0.5, 3, 6, 1.0), (100, 0.1, 10, 30, 10.0), (1000, 0.01, 20, 50, 100.0)]:
    A, S = sparse_fixture(n, prob)
    lu = spla.splu((A - sigma * sp.identity(n)).tocsc())
    for rule in ["LargestMagn", "LargestReal", "LargestImag", "SmallestReal"]:
        rec(f"genshift_{n}_{rule}", R.geneigs_real_shift(R.Op.callback(n, lu.solve), k, m, sigma, selection=getattr(R, rule)))
np.savez_compressed(os.path.join(HERE, "ref_pin_golden.npz"), **out)
print("wrote ref_pin_golden.npz with", len(out), "arrays")

# symeigs_golden.npz: the oracle_* counters / eigenvalues now come from the reference's code
path = os.path.join(HERE, "symeigs_golden.npz")
old = dict(np.load(path))
for n, prob, k, m in SPARSE_CASES:
    for rule in RULES_SYM:
        c = out[f"sym_{n}_{rule}"]
        old[f"oracle_{n}_{rule}"] = np.array([c[0], c[1], c[2], c[3]], dtype=np.int64)
        old[f"oracle_evals_{n}_{rule}"] = out[f"sym_{n}_{rule}_evals"]
old["oracle_entries_source"] = np.frombuffer(b"oracle/_ref (the reference's own headers), tests/golden/make_ref_golden.py", dtype=np.uint8)
np.savez_compressed(path, **old)
print("rewrote the oracle_* entries of symeigs_golden.npz from oracle/_ref")
