"""spectra_amd — MI355X-native implicitly-restarted Lanczos eigensolver (drop-in for the Spectra hot path).

Python mirror of the reference's operator / solver interface for the path this repository implements
(yixuan/spectra v1.2.0, include/Spectra/): `SparseSymMatProd`, `SparseGenMatProd`, `SymEigsSolver`,
`SortRule`, `CompInfo` — same names, argument meaning and error behaviour — on top of the C ABI of
`libmispec.so` (include/mispec.h, HIP kernels for gfx950).  The C++ users' entry point is the header-only
API in include/Spectra/; this module exists for tests, the benchmark and Python callers.

There is NO CPU fallback: constructing a `Context` without a visible HIP device raises.
"""
import ctypes as C
import enum
import os

import numpy as np

from . import _capi
from ._capi import MispecError, Profile, build_library, check, lib

__all__ = ["SortRule", "CompInfo", "Context", "SparseSymMatProd", "SparseGenMatProd", "SparseSymShiftSolve", "SymEigsSolver",
           "SymEigsShiftSolver", "GenEigsSolver", "SVDMatOp", "PartialSVDSolver", "SparseRegularInverse", "SparseCholesky", "SymGEigsSolver", "SymShiftInvert", "SymGEigsShiftSolver", "SparseGenRealShiftSolve", "GenEigsRealShiftSolver", "shard_block",
           "Factorization", "tridiag_qr", "tridiag_eigen", "hess_qr", "double_shift_qr", "hess_schur", "hess_eigen", "MispecError", "build_library", "shard_range", "BAND_OFFSETS", "SYNTH_SEED"]

BAND_OFFSETS = (1, 2, 3, 1000, 1001, 100000, 100001)  # SURVEY.md §8(d) "M-band": 15 nnz/row with the diagonal
SYNTH_SEED = 20240607


# include/mispec.h MISPEC_ORTH_*: the reference's control flow, or the opt-in one-sweep steps; "onesweep-eager" applies the
# last correction of every sweep at once instead of letting it ride on the restart's V*Q pass, "onesweep-recorrect" is the test
# hook that lets one more correction follow every such fused restart
# "onesweep-restart-check" is the test hook of the restart without a host turn: its device-side test always reports a failure
# "onesweep-onered" / "onesweep-twored": one reduction per lagged step (include/mispec.h MISPEC_ORTH_ONE_REDUCTION) or the separate
# alpha reduction, whatever the library default is
ORTH_MODES = {"reference": 0, "onesweep": 1, "onesweep-eager": 1 | 0x100, "onesweep-recorrect": 1 | 0x200, "onesweep-restart-check": 1 | 0x400,
              "onesweep-onered": 1 | 0x800, "onesweep-twored": 1 | 0x1000, "onesweep-onered-eager": 1 | 0x800 | 0x100}


def _orth_mode_value(mode):
    """Name or integer of an orthogonalisation mode -> the integer of include/mispec.h; anything else is a ValueError that
    lists the names."""
    if isinstance(mode, str):
        if mode not in ORTH_MODES:
            raise ValueError("unknown orthogonalisation mode %r: one of %s" % (mode, ", ".join(sorted(ORTH_MODES))))
        return ORTH_MODES[mode]
    if isinstance(mode, int) and mode in ORTH_MODES.values():
        return mode
    raise ValueError("unknown orthogonalisation mode %r: one of %s" % (mode, ", ".join(sorted(ORTH_MODES))))



class SortRule(enum.IntEnum):
    """Util/SelectionRule.h:33-58 (same order as the C++ enum)."""
    LargestMagn = 0
    LargestReal = 1
    LargestImag = 2
    LargestAlge = 3
    SmallestMagn = 4
    SmallestReal = 5
    SmallestImag = 6
    SmallestAlge = 7
    BothEnds = 8


class CompInfo(enum.IntEnum):
    """Util/CompInfo.h:17-32."""
    Successful = 0
    NotComputed = 1
    NotConverging = 2
    NumericalIssue = 3


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def tiles_spmv_host(mat, x):
    """y = A x through the HOST image of the column-blocked tile format, in the kernel's summation order (mispec_tiles_spmv_host;
    no device needed).  mat: scipy CSR with sorted rows.  Returns (y or None when the format does not apply, stats dict)."""
    rp, ci, v = _i32(mat.indptr), _i32(mat.indices), _f64(mat.data)
    x = _f64(x)
    y = np.zeros(mat.shape[0])
    built = C.c_int(0)
    st = (C.c_int64 * 3)()
    check(lib().mispec_tiles_spmv_host(mat.shape[0], mat.shape[1], _ip(rp), _ip(ci), _dp(v), _dp(x), _dp(y), C.byref(built), st))
    return (y if built.value else None), {"entries": st[0], "padding": st[1], "chunks": st[2]}


def windows_host(mat, row_begin=0):
    """The x-window table of a CSR matrix (scipy, sorted rows) built on the HOST by the code the device builder runs
    (mispec_csr_windows_host; no device needed): (blocks x 32) int32, see include/mispec.h mispec_csr_windows_table."""
    rp, ci = _i32(mat.indptr), _i32(mat.indices)
    out = np.zeros(((mat.shape[0] + 255) // 256, 32), dtype=np.int32)
    check(lib().mispec_csr_windows_host(mat.shape[0], mat.shape[1], int(row_begin), _ip(rp), _ip(ci), _ip(out)))
    return out


def staged_spmv_host(mat, x):
    """y = A x through the HOST image of the staged format, in the order of its two kernels (mispec_staged_spmv_host; no device
    needed).  mat: scipy CSR with sorted rows.  Returns (y or None when the format does not apply, stats dict)."""
    rp, ci, v = _i32(mat.indptr), _i32(mat.indices), _f64(mat.data)
    x = _f64(x)
    y = np.zeros(mat.shape[0])
    built = C.c_int(0)
    st = (C.c_int64 * 5)()
    check(lib().mispec_staged_spmv_host(mat.shape[0], mat.shape[1], _ip(rp), _ip(ci), _dp(v), _dp(x), _dp(y), C.byref(built), st))
    return (y if built.value else None), {"bins": st[0], "slots": st[1], "batches": st[2], "chunks": st[3], "max_rounds": st[4],
                                          "well_filled": built.value == 1}


def last_ingest_info():
    """Seconds the host stages of the last matrix ingest on this thread took (mispec_last_ingest_info)."""
    out = np.zeros(10)
    check(lib().mispec_last_ingest_info(_dp(out), 10))
    keys = ("total", "mirror_triangle", "validate", "index_formats_and_h2d", "far_statistics_and_reordering", "tiles_build", "tiles_upload",
            "unused", "staged_build", "staged_upload")
    d = dict(zip(keys, out.tolist()))
    d.pop("unused")
    return d


def mirror_triangle_host(mat, uplo="L"):
    """The full symmetric CSR matrix that mispec_csr_from_triangle derives from the `uplo` triangle of a compressed scipy matrix
    (CSC or CSR; the other triangle is ignored) — host only, no device needed (mispec_mirror_triangle_host)."""
    import scipy.sparse as sp

    n, nc, outer, inner, val, row_major = _compressed(mat)
    cap = 2 * len(val) + 1
    rp = np.zeros(n + 1, dtype=np.int32)
    ci = np.zeros(cap, dtype=np.int32)
    v = np.zeros(cap)
    nnz = C.c_int64(0)
    check(lib().mispec_mirror_triangle_host(n, _ip(outer), _ip(inner), _dp(val), uplo.encode()[0:1], int(row_major), _ip(rp), _ip(ci), _dp(v),
                                           cap, C.byref(nnz)))
    return sp.csr_matrix((v[:nnz.value], ci[:nnz.value], rp), shape=(n, n))


def rcm_order(rowptr, colind, symmetric_pattern=True):
    """Reverse Cuthill-McKee ordering of an n x n CSR pattern on the host (mispec_rcm_order; no device needed).
    Returns (perm with perm[new] = old, gave_up, widest_level)."""
    rp, ci = _i32(rowptr), _i32(colind)
    n = len(rp) - 1
    perm = np.empty(n, dtype=np.int32)
    gave_up, widest = C.c_int(0), C.c_int64(0)
    check(lib().mispec_rcm_order(n, _ip(rp), _ip(ci), int(symmetric_pattern), _ip(perm), C.byref(gave_up), C.byref(widest)))
    return perm, bool(gave_up.value), int(widest.value)


def shard_block(n, world):
    """Rows per rank of the equal-block row partition (mispec_shard_block)."""
    return int(lib().mispec_shard_block(n, world))


def shard_range(n, world, rank):
    """Rows [begin, end) owned by `rank` (equal even-sized blocks, see mispec_shard_range)."""
    b, e = C.c_int64(), C.c_int64()
    check(lib().mispec_shard_range(n, world, rank, C.byref(b), C.byref(e)))
    return b.value, e.value


class Context:
    """One device + one HIP stream (+ the communicator of a row-sharded run)."""

    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        check(lib().mispec_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
        self.h = h
        self.device = int(device)
        self.rank, self.world = 0, 1
        self._keep = []

    def sync(self):
        check(lib().mispec_ctx_sync(self.h))

    @property
    def stream(self):
        return lib().mispec_ctx_stream(self.h)

    def set_comm_rccl(self, rank, world, unique_id):
        check(lib().mispec_ctx_set_comm_rccl(self.h, rank, world, unique_id))
        self.rank, self.world = rank, world

    def set_comm_callbacks(self, rank, world, allgather, allreduce_sum, exchange=None):
        """allgather(send_ptr, recv_ptr, count_per_rank, stream) / allreduce_sum(buf_ptr, count, stream) -> 0 on success;
        optional exchange(send_ptr, send_off[], send_count[], recv_ptr, recv_off[], recv_count[], stream) (mispec_comm.exchange)."""
        ag = _capi.allgather_fn(lambda user, s, r, cnt, st: int(allgather(s, r, cnt, st) or 0))
        ar = _capi.allreduce_fn(lambda user, b, cnt, st: int(allreduce_sum(b, cnt, st) or 0))
        comm = _capi.Comm(rank, world, ag, ar, None)
        if exchange is not None:
            def ex_tramp(user, s, so, sc, r, ro, rc, st):
                lst = lambda a: [int(a[i]) for i in range(world)]
                return int(exchange(int(s or 0), lst(so), lst(sc), int(r or 0), lst(ro), lst(rc), st) or 0)
            ex = _capi.exchange_fn(ex_tramp)
            comm.exchange = ex
            self._keep.append(ex)
        self._keep += [ag, ar, comm]
        check(lib().mispec_ctx_set_comm(self.h, C.byref(comm)))
        self.rank, self.world = rank, world

    @staticmethod
    def rccl_unique_id():
        buf = C.create_string_buffer(128)
        check(lib().mispec_rccl_unique_id(buf))
        return buf.raw

    def __del__(self):
        try:
            lib().mispec_ctx_destroy(self.h)
        except Exception:
            pass


_default_ctx = None


def set_option(name, value=None):
    """mispec_set_option: a tuning switch / test hook by name (include/mispec.h lists them); value None restores the default."""
    check(lib().mispec_set_option(name.encode(), None if value is None else str(value).encode()))


def get_option(name):
    v = lib().mispec_get_option(name.encode())
    return None if v is None else v.decode()


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class _DeviceMatrix:
    """A CSR row shard in HBM (mispec_csr)."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.h = handle

    def rows(self):
        return lib().mispec_csr_rows(self.h)

    def cols(self):
        return lib().mispec_csr_cols(self.h)

    def local_rows(self):
        return lib().mispec_csr_local_rows(self.h)

    def nnz(self):
        return lib().mispec_csr_local_nnz(self.h)

    def perform_op(self, x_in, y_out=None):
        """y_out = A * x_in with HOST arrays (the reference's perform_op contract)."""
        x = _f64(x_in)
        if x.shape != (self.cols(),):
            raise ValueError("perform_op: x_in must have cols() entries")
        y = np.empty(self.rows()) if y_out is None else y_out
        check(lib().mispec_spmv_host(self.h, _dp(x), _dp(y)))
        return y

    def __matmul__(self, X):
        """operator*: A @ X for a dense block (SparseSymMatProd.h:93-96)."""
        X = np.asarray(X, dtype=np.float64)
        if X.ndim == 1:
            return self.perform_op(X)
        Xf = np.asfortranarray(X)
        Y = np.empty((self.rows(), X.shape[1]), order="F")
        check(lib().mispec_spmm_host(self.h, _dp(Xf), Xf.shape[0], X.shape[1], _dp(Y), Y.shape[0]))
        return Y

    def __call__(self, i, j):
        """operator()(i, j): coefficient of the (mirrored) operator."""
        v = C.c_double()
        check(lib().mispec_csr_coeff(self.h, int(i), int(j), C.byref(v)))
        return v.value

    def spmv_device(self, x_ptr, y_ptr):
        """y = A x with raw DEVICE pointers (x: cols() doubles, y: local_rows())."""
        check(lib().mispec_spmv(self.h, C.c_void_p(x_ptr), C.c_void_p(y_ptr)))

    def spmv_time(self, x_ptr, y_ptr, reps):
        ms = C.c_float()
        check(lib().mispec_spmv_time(self.h, C.c_void_p(x_ptr), C.c_void_p(y_ptr), reps, C.byref(ms)))
        return ms.value

    def algorithmic_bytes(self):
        return 12.0 * self.nnz() + 4.0 * (self.local_rows() + 1) + 8.0 * self.cols() + 8.0 * self.local_rows()

    def offset_codes(self):
        """0: the SpMV reads int32 column indices; d > 0: one-byte codes into a dictionary of d diagonals."""
        return int(lib().mispec_csr_offset_codes(self.h))

    def use_offset_codes(self, enable=True):
        check(lib().mispec_csr_use_offset_codes(self.h, 1 if enable else 0))

    def use_windows(self, enable=True):
        """Per-matrix switch of the int32 CSR kernel: x staged through LDS windows (True) or gathered entry by entry (False);
        None: automatic (windows when the table was adopted at ingest and the rows hold at least 9 entries on average)."""
        check(lib().mispec_csr_use_windows(self.h, -1 if enable is None else (1 if enable else 0)))

    def windows_info(self):
        """{blocks, covered_entries, lds_doubles} of the x windows of the int32 CSR kernel (lds_doubles = 0: not adopted)."""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(lib().mispec_csr_windows_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"blocks": a.value, "covered_entries": b.value, "lds_doubles": c.value}

    def windows_in_use(self):
        """True when format 0 of this matrix runs the kernel with x windows (per-matrix switch and automatic rule applied)."""
        return bool(lib().mispec_csr_windows_in_use(self.h))

    def windows_table(self):
        """The per-block window records (blocks x 32 int32; include/mispec.h mispec_csr_windows_table)."""
        nb = (self.local_rows() + 255) // 256
        out = np.empty((nb, 32), dtype=np.int32)
        check(lib().mispec_csr_windows_table(self.h, _ip(out), out.size))
        return out

    def spmv_format(self):
        """0: CSR with int32 column indices, 1: CSR with offset codes, 2: diagonal storage — what the SpMV uses."""
        return int(lib().mispec_csr_spmv_format(self.h))

    def set_spmv_format(self, fmt=-1):
        """Force a storage format for this matrix (-1: automatic); all formats give bit-identical products."""
        check(lib().mispec_csr_set_spmv_format(self.h, int(fmt)))

    def stored_bytes(self):
        """Compulsory SpMV traffic with the index format in use (9 instead of 12 bytes per entry with offset codes)."""
        return float(lib().mispec_csr_spmv_bytes(self.h, 1))

    def staged_info(self):
        """{bins, slots, batches, chunks} of the staged format (two streaming phases with x and y in LDS; bins = 0: not built)."""
        a, b, c, d = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(lib().mispec_csr_staged_info(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"bins": a.value, "slots": b.value, "batches": c.value, "chunks": d.value}

    def tiles_info(self):
        """{segments, entries, padding, chunks} of the column-blocked tile format (segments = 0: not built)."""
        a, b, c, d = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(lib().mispec_csr_tiles_info(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"segments": a.value, "entries": b.value, "padding": c.value, "chunks": d.value}

    def reorder(self, method="rcm"):
        """Symmetric reordering of the stored matrix (mispec_csr_reorder): "rcm" always, "auto" only when it pays.  Returns
        True when the matrix is reordered afterwards.  Products and results keep the caller's index order."""
        applied = C.c_int(0)
        check(lib().mispec_csr_reorder(self.h, {"rcm": 1, "auto": -1}[method], C.byref(applied)))
        return bool(applied.value)

    def reordering(self):
        """"none" or "rcm"."""
        return {0: "none", 1: "rcm"}[int(lib().mispec_csr_reordering(self.h, None, None))]

    def reordering_info(self):
        fb, fa = C.c_double(0.0), C.c_double(0.0)
        m = int(lib().mispec_csr_reordering(self.h, C.byref(fb), C.byref(fa)))
        return {"method": {0: "none", 1: "rcm"}[m], "far_fraction_before": fb.value, "far_fraction_after": fa.value}

    def permutation(self):
        """perm[new] = old (identity when the matrix is not reordered)."""
        out = np.empty(self.rows(), dtype=np.int32)
        check(lib().mispec_csr_permutation(self.h, _ip(out)))
        return out

    def to_host_csr(self):
        nl, nnz = self.local_rows(), self.nnz()
        rp = np.empty(nl + 1, dtype=np.int32)
        ci = np.empty(nnz, dtype=np.int32)
        v = np.empty(nnz)
        check(lib().mispec_csr_download(self.h, _ip(rp), _ip(ci), _dp(v)))
        return rp, ci, v

    def __del__(self):
        try:
            lib().mispec_csr_destroy(self.h)
        except Exception:
            pass


def _compressed(mat):
    """(rows, cols, outer, inner, values, row_major) of a scipy.sparse CSR/CSC matrix."""
    fmt = getattr(mat, "format", None)
    if fmt not in ("csr", "csc"):
        raise TypeError("expected a scipy.sparse csr_matrix or csc_matrix (compressed storage)")
    m = mat
    if not mat.has_canonical_format:  # unsorted inner indices or duplicates: canonicalise a copy, never the caller's matrix
        m = mat.copy()
        m.sum_duplicates()
        m.sort_indices()
    # a canonical matrix is handed over as it is (the copy was 0.15 s of single-threaded memcpy per GB at n = 1e7)
    return m.shape[0], m.shape[1], _i32(m.indptr), _i32(m.indices), _f64(m.data), fmt == "csr"


class _reorder_env:
    """reorder=None keeps the library default (MISPEC_REORDER or "auto"); "none" / "rcm" / "auto" override it for one ingest."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.old = os.environ.get("MISPEC_REORDER")
        if self.mode is not None:
            if self.mode not in ("none", "rcm", "auto"):
                raise ValueError("reorder must be None, 'none', 'rcm' or 'auto'")
            os.environ["MISPEC_REORDER"] = self.mode

    def __exit__(self, *exc):
        if self.mode is not None:
            if self.old is None:
                os.environ.pop("MISPEC_REORDER", None)
            else:
                os.environ["MISPEC_REORDER"] = self.old


class SparseSymMatProd(_DeviceMatrix):
    """MatOp/SparseSymMatProd.h: y = selfadjointView<Uplo>(A) * x; only the `uplo` triangle of A is read."""

    def __init__(self, mat, uplo="L", ctx=None, reorder=None):
        ctx = ctx or default_context()
        n, nc, outer, inner, val, row_major = _compressed(mat)
        if n != nc:
            raise ValueError("SparseSymMatProd: matrix must be square")
        h = C.c_void_p()
        with _reorder_env(reorder):
            check(lib().mispec_csr_from_triangle(ctx.h, n, _ip(outer), _ip(inner), _dp(val), uplo.encode()[0:1], int(row_major),
                                                 C.byref(h)))
        super().__init__(ctx, h)

    @classmethod
    def synth_band(cls, n, offsets=BAND_OFFSETS, seed=SYNTH_SEED, ctx=None):
        """The synthetic symmetric benchmark matrix of SURVEY.md §8(d), generated directly in HBM."""
        return _synth(cls, n, offsets, seed, True, ctx)


class SparseGenMatProd(_DeviceMatrix):
    """MatOp/SparseGenMatProd.h: y = A * x for a general sparse A (CSR or CSC input)."""

    def __init__(self, mat, ctx=None, reorder=None):
        ctx = ctx or default_context()
        nr, nc, outer, inner, val, row_major = _compressed(mat)
        h = C.c_void_p()
        fn = lib().mispec_csr_upload if row_major else lib().mispec_csr_from_csc
        with _reorder_env(reorder):
            check(fn(ctx.h, nr, nc, _ip(outer), _ip(inner), _dp(val), C.byref(h)))
        super().__init__(ctx, h)

    @classmethod
    def synth_band(cls, n, offsets=BAND_OFFSETS, seed=SYNTH_SEED, ctx=None):
        return _synth(cls, n, offsets, seed, False, ctx)


def _synth(cls, n, offsets, seed, symmetric, ctx):
    ctx = ctx or default_context()
    offs = np.ascontiguousarray(offsets, dtype=np.int64)
    h = C.c_void_p()
    check(lib().mispec_csr_synth_band(ctx.h, n, seed, offs.ctypes.data_as(C.POINTER(C.c_int64)), len(offs), int(symmetric),
                                      C.byref(h)))
    obj = cls.__new__(cls)
    _DeviceMatrix.__init__(obj, ctx, h)
    return obj


class SparseSymShiftSolve:
    """MatOp/SparseSymShiftSolve.h: y = (A - sigma I)^{-1} x, solved on the GPU after set_shift(sigma)."""

    def __init__(self, mat, uplo="L", ctx=None):
        self.ctx = ctx or default_context()
        n, nc, outer, inner, val, row_major = _compressed(mat)
        if n != nc:
            raise ValueError("SparseSymShiftSolve: matrix must be square")
        h = C.c_void_p()
        check(lib().mispec_symshift_create(self.ctx.h, n, _ip(outer), _ip(inner), _dp(val), uplo.encode()[0:1], int(row_major),
                                           C.byref(h)))
        self.h = h
        self.n = n

    def rows(self):
        return self.n

    def cols(self):
        return self.n

    def local_rows(self):
        return self.n

    def set_shift(self, sigma):
        check(lib().mispec_symshift_set_shift(self.h, float(sigma)))

    def bandwidth_info(self):
        """{as_given, stored, reordered}: half-bandwidth of the matrix as it came and as it is stored (reverse Cuthill-McKee at
        construction when the band is too wide as given, include/mispec.h mispec_symshift_bandwidth)."""
        a, b, r = C.c_int64(0), C.c_int64(0), C.c_int(0)
        check(lib().mispec_symshift_bandwidth(self.h, C.byref(a), C.byref(b), C.byref(r)))
        return {"as_given": a.value, "stored": b.value, "reordered": bool(r.value)}

    def refinement_info(self):
        """Banded path: {refine_steps, boosted_pivots, min_pivot_ratio, probe_backward_error} of the last set_shift()."""
        st, bo, mr, om = C.c_int(0), C.c_int64(0), C.c_double(0.0), C.c_double(0.0)
        check(lib().mispec_symshift_refinement_info(self.h, C.byref(st), C.byref(bo), C.byref(mr), C.byref(om)))
        return {"refine_steps": st.value, "boosted_pivots": bo.value, "min_pivot_ratio": mr.value, "probe_backward_error": om.value}

    def perform_op(self, x_in):
        x = _f64(x_in)
        if x.shape != (self.n,):
            raise ValueError("perform_op: x_in must have n entries")
        y = np.empty(self.n)
        check(lib().mispec_symshift_solve_host(self.h, _dp(x), _dp(y)))
        return y

    def solve_device(self, x_ptr, y_ptr):
        """y = (A - sigma I)^{-1} x on device pointers, enqueued on the context's stream (mispec_symshift_solve)."""
        check(lib().mispec_symshift_solve(self.h, C.c_void_p(int(x_ptr)), C.c_void_p(int(y_ptr))))

    def __del__(self):
        try:
            lib().mispec_symshift_destroy(self.h)
        except Exception:
            pass


class _DenseMatrix:
    """A dense matrix in HBM (mispec_dense), row-major there."""

    def __init__(self, mat, uplo, ctx):
        self.ctx = ctx or default_context()
        M = np.asarray(mat, dtype=np.float64)
        if M.ndim != 2:
            raise ValueError("dense operator: a 2-d array is expected")
        row_major = M.flags.c_contiguous and not M.flags.f_contiguous
        if not (M.flags.c_contiguous or M.flags.f_contiguous):
            M = np.asfortranarray(M)
        ld = M.shape[1] if row_major else M.shape[0]
        h = C.c_void_p()
        check(lib().mispec_dense_upload(self.ctx.h, M.shape[0], M.shape[1], _dp(M), max(int(ld), 1), int(row_major),
                                        uplo.encode()[0:1] if uplo else b"\0", C.byref(h)))
        self.h = h

    def rows(self):
        return lib().mispec_dense_rows(self.h)

    def cols(self):
        return lib().mispec_dense_cols(self.h)

    def local_rows(self):
        return self.rows()

    def perform_op(self, x_in, y_out=None):
        """y_out = A * x_in with HOST arrays (the reference's perform_op contract)."""
        x = _f64(x_in)
        if x.shape != (self.cols(),):
            raise ValueError("perform_op: x_in must have cols() entries")
        y = np.empty(self.rows()) if y_out is None else y_out
        check(lib().mispec_dense_gemv_host(self.h, _dp(x), _dp(y)))
        return y

    def __matmul__(self, X):
        X = np.asarray(X, dtype=np.float64)
        if X.ndim == 1:
            return self.perform_op(X)
        Xf = np.asfortranarray(X)
        Y = np.empty((self.rows(), X.shape[1]), order="F")
        check(lib().mispec_dense_gemm_host(self.h, _dp(Xf), Xf.shape[0], X.shape[1], _dp(Y), Y.shape[0]))
        return Y

    def __call__(self, i, j):
        v = C.c_double()
        check(lib().mispec_dense_coeff(self.h, int(i), int(j), C.byref(v)))
        return v.value

    def gemv_device(self, x_ptr, y_ptr):
        check(lib().mispec_dense_gemv(self.h, C.c_void_p(x_ptr), C.c_void_p(y_ptr)))

    def gemv_time(self, x_ptr, y_ptr, reps):
        ms = C.c_float()
        check(lib().mispec_dense_gemv_time(self.h, C.c_void_p(x_ptr), C.c_void_p(y_ptr), reps, C.byref(ms)))
        return ms.value

    def algorithmic_bytes(self):
        return 8.0 * self.rows() * self.cols() + 8.0 * self.cols() + 8.0 * self.rows()

    def __del__(self):
        try:
            lib().mispec_dense_destroy(self.h)
        except Exception:
            pass


class DenseSymMatProd(_DenseMatrix):
    """MatOp/DenseSymMatProd.h: y = A x for a symmetric dense A, reading only the `uplo` triangle of the input."""

    def __init__(self, mat, uplo="L", ctx=None):
        M = np.asarray(mat)
        if M.ndim != 2 or M.shape[0] != M.shape[1]:
            raise ValueError("DenseSymMatProd: matrix must be square")
        if uplo not in ("L", "U"):
            raise ValueError("DenseSymMatProd: uplo must be 'L' or 'U'")
        super().__init__(mat, uplo, ctx)


class DenseGenMatProd(_DenseMatrix):
    """MatOp/DenseGenMatProd.h: y = A x for a general dense A."""

    def __init__(self, mat, ctx=None):
        super().__init__(mat, None, ctx)


class DeviceOp:
    """A user operator on DEVICE pointers: fn(x_ptr, y_ptr, stream) must enqueue y = Op(x) (n doubles each) on the
    given HIP stream.  The perform_op concept of the reference with both vectors in HBM: nothing is staged."""

    def __init__(self, n, fn, ctx=None):
        self.ctx = ctx or default_context()
        self.n = int(n)
        self.fn = fn

        def tramp(user, x_ptr, y_ptr, stream):
            try:
                fn(int(x_ptr or 0), int(y_ptr or 0), int(stream or 0))
                return 0
            except Exception:  # noqa: BLE001 - reported through the return code
                return 1

        self.cb = _capi.device_op_fn(tramp)

    def rows(self):
        return self.n

    def cols(self):
        return self.n

    def local_rows(self):
        return self.n


class _UserOp:
    """Adapter for a Python operator with rows(), cols(), perform_op(x_in) -> y (host numpy arrays)."""

    def __init__(self, op):
        self.op = op
        n = op.rows()

        def tramp(user, x_ptr, y_ptr):
            try:
                x = np.ctypeslib.as_array(x_ptr, shape=(n,))
                y = np.ctypeslib.as_array(y_ptr, shape=(n,))
                y[:] = op.perform_op(x)
                return 0
            except Exception:  # noqa: BLE001 - reported through the return code
                return 1

        self.cb = _capi.op_fn(tramp)


class SparseCholesky:
    """MatOp/SparseCholesky.h: B = G G' (dense factor on the GPU for n <= 4096, the partitioned band factorisation for a
    larger banded B); lower_triangular_solve = G^{-1} x, upper_triangular_solve = G^{-T} x; info() like the reference
    (NumericalIssue when B is not positive definite)."""

    def __init__(self, mat, uplo="L", ctx=None):
        self.ctx = ctx or default_context()
        n, nc, outer, inner, val, row_major = _compressed(mat)
        if n != nc:
            raise ValueError("SparseCholesky: matrix must be square")
        h = C.c_void_p()
        check(lib().mispec_cholesky_create(self.ctx.h, n, _ip(outer), _ip(inner), _dp(val), uplo.encode()[0:1], int(row_major),
                                           C.byref(h)))
        self.h = h
        self.n = n

    def rows(self):
        return self.n

    cols = rows

    def info(self):
        return CompInfo(lib().mispec_cholesky_info(self.h))

    def _solve(self, fn, x_in):
        x = _f64(x_in)
        if x.shape != (self.n,):
            raise ValueError("triangular solve: x_in must have n entries")
        y = np.empty(self.n)
        check(fn(self.h, _dp(x), _dp(y)))
        return y

    def lower_triangular_solve(self, x_in):
        return self._solve(lib().mispec_cholesky_lower_solve_host, x_in)

    def upper_triangular_solve(self, x_in):
        return self._solve(lib().mispec_cholesky_upper_solve_host, x_in)

    def __del__(self):
        try:
            lib().mispec_cholesky_destroy(self.h)
        except Exception:
            pass


class SymShiftInvert:
    """MatOp/SymShiftInvert.h (sparse A, sparse B): y = (A - sigma B)^{-1} x, factored on the GPU at set_shift(sigma)."""

    def __init__(self, A, B, uplo_a="L", uplo_b="L", ctx=None):
        self.ctx = ctx or default_context()
        n, nc, ao, ai, av, arm = _compressed(A)
        nb, ncb, bo, bi, bv, brm = _compressed(B)
        if n != nc or nb != n or ncb != n:
            raise ValueError("SymShiftInvert: A and B must be square matrices of the same size")
        h = C.c_void_p()
        check(lib().mispec_symshift_create_pencil(self.ctx.h, n, _ip(ao), _ip(ai), _dp(av), uplo_a.encode()[0:1], int(arm),
                                                  _ip(bo), _ip(bi), _dp(bv), uplo_b.encode()[0:1], int(brm), C.byref(h)))
        self.h = h
        self.n = n

    def rows(self):
        return self.n

    cols = rows

    def set_shift(self, sigma):
        check(lib().mispec_symshift_set_shift(self.h, float(sigma)))

    def perform_op(self, x_in):
        x = _f64(x_in)
        if x.shape != (self.n,):
            raise ValueError("perform_op: x_in must have n entries")
        y = np.empty(self.n)
        check(lib().mispec_symshift_solve_host(self.h, _dp(x), _dp(y)))
        return y

    def __del__(self):
        try:
            lib().mispec_symshift_destroy(self.h)
        except Exception:
            pass


class SparseRegularInverse:
    """MatOp/SparseRegularInverse.h: the B operator of a generalized problem — perform_op = B x, solve = B^{-1} x by a
    conjugate-gradient iteration on the GPU (the reference's Eigen::ConjugateGradient defaults)."""

    def __init__(self, mat, uplo="L", ctx=None):
        self.ctx = ctx or default_context()
        n, nc, outer, inner, val, row_major = _compressed(mat)
        if n != nc:
            raise ValueError("SparseRegularInverse: matrix must be square")
        h = C.c_void_p()
        check(lib().mispec_reginv_create(self.ctx.h, n, _ip(outer), _ip(inner), _dp(val), uplo.encode()[0:1], int(row_major),
                                         C.byref(h)))
        self.h = h
        self.n = n

    def rows(self):
        return self.n

    cols = rows

    def perform_op(self, x_in):
        x = _f64(x_in)
        if x.shape != (self.n,):
            raise ValueError("perform_op: x_in must have n entries")
        y = np.empty(self.n)
        check(lib().mispec_reginv_perform_op_host(self.h, _dp(x), _dp(y)))
        return y

    def solve(self, x_in):
        x = _f64(x_in)
        if x.shape != (self.n,):
            raise ValueError("solve: x_in must have n entries")
        y = np.empty(self.n)
        check(lib().mispec_reginv_solve_host(self.h, _dp(x), _dp(y)))
        return y

    def last_iterations(self):
        return int(lib().mispec_reginv_last_iterations(self.h))

    def __del__(self):
        try:
            lib().mispec_reginv_destroy(self.h)
        except Exception:
            pass


class SVDMatOp:
    """contrib/PartialSVDSolver.h:16-110: the operator A'A (tall A, SVDTallMatOp) or AA' (wide A, SVDWideMatOp) as two
    device CSR matrices applied one after the other inside the device Lanczos loop."""

    def __init__(self, mat, ctx=None):
        import scipy.sparse as sp

        self.ctx = ctx or default_context()
        mat = sp.csr_matrix(mat)
        self.m, self.n = mat.shape
        self.mat = SparseGenMatProd(mat, ctx=self.ctx)
        self.mat_t = SparseGenMatProd(sp.csr_matrix(mat.T), ctx=self.ctx)
        self.tall = self.m > self.n
        self.first, self.second = (self.mat, self.mat_t) if self.tall else (self.mat_t, self.mat)

    def rows(self):
        return min(self.m, self.n)

    cols = rows

    def local_rows(self):
        return self.rows()

    def perform_op(self, x):
        return self.second.perform_op(self.first.perform_op(x))


class PartialSVDSolver:
    """contrib/PartialSVDSolver.h:112-209: the ncomp largest singular triplets through SymEigsSolver(LargestAlge)."""

    def __init__(self, mat, ncomp, ncv, ctx=None):
        self.op = SVDMatOp(mat, ctx)
        self.eigs = SymEigsSolver(self.op, ncomp, ncv)
        self.nconv = 0
        self._evecs = None

    def compute(self, maxit=1000, tol=1e-10):
        self.eigs.init()
        self.nconv = self.eigs.compute(SortRule.LargestAlge, maxit, tol)
        self._evecs = None
        return self.nconv

    def singular_values(self):
        return np.sqrt(self.eigs.eigenvalues())

    def _vectors(self):
        if self._evecs is None:
            self._evecs = self.eigs.eigenvectors()
        return self._evecs

    def _scaled(self, A, k):
        ev = self.eigs.eigenvalues()
        return A @ np.asfortranarray(self._vectors()[:, :k] / np.sqrt(ev[:k]))

    def matrix_U(self, nu):
        nu = min(int(nu), self.nconv)
        return self._vectors()[:, :nu] if self.op.m <= self.op.n else self._scaled(self.op.mat, nu)

    def matrix_V(self, nv):
        nv = min(int(nv), self.nconv)
        return self._vectors()[:, :nv] if self.op.m > self.op.n else self._scaled(self.op.mat_t, nv)


class _GEigsRegInvOp:
    """MatOp/internal/SymGEigsRegInvOp.h: y = B^{-1} A x."""

    def __init__(self, A, B):
        if not isinstance(A, SparseSymMatProd) or not isinstance(B, SparseRegularInverse):
            raise TypeError("SymGEigsSolver: needs a SparseSymMatProd and a SparseRegularInverse")
        if A.rows() != B.rows():
            raise ValueError("SymGEigsSolver: A and B must have the same size")
        self.A, self.B = A, B

    def rows(self):
        return self.B.rows()

    cols = rows
    local_rows = rows

    def perform_op(self, x):
        return self.B.solve(self.A.perform_op(x))


class SymEigsSolver:
    """SymEigsSolver.h:133-160 / HermEigsBase.h: init(), compute(), info(), eigenvalues(), eigenvectors() ..."""

    def __init__(self, op, nev, ncv, ctx=None):
        self.op = op
        h = C.c_void_p()
        if isinstance(op, _DeviceMatrix):
            self.ctx = op.ctx
            check(lib().mispec_symeigs_create(self.ctx.h, op.h, int(nev), int(ncv), C.byref(h)))
            self._user = None
        elif isinstance(op, _GEigsRegInvOp):  # y = B^{-1}(A x) in the B-inner product
            self.ctx = op.A.ctx
            check(lib().mispec_symeigs_create_geigs_reginv(self.ctx.h, op.A.h, op.B.h, int(nev), int(ncv), C.byref(h)))
            self._user = None
        elif isinstance(op, _GEigsCholeskyOp):  # y = L^{-1} A L^{-T} x
            self.ctx = op.A.ctx
            check(lib().mispec_symeigs_create_geigs_cholesky(self.ctx.h, op.A.h, op.B.h, int(nev), int(ncv), C.byref(h)))
            self._user = None
        elif isinstance(op, _GEigsShiftOp):  # y = (A - sigma B)^{-1} B x (+ Cayley), B-inner product
            self.ctx = op.S.ctx
            check(lib().mispec_symeigs_create_geigs_shift(self.ctx.h, op.S.h, op.B.h, op.mode, int(nev), int(ncv), float(op.sigma),
                                                          C.byref(h)))
            self._user = None
        elif isinstance(op, SVDMatOp):  # y = A2 (A x), both factors in HBM
            self.ctx = op.ctx
            check(lib().mispec_symeigs_create_product(self.ctx.h, op.first.h, op.second.h, int(nev), int(ncv), C.byref(h)))
            self._user = None
        elif isinstance(op, _DenseMatrix):  # dense GEMV on the device
            self.ctx = op.ctx
            check(lib().mispec_symeigs_create_dense(self.ctx.h, op.h, int(nev), int(ncv), C.byref(h)))
            self._user = None
        elif isinstance(op, DeviceOp):  # user operator on device pointers
            self.ctx = op.ctx
            check(lib().mispec_symeigs_create_device_op(self.ctx.h, op.cb, None, op.n, int(nev), int(ncv), C.byref(h)))
            self._user = None
        else:  # any object with rows(), cols(), perform_op(x) -> y: the reference's OpType concept
            self.ctx = ctx or default_context()
            self._user = _UserOp(op)
            check(lib().mispec_symeigs_create_op(self.ctx.h, self._user.cb, None, int(op.rows()), int(nev), int(ncv),
                                                 C.byref(h)))
        self.h = h
        self.nev, self.ncv = int(nev), int(ncv)

    def init(self, init_resid=None):
        v0 = None if init_resid is None else _f64(init_resid)
        if v0 is not None and v0.shape != (self.op.rows(),):
            raise ValueError("init: the initial residual vector must have n entries")
        check(lib().mispec_symeigs_init(self.h, _dp(v0)))

    def compute(self, selection=SortRule.LargestMagn, maxit=1000, tol=1e-10, sorting=SortRule.LargestAlge):
        nconv = C.c_int64()
        check(lib().mispec_symeigs_compute(self.h, int(selection), int(maxit), float(tol), int(sorting), C.byref(nconv)))
        return nconv.value

    def info(self):
        return CompInfo(lib().mispec_symeigs_info(self.h))

    def num_iterations(self):
        return lib().mispec_symeigs_num_iterations(self.h)

    def num_operations(self):
        return lib().mispec_symeigs_num_operations(self.h)

    def eigenvalues(self):
        out = np.empty(self.nev)
        cnt = C.c_int64()
        check(lib().mispec_symeigs_eigenvalues(self.h, _dp(out), C.byref(cnt)))
        return out[:cnt.value].copy()

    def local_rows(self):
        return self.op.local_rows() if isinstance(self.op, (_DeviceMatrix, SVDMatOp, _GEigsRegInvOp, _GEigsShiftOp, _GEigsCholeskyOp,
                                                            _DenseMatrix, DeviceOp)) else self.op.rows()

    def eigenvectors(self, nvec=None, to_host=True, out=None):
        """n x nconv (this shard's rows).  to_host=False leaves the result in HBM and returns the column count.  out: a
        column-major float64 array of at least local_rows x min(nvec, nev) to be filled instead of a fresh one (the reference's
        eigenvectors() returns a new matrix every time; a caller that solves repeatedly saves the first-touch page faults of
        1.6 GB at n = 1e7 by handing the previous result back — SymEigsSolver::eigenvectors_to in the C++ headers)."""
        nvec = self.nev if nvec is None else int(nvec)
        cnt = C.c_int64()
        if not to_host:
            check(lib().mispec_symeigs_eigenvectors(self.h, nvec, None, C.byref(cnt)))
            return cnt.value
        ncol = max(min(nvec, self.nev), 1)
        if out is None:
            out = np.empty((self.local_rows(), ncol), order="F")  # filled by the library's host threads
        elif not (isinstance(out, np.ndarray) and out.dtype == np.float64 and out.flags.f_contiguous and out.ndim == 2 and
                  out.shape[0] == self.local_rows() and out.shape[1] >= ncol):
            raise ValueError("eigenvectors(out=...): need a column-major float64 array of local_rows x >= min(nvec, nev)")
        check(lib().mispec_symeigs_eigenvectors(self.h, nvec, _dp(out), C.byref(cnt)))
        return out[:, :cnt.value] if out.shape[1] != cnt.value else out

    def residuals(self):
        """||A x - lambda x|| / ||x|| of the converged pairs, evaluated on the device."""
        out = np.empty(self.nev)
        cnt = C.c_int64()
        check(lib().mispec_symeigs_residuals(self.h, _dp(out), C.byref(cnt)))
        return out[:cnt.value].copy()

    def profile(self, enable):
        check(lib().mispec_symeigs_profile(self.h, int(enable)))  # 0 off, 1 all families, 2 SpMV only

    def get_profile(self):
        p = Profile()
        check(lib().mispec_symeigs_get_profile(self.h, C.byref(p)))
        return p.as_dict()

    def set_orth_mode(self, mode):
        """'onesweep' (default: the correction of a step rides on the next step's pass over V, include/mispec.h
        mispec_fac_set_orth_mode) or 'reference' (Lanczos.h:145-181, two passes over V per step; also MISPEC_ORTH=reference in
        the environment).  Call before init().  Where the one-sweep steps do not apply (ncv > 128, generalized problems, host-pointer user
        operators) the reference flow runs whatever is set: orth_info()['mode'] says which one is in effect."""
        check(lib().mispec_symeigs_set_orth_mode(self.h, _orth_mode_value(mode)))

    def orth_info(self):
        mode, a, b, c = C.c_int(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        r, k = C.c_double(0.0), C.c_double(0.0)
        check(lib().mispec_symeigs_orth_info(self.h, C.byref(mode), C.byref(a), C.byref(b), C.byref(c), C.byref(r), C.byref(k)))
        fused, again = C.c_int64(0), C.c_int64(0)
        check(lib().mispec_symeigs_restart_info(self.h, C.byref(fused), C.byref(again)))
        ored = C.c_int64(0)
        check(lib().mispec_symeigs_onered_steps(self.h, C.byref(ored)))
        return {"mode": "onesweep" if (mode.value & 0xFF) else "reference", "eager_last": bool(mode.value & 0x100),
                "recorrect_hook": bool(mode.value & 0x200), "one_reduction": bool(mode.value & 0x800), "lagged_steps": a.value, "check_stops": b.value,
                "state_stops": c.value, "max_rel_c": r.value, "max_chk": k.value, "fused_restarts": fused.value,
                "fused_recorrected": again.value, "one_reduction_steps": ored.value}

    def turn_info(self):
        """Host turns of the restarts: count, host seconds between 'sweep state seen' and 'restart enqueued', copy fallbacks."""
        turns, fb, sec = C.c_int64(0), C.c_int64(0), C.c_double(0.0)
        check(lib().mispec_symeigs_turn_info(self.h, C.byref(turns), C.byref(sec), C.byref(fb)))
        return {"turns": turns.value, "host_seconds": sec.value, "fallbacks": fb.value}

    def overlap_info(self):
        """(first interior 256-row block, interior blocks, all blocks): what is multiplied while the exchange is in flight."""
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib().mispec_symeigs_overlap_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def exchange_info(self):
        """(halo, doubles received per product): how a sharded matrix moves the Krylov vector (mispec_fac_exchange_info)."""
        halo, cnt = C.c_int(0), C.c_int64(0)
        check(lib().mispec_symeigs_exchange_info(self.h, C.byref(halo), C.byref(cnt)))
        return bool(halo.value), int(cnt.value)

    def __del__(self):
        try:
            lib().mispec_symeigs_destroy(self.h)
        except Exception:
            pass


class DavidsonSymEigsSolver:
    """DavidsonSymEigsSolver.h:18-90 / JDSymEigsBase.h:28-187: block Davidson with the diagonal (DPR) correction.
    op: SparseSymMatProd, DenseSymMatProd, or a DeviceOp together with `diagonal` (the solver needs op(i, i))."""

    def __init__(self, op, nev, nvec_init=None, nvec_max=None, diagonal=None):
        nev = int(nev)
        nvec_init = 2 * nev if nvec_init is None else int(nvec_init)
        nvec_max = 10 * nev if nvec_max is None else int(nvec_max)
        self.op = op
        self.ctx = op.ctx
        self.nev = nev
        h = C.c_void_p()
        if isinstance(op, _DeviceMatrix):
            check(lib().mispec_davidson_create(self.ctx.h, op.h, nev, nvec_init, nvec_max, C.byref(h)))
        elif isinstance(op, _DenseMatrix):
            check(lib().mispec_davidson_create_dense(self.ctx.h, op.h, nev, nvec_init, nvec_max, C.byref(h)))
        elif isinstance(op, DeviceOp):
            if diagonal is None:
                raise ValueError("DavidsonSymEigsSolver: a device operator needs its diagonal")
            d = _f64(diagonal)
            if d.shape != (op.n,):
                raise ValueError("DavidsonSymEigsSolver: diagonal must have n entries")
            check(lib().mispec_davidson_create_device_op(self.ctx.h, op.cb, None, op.n, _dp(d), nev, nvec_init, nvec_max, C.byref(h)))
        else:
            raise TypeError("DavidsonSymEigsSolver: the operator must live on the device (SparseSymMatProd, DenseSymMatProd, DeviceOp)")
        self.h = h

    def set_max_search_space_size(self, m):
        check(lib().mispec_davidson_set_sizes(self.h, -1, int(m), -1))

    def set_correction_size(self, m):
        check(lib().mispec_davidson_set_sizes(self.h, -1, -1, int(m)))

    def set_initial_search_space_size(self, m):
        check(lib().mispec_davidson_set_sizes(self.h, int(m), -1, -1))

    def sizes(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib().mispec_davidson_get_sizes(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def compute(self, selection=SortRule.LargestMagn, maxit=100, tol=1e-10):
        n = C.c_int64()
        check(lib().mispec_davidson_compute(self.h, int(selection), int(maxit), float(tol), None, 0, 0, C.byref(n)))
        return n.value

    def compute_with_guess(self, initial_space, selection=SortRule.LargestMagn, maxit=100, tol=1e-10):
        G = np.asfortranarray(initial_space, dtype=np.float64)
        if G.ndim != 2 or G.shape[0] != self.op.rows():
            raise ValueError("compute_with_guess: the initial space must have n rows")
        n = C.c_int64()
        check(lib().mispec_davidson_compute(self.h, int(selection), int(maxit), float(tol), _dp(G), G.shape[1], G.shape[0], C.byref(n)))
        return n.value

    def info(self):
        return CompInfo(lib().mispec_davidson_info(self.h))

    def num_iterations(self):
        return lib().mispec_davidson_num_iterations(self.h)

    def num_operations(self):
        return lib().mispec_davidson_num_operations(self.h)

    def eigenvalues(self):
        out = np.empty(self.nev)
        check(lib().mispec_davidson_eigenvalues(self.h, _dp(out)))
        return out

    def eigenvectors(self):
        n = self.op.rows()
        out = np.empty((n, self.nev), order="F")
        check(lib().mispec_davidson_eigenvectors(self.h, _dp(out), n))
        return out

    def __del__(self):
        try:
            lib().mispec_davidson_destroy(self.h)
        except Exception:
            pass


class _GEigsCholeskyOp:
    """MatOp/internal/SymGEigsCholeskyOp.h: y = L^{-1} A L^{-T} x."""

    def __init__(self, A, B):
        if not isinstance(A, SparseSymMatProd) or not isinstance(B, SparseCholesky):
            raise TypeError("SymGEigsSolver (Cholesky mode): needs a SparseSymMatProd and a SparseCholesky")
        if A.rows() != B.rows():
            raise ValueError("SymGEigsSolver: A and B must have the same size")
        self.A, self.B = A, B

    def rows(self):
        return self.B.rows()

    cols = rows
    local_rows = rows

    def perform_op(self, x):
        return self.B.lower_triangular_solve(self.A.perform_op(self.B.upper_triangular_solve(x)))


class _GEigsShiftOp:
    """MatOp/internal/SymGEigs{ShiftInvert,Buckling,Cayley}Op.h."""

    MODES = {"ShiftInvert": 0, "Buckling": 1, "Cayley": 2}

    def __init__(self, S, B, sigma, mode):
        if not isinstance(S, SymShiftInvert) or not isinstance(B, SparseSymMatProd):
            raise TypeError("SymGEigsShiftSolver: needs a SymShiftInvert and a SparseSymMatProd")
        if mode not in self.MODES:
            raise ValueError("SymGEigsShiftSolver: mode must be ShiftInvert, Buckling or Cayley")
        if S.rows() != B.rows():
            raise ValueError("SymGEigsShiftSolver: the operators must have the same size")
        self.S, self.B, self.sigma, self.mode = S, B, float(sigma), self.MODES[mode]

    def rows(self):
        return self.S.rows()

    cols = rows
    local_rows = rows


class SymGEigsShiftSolver(SymEigsSolver):
    """SymGEigsShiftSolver<SymShiftInvert, SparseSymMatProd, mode> (SymGEigsShiftSolver.h:36-207): eigenvalues of the pencil
    near sigma.  op = SymShiftInvert(A, B) (for Buckling: (K, KG)), Bop = SparseSymMatProd of the inner-product matrix
    (B, or K for Buckling).  set_shift(sigma) is called on op by the solver."""

    def __init__(self, op, Bop, nev, ncv, sigma, mode="ShiftInvert"):
        super().__init__(_GEigsShiftOp(op, Bop, sigma, mode), nev, ncv)


class SymGEigsSolver(SymEigsSolver):
    """SymGEigsSolver<OpType, BOpType, mode> (SymGEigsSolver.h:142-238): A x = lambda B x.
    mode "RegularInverse": Bop = SparseRegularInverse(B), B-orthonormal eigenvectors straight from the Lanczos basis;
    mode "Cholesky": Bop = SparseCholesky(B), standard problem L^{-1} A L^{-T}, eigenvectors back-transformed by L^{-T}."""

    def __init__(self, op, Bop, nev, ncv, mode="RegularInverse"):
        if mode == "RegularInverse":
            super().__init__(_GEigsRegInvOp(op, Bop), nev, ncv)
        elif mode == "Cholesky":
            super().__init__(_GEigsCholeskyOp(op, Bop), nev, ncv)
        else:
            raise ValueError("SymGEigsSolver: mode must be RegularInverse or Cholesky (the shift modes are SymGEigsShiftSolver)")


class SymEigsShiftSolver(SymEigsSolver):
    """SymEigsShiftSolver.h:190-195: eigenvalues closest to sigma via (A - sigma I)^{-1}; the constructor calls
    op.set_shift(sigma) and eigenvalues() are mapped back (lambda = 1/nu + sigma)."""

    def __init__(self, op, nev, ncv, sigma, ctx=None):
        if isinstance(op, SparseSymShiftSolve):
            self.op = op
            self.ctx = op.ctx
            self._user = None
            h = C.c_void_p()
            check(lib().mispec_symeigs_create_shift(self.ctx.h, op.h, int(nev), int(ncv), float(sigma), C.byref(h)))
            self.h = h
            self.nev, self.ncv = int(nev), int(ncv)
        else:
            raise TypeError("SymEigsShiftSolver: pass a SparseSymShiftSolve operator")


class SparseGenRealShiftSolve:
    """MatOp/SparseGenRealShiftSolve.h: y = (A - sigma I)^{-1} x for a general sparse A (dense device factorisation,
    n <= 4096)."""

    def __init__(self, mat, ctx=None):
        self.ctx = ctx or default_context()
        n, nc, outer, inner, val, row_major = _compressed(mat)
        if n != nc:
            raise ValueError("SparseGenRealShiftSolve: matrix must be square")
        h = C.c_void_p()
        check(lib().mispec_symshift_create_general(self.ctx.h, n, _ip(outer), _ip(inner), _dp(val), int(row_major), C.byref(h)))
        self.h = h
        self.n = n

    def rows(self):
        return self.n

    cols = rows
    local_rows = rows

    def set_shift(self, sigma):
        check(lib().mispec_symshift_set_shift(self.h, float(sigma)))

    def perform_op(self, x_in):
        x = _f64(x_in)
        if x.shape != (self.n,):
            raise ValueError("perform_op: x_in must have n entries")
        y = np.empty(self.n)
        check(lib().mispec_symshift_solve_host(self.h, _dp(x), _dp(y)))
        return y

    def __del__(self):
        try:
            lib().mispec_symshift_destroy(self.h)
        except Exception:
            pass


class _GenComplexShiftBinding:
    def __init__(self, S, sigmar, sigmai):
        if not isinstance(S, SparseGenRealShiftSolve):
            raise TypeError("GenEigsComplexShiftSolver: needs a SparseGenComplexShiftSolve / DenseGenComplexShiftSolve operator")
        self.S, self.sigmar, self.sigmai = S, float(sigmar), float(sigmai)

    def rows(self):
        return self.S.rows()

    cols = rows
    local_rows = rows


class _GenShiftBinding:
    def __init__(self, S, sigma):
        if not isinstance(S, SparseGenRealShiftSolve):
            raise TypeError("GenEigsRealShiftSolver: needs a SparseGenRealShiftSolve")
        self.S, self.sigma = S, float(sigma)

    def rows(self):
        return self.S.rows()

    cols = rows
    local_rows = rows


class GenEigsSolver:
    """GenEigsSolver.h:139-186 / GenEigsBase.h: complex eigenvalues / eigenvectors of a general real matrix."""

    def __init__(self, op, nev, ncv, ctx=None):
        self.op = op
        h = C.c_void_p()
        if isinstance(op, _DeviceMatrix):
            self.ctx = op.ctx
            check(lib().mispec_geneigs_create(self.ctx.h, op.h, int(nev), int(ncv), C.byref(h)))
            self._user = None
        elif isinstance(op, _GenComplexShiftBinding):  # Arnoldi on Re((A - sigma I)^{-1} .), complex sigma
            self.ctx = op.S.ctx
            check(lib().mispec_geneigs_create_complex_shift(self.ctx.h, op.S.h, int(nev), int(ncv), float(op.sigmar), float(op.sigmai),
                                                            C.byref(h)))
            self._user = None
        elif isinstance(op, _GenShiftBinding):  # Arnoldi on (A - sigma I)^{-1}
            self.ctx = op.S.ctx
            check(lib().mispec_geneigs_create_shift(self.ctx.h, op.S.h, int(nev), int(ncv), float(op.sigma), C.byref(h)))
            self._user = None
        elif isinstance(op, _DenseMatrix):
            self.ctx = op.ctx
            check(lib().mispec_geneigs_create_dense(self.ctx.h, op.h, int(nev), int(ncv), C.byref(h)))
            self._user = None
        elif isinstance(op, DeviceOp):
            self.ctx = op.ctx
            check(lib().mispec_geneigs_create_device_op(self.ctx.h, op.cb, None, op.n, int(nev), int(ncv), C.byref(h)))
            self._user = None
        else:
            self.ctx = ctx or default_context()
            self._user = _UserOp(op)
            check(lib().mispec_geneigs_create_op(self.ctx.h, self._user.cb, None, int(op.rows()), int(nev), int(ncv),
                                                 C.byref(h)))
        self.h = h
        self.nev, self.ncv = int(nev), int(ncv)

    def init(self, init_resid=None):
        v0 = None if init_resid is None else _f64(init_resid)
        if v0 is not None and v0.shape != (self.op.rows(),):
            raise ValueError("init: the initial residual vector must have n entries")
        check(lib().mispec_geneigs_init(self.h, _dp(v0)))

    def compute(self, selection=SortRule.LargestMagn, maxit=1000, tol=1e-10, sorting=SortRule.LargestMagn):
        nconv = C.c_int64()
        check(lib().mispec_geneigs_compute(self.h, int(selection), int(maxit), float(tol), int(sorting), C.byref(nconv)))
        return nconv.value

    def info(self):
        return CompInfo(lib().mispec_geneigs_info(self.h))

    def num_iterations(self):
        return lib().mispec_geneigs_num_iterations(self.h)

    def num_operations(self):
        return lib().mispec_geneigs_num_operations(self.h)

    def eigenvalues(self):
        out = np.empty(self.nev, dtype=np.complex128)
        cnt = C.c_int64()
        check(lib().mispec_geneigs_eigenvalues(self.h, out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(cnt)))
        return out[:cnt.value].copy()

    def eigenvectors(self, nvec=None):
        nvec = self.nev if nvec is None else int(nvec)
        rows = self.op.local_rows() if isinstance(self.op, _DeviceMatrix) else self.op.rows()
        out = np.zeros((rows, max(min(nvec, self.nev), 1)), dtype=np.complex128, order="F")
        cnt = C.c_int64()
        check(lib().mispec_geneigs_eigenvectors(self.h, nvec, out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(cnt)))
        return np.asfortranarray(out[:, :cnt.value])

    def residuals(self):
        """||A x - lambda x|| / ||x|| of the converged (complex) pairs, evaluated on the device."""
        out = np.empty(self.nev)
        cnt = C.c_int64()
        check(lib().mispec_geneigs_residuals(self.h, _dp(out), C.byref(cnt)))
        return out[:cnt.value].copy()

    def profile(self, enable):
        check(lib().mispec_geneigs_profile(self.h, int(enable)))

    def get_profile(self):
        p = Profile()
        check(lib().mispec_geneigs_get_profile(self.h, C.byref(p)))
        return p.as_dict()

    def __del__(self):
        try:
            lib().mispec_geneigs_destroy(self.h)
        except Exception:
            pass



class GenEigsRealShiftSolver(GenEigsSolver):
    """GenEigsRealShiftSolver.h:36-82: eigenvalues of a general real A closest to the real shift sigma; op is a
    SparseGenRealShiftSolve (the solver calls set_shift(sigma)); lambda = 1/nu + sigma."""

    def __init__(self, op, nev, ncv, sigma):
        super().__init__(_GenShiftBinding(op, sigma), nev, ncv)

class SparseGenComplexShiftSolve(SparseGenRealShiftSolve):
    """MatOp/SparseGenComplexShiftSolve.h: y = Re((A - sigma I)^{-1} x) for a general sparse A and a complex shift (n <= 4096)."""

    def set_shift(self, sigmar, sigmai=0.0):
        check(lib().mispec_symshift_set_shift_complex(self.h, float(sigmar), float(sigmai)))


class GenEigsComplexShiftSolver(GenEigsSolver):
    """GenEigsComplexShiftSolver.h:20-150: eigenvalues of a general real A closest to the complex shift sigmar + i sigmai."""

    def __init__(self, op, nev, ncv, sigmar, sigmai):
        super().__init__(_GenComplexShiftBinding(op, sigmar, sigmai), nev, ncv)


def _dense_as_csc(mat, name):
    """The compressed form of a dense host matrix (exact zeros dropped): what the device factorisations ingest."""
    import scipy.sparse as sp

    M = np.asarray(mat, dtype=np.float64)
    if M.ndim != 2 or M.shape[0] != M.shape[1]:
        raise ValueError(f"{name}: matrix must be square")
    return sp.csc_matrix(M)


class DenseSymShiftSolve(SparseSymShiftSolve):
    """MatOp/DenseSymShiftSolve.h: y = (A - sigma I)^{-1} x for a dense symmetric A (the `uplo` triangle is read)."""

    def __init__(self, mat, uplo="L", ctx=None):
        super().__init__(_dense_as_csc(mat, "DenseSymShiftSolve"), uplo, ctx)


class DenseGenRealShiftSolve(SparseGenRealShiftSolve):
    """MatOp/DenseGenRealShiftSolve.h: y = (A - sigma I)^{-1} x for a general dense A and a real shift (n <= 4096)."""

    def __init__(self, mat, ctx=None):
        super().__init__(_dense_as_csc(mat, "DenseGenRealShiftSolve"), ctx)


class DenseGenComplexShiftSolve(SparseGenComplexShiftSolve):
    """MatOp/DenseGenComplexShiftSolve.h: the dense form of SparseGenComplexShiftSolve."""

    def __init__(self, mat, ctx=None):
        super().__init__(_dense_as_csc(mat, "DenseGenComplexShiftSolve"), ctx)


class DenseCholesky(SparseCholesky):
    """MatOp/DenseCholesky.h: B = L L' for a dense positive-definite B (n <= 4096)."""

    def __init__(self, mat, uplo="L", ctx=None):
        super().__init__(_dense_as_csc(mat, "DenseCholesky"), uplo, ctx)


def hess_qr(H, shift):
    """UpperHessenbergQR on the host (the general restart's real-shift step): returns (Q, Q'HQ)."""
    H = np.asfortranarray(H, dtype=np.float64)
    n = H.shape[0]
    Q, D = np.empty((n, n), order="F"), np.empty((n, n), order="F")
    check(lib().mispec_hess_qr_host(n, _dp(H), float(shift), _dp(Q), _dp(D)))
    return Q, D


def double_shift_qr(H, s, t):
    """DoubleShiftQR on the host: H^2 - sH + tI = QR; returns (Q, Q'HQ)."""
    H = np.asfortranarray(H, dtype=np.float64)
    n = H.shape[0]
    Q, D = np.empty((n, n), order="F"), np.empty((n, n), order="F")
    check(lib().mispec_double_shift_qr_host(n, _dp(H), float(s), float(t), _dp(Q), _dp(D)))
    return Q, D


def _sweep(fn, H, *args):
    H = np.asfortranarray(H, dtype=np.float64)
    n = H.shape[0]
    Q, D = np.empty((n, n), order="F"), np.empty((n, n), order="F")
    check(fn(*args[:1], n, _dp(H), *args[1:], _dp(Q), _dp(D)) if args and hasattr(args[0], "value") else fn(n, _dp(H), *args, _dp(Q), _dp(D)))
    return Q, D


def hess_qr_device(H, shift, ctx=None):
    """UpperHessenbergQR as a HIP kernel (k_hess_restart, one real shift): returns (Q, Q'HQ).  3 <= n <= 96."""
    ctx = ctx or default_context()
    return _sweep(lib().mispec_hess_qr, H, ctx.h, float(shift))


def double_shift_qr_device(H, s, t, ctx=None):
    """DoubleShiftQR (one Francis step) as a HIP kernel: returns (Q, Q'HQ).  3 <= n <= 96."""
    ctx = ctx or default_context()
    return _sweep(lib().mispec_double_shift_qr, H, ctx.h, float(s), float(t))


def hess_qr_lanes_host(H, shift):
    """The kernel's source (internal/SmallDenseGenLanes.h) run with one lane on the host."""
    return _sweep(lib().mispec_hess_qr_lanes_host, H, float(shift))


def double_shift_qr_lanes_host(H, s, t):
    return _sweep(lib().mispec_double_shift_qr_lanes_host, H, float(s), float(t))


def hess_schur(H):
    """UpperHessenbergSchur on the host: returns (T, U) with H = U T U'."""
    H = np.asfortranarray(H, dtype=np.float64)
    n = H.shape[0]
    T, U = np.empty((n, n), order="F"), np.empty((n, n), order="F")
    check(lib().mispec_hess_schur_host(n, _dp(H), _dp(T), _dp(U)))
    return T, U


def hess_eigen(H):
    """UpperHessenbergEigen on the host: complex (eigenvalues, eigenvectors)."""
    H = np.asfortranarray(H, dtype=np.float64)
    n = H.shape[0]
    ev = np.empty(n, dtype=np.complex128)
    V = np.empty((n, n), dtype=np.complex128, order="F")
    check(lib().mispec_hess_eigen_host(n, _dp(H), ev.ctypes.data_as(C.POINTER(C.c_double)),
                                       V.ctypes.data_as(C.POINTER(C.c_double))))
    return ev, V


class Factorization:
    """LinAlg/Arnoldi.h + LinAlg/Lanczos.h on the device (mispec_fac), for the L2 parity tests."""

    def __init__(self, op, m, symmetric=True, ctx=None):
        self.op = op
        self.m = int(m)
        h = C.c_void_p()
        if isinstance(op, _DeviceMatrix):
            self.ctx = op.ctx
            self.n = op.rows()
            check(lib().mispec_fac_create(self.ctx.h, op.h, _capi.op_fn(), None, self.n, self.m, int(symmetric), C.byref(h)))
            self._user = None
        elif isinstance(op, _DenseMatrix):
            self.ctx = op.ctx
            self.n = op.rows()
            check(lib().mispec_fac_create_dense(self.ctx.h, op.h, self.m, int(symmetric), C.byref(h)))
            self._user = None
        elif isinstance(op, DeviceOp):
            self.ctx = op.ctx
            self.n = op.n
            check(lib().mispec_fac_create_device_op(self.ctx.h, op.cb, None, self.n, self.m, int(symmetric), C.byref(h)))
            self._user = None
        else:
            self.ctx = ctx or default_context()
            self.n = op.rows()
            self._user = _UserOp(op)
            check(lib().mispec_fac_create(self.ctx.h, None, self._user.cb, None, self.n, self.m, int(symmetric), C.byref(h)))
        self.h = h
        self.nmatop = C.c_int64(0)

    def init(self, v0):
        v0 = _f64(v0)
        check(lib().mispec_fac_init(self.h, _dp(v0), C.byref(self.nmatop)))

    def set_orth_mode(self, mode):
        """'onesweep' (default) or 'reference' (mispec_fac_set_orth_mode); call before factorize_from."""
        check(lib().mispec_fac_set_orth_mode(self.h, _orth_mode_value(mode)))

    def init_random(self, seed=0):
        check(lib().mispec_fac_init_random(self.h, seed, C.byref(self.nmatop)))

    def factorize_from(self, from_k, to_m):
        check(lib().mispec_fac_factorize(self.h, int(from_k), int(to_m), C.byref(self.nmatop)))

    def subspace_dim(self):
        return lib().mispec_fac_subspace_dim(self.h)

    def num_operations(self):
        return self.nmatop.value

    def f_norm(self):
        b = C.c_double()
        check(lib().mispec_fac_f_norm(self.h, C.byref(b)))
        return b.value

    def local_rows(self):
        return lib().mispec_fac_local_rows(self.h)

    def matrix_H(self):
        H = np.empty((self.m, self.m), order="F")
        check(lib().mispec_fac_get_H(self.h, _dp(H)))
        return H

    def matrix_V(self, ncols=None):
        ncols = self.m if ncols is None else ncols
        V = np.empty((self.local_rows(), ncols), order="F")
        check(lib().mispec_fac_get_V(self.h, ncols, _dp(V)))
        return V

    def vector_f(self):
        f = np.empty(self.local_rows())
        check(lib().mispec_fac_get_f(self.h, _dp(f)))
        return f

    def tridiag_eigen(self):
        ev = np.empty(self.m)
        U = np.empty((self.m, self.m), order="F")
        check(lib().mispec_fac_tridiag_eigen(self.h, _dp(ev), _dp(U)))
        return ev, U

    def restart_sym(self, shifts):
        s = _f64(shifts)
        check(lib().mispec_fac_restart_sym(self.h, _dp(s), len(s)))

    def restart_info(self):
        fused, again = C.c_int64(0), C.c_int64(0)
        check(lib().mispec_fac_restart_info(self.h, C.byref(fused), C.byref(again)))
        return {"fused_restarts": fused.value, "fused_recorrected": again.value}

    def turn_info(self):
        turns, fb, sec = C.c_int64(0), C.c_int64(0), C.c_double(0.0)
        check(lib().mispec_fac_turn_info(self.h, C.byref(turns), C.byref(sec), C.byref(fb)))
        return {"turns": turns.value, "host_seconds": sec.value, "fallbacks": fb.value}

    def compress_V(self, Q, H, new_k):
        Q = np.asfortranarray(Q, dtype=np.float64)
        H = np.asfortranarray(H, dtype=np.float64)
        check(lib().mispec_fac_compress_V(self.h, _dp(Q), _dp(H), int(new_k)))

    def ritz_vectors(self, Y):
        Y = np.asfortranarray(Y, dtype=np.float64)
        X = np.empty((self.local_rows(), Y.shape[1]), order="F")
        check(lib().mispec_fac_ritz_vectors(self.h, _dp(Y), Y.shape[1], _dp(X), None))
        return X

    def residuals(self, lam):
        lam = _f64(lam)
        out = np.empty(len(lam))
        check(lib().mispec_fac_residuals(self.h, _dp(lam), len(lam), _dp(out)))
        return out

    def profile(self, enable):
        check(lib().mispec_fac_profile(self.h, int(enable)))

    def get_profile(self):
        p = Profile()
        check(lib().mispec_fac_get_profile(self.h, C.byref(p)))
        return p.as_dict()

    def exchange_info(self):
        halo, cnt = C.c_int(0), C.c_int64(0)
        check(lib().mispec_fac_exchange_info(self.h, C.byref(halo), C.byref(cnt)))
        return bool(halo.value), int(cnt.value)

    def __del__(self):
        try:
            lib().mispec_fac_destroy(self.h)
        except Exception:
            pass


def tridiag_qr(T, shift, ctx=None):
    """TridiagQR on the device (single-workgroup LDS kernel): returns (Q, Q'TQ), both (n, n)."""
    ctx = ctx or default_context()
    T = np.asfortranarray(T, dtype=np.float64)
    n = T.shape[0]
    Q = np.empty((n, n), order="F")
    D = np.empty((n, n), order="F")
    check(lib().mispec_tridiag_qr(ctx.h, n, _dp(T), float(shift), _dp(Q), _dp(D)))
    return Q, D


RESTART_SWEEP_VARIANTS = {"host-serial": 0, "host-pipelined": 1, "device-pipelined": 2, "device-wavefront": 3}


def restart_sweeps(diag, subd, shifts, variant="host-pipelined", reps=1, ctx=None):
    """All shifted QR sweeps of one restart on the tridiagonal (diag[n], subd[n-1]) — HermEigsBase.h:124-147 — by variant
    (RESTART_SWEEP_VARIANTS; mispec_restart_sweeps).  Returns (diag', subd', Q, microseconds per call); the host variants and
    the pipelined kernel agree bit for bit.  The host variants need no device."""
    v = RESTART_SWEEP_VARIANTS[variant]
    d, e, mu = _f64(diag), _f64(subd), _f64(shifts)
    n = d.size
    h = None
    if v >= 2:
        h = (ctx or default_context()).h
    do, eo, Q = np.empty(n), np.empty(max(n - 1, 0)), np.empty((n, n), order="F")
    us = C.c_double(0.0)
    check(lib().mispec_restart_sweeps(h, n, _dp(d), _dp(e), _dp(mu), mu.size, v, int(reps), _dp(do), _dp(eo), _dp(Q), C.byref(us)))
    return do, eo, Q, us.value


def tridiag_eigen(T, ctx=None):
    """TridiagEigen on the device: returns (eigenvalues, eigenvectors)."""
    ctx = ctx or default_context()
    T = np.asfortranarray(T, dtype=np.float64)
    n = T.shape[0]
    ev = np.empty(n)
    U = np.empty((n, n), order="F")
    check(lib().mispec_tridiag_eigen(ctx.h, n, _dp(T), _dp(ev), _dp(U)))
    return ev, U
