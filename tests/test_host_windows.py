"""The x windows of the int32 CSR kernel (csr.hip k_build_windows / k_spmv_csr_win) on the HOST: `mispec_csr_windows_host` runs the
selection code the device builder runs (win_find_runs / win_select are __host__ __device__), so the rules — which runs of x a
256-row block stages through LDS, which entries keep the gather — are tested without a GPU; tests/test_gpu_windows.py requires
the device's table to equal this one and checks the products."""
import numpy as np
import scipy.sparse as sp

import spectra_amd as sa
from spectra_amd import workloads
from helpers import check_window_records


def local_random(n, per_row, spread, seed, far=0):
    rng = np.random.default_rng(seed)
    counts = rng.integers(0, 2 * per_row + 1, n)
    rows = np.repeat(np.arange(n), counts)
    cols = np.clip(rows + rng.integers(-spread, spread + 1, rows.size), 0, n - 1)
    if far:
        fr = np.repeat(np.arange(n), far)
        rows = np.concatenate([rows, fr])
        cols = np.concatenate([cols, rng.integers(0, n, fr.size)])
    A = sp.coo_matrix((rng.uniform(-1, 1, rows.size), (rows, cols)), shape=(n, n)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A


def table(A, row_begin=0):
    A = A.tocsr()
    A.sort_indices()
    T = sa.windows_host(A, row_begin)
    return T, check_window_records(T, A, row_begin)


def test_banded_and_jittered_patterns_are_fully_covered():
    n = 200_000
    offs = (1, 2, 3, 1000, 1001, 20000, 20001)
    band = sp.diags([np.ones(n - o) for o in offs] * 2 + [np.ones(n)], list(offs) + [-o for o in offs] + [0], format="csr")
    T, covered = table(band)
    assert covered == band.nnz and not np.any(T[:, 0] >> 8)
    inner = T[100:-100]
    assert np.all((inner[:, 0] & 255) == 5) and np.all(inner[:, 1] <= 5 * 288)  # 5 clusters of (256 + span) columns, 128-byte aligned
    J = workloads.jitter_band(n, offsets=offs)
    T, covered = table(J)
    # (at the matrix's ends a cluster may hold fewer than 8 entries of a block: those few are gathered)
    assert J.nnz - 64 <= covered <= J.nnz and np.all((T[:, 0] & 255) <= 8) and T[:, 1].max() <= 6 * (256 + 2 * 64 + 16 + 32)


def test_ragged_rows_empty_blocks_and_odd_sizes():
    for n in (1, 2, 255, 256, 257, 4097, 70001):
        A = local_random(n, 7, 300, n)
        T, covered = table(A)
        assert covered <= A.nnz and (n < 255 or covered >= A.nnz - 16)  # (a block with a handful of entries gathers them)
    E = sp.csr_matrix((1000, 1000))
    T, covered = table(E)
    assert covered == 0 and np.all(T[:, 0] == 1 << 8) and np.all(T[:, 1] == 0)  # no entries: no windows, the (empty) gather path


def test_far_entries_thin_runs_and_the_lds_budget():
    n = 600_000
    A = local_random(n, 6, 500, 21, far=1)  # one entry per row anywhere: isolated lines, left to the gather
    T, covered = table(A)
    assert 0.8 * A.nnz < covered < A.nnz and np.all(T[:, 0] >> 8 == 1) and np.all((T[:, 0] & 255) >= 1)
    R = workloads.m_rand(n, seed=3)  # all far: nothing is worth a window
    T, covered = table(R)
    assert covered < 0.1 * R.nnz
    # 256 rows whose columns cover 40000 contiguous columns: over the 48 KiB budget -> that block gathers
    m = 100_000
    rng = np.random.default_rng(9)
    wide_r = np.repeat(np.arange(5120, 5376), 160)
    wide_c = (np.tile(np.arange(160) * 250, 256) + rng.integers(0, 250, wide_r.size)) + 5000
    band = sp.diags([rng.uniform(-1, 1, m - abs(k)) for k in (-3, -1, 0, 1, 3)], [-3, -1, 0, 1, 3], format="csr")
    W = (band + sp.coo_matrix((rng.uniform(-1, 1, wide_r.size), (wide_r, wide_c)), shape=(m, m))).tocsr()
    W.sum_duplicates()
    T, covered = table(W)
    assert (T[20, 0] & 255) == 0 and T[20, 0] >> 8 == 1 and np.all((np.delete(T, 20, axis=0)[:, 0] & 255) >= 1)
    # 40 clusters of 256 columns each in one block: at most 8 windows survive, the lightest runs are gathered
    k = 200_000
    rows = np.concatenate([np.repeat(np.arange(1024, 1280), 40), np.arange(k)])
    cols = np.concatenate([np.tile(np.arange(40) * 1000, 256) + np.repeat(np.arange(1024, 1280), 40), np.arange(k)])
    C = sp.coo_matrix((np.ones(rows.size), (rows, cols)), shape=(k, k)).tocsr()
    C.sum_duplicates()
    T, covered = table(C)
    assert (T[4, 0] & 255) == 8 and T[4, 0] >> 8 == 1 and T[4, 2] >= 8 * 256


def test_row_shards_use_global_columns():
    # a shard holds rows [b, e) with GLOBAL column indices: the windows follow row_begin + local row
    n = 300_000
    A = workloads.jitter_band(n, offsets=(1, 2, 3, 1000, 1001, 20000, 20001))
    whole, _ = table(A)
    for world, rank in ((2, 1), (3, 1), (8, 5)):
        b, e = sa.shard_range(n, world, rank)
        part = A[b:e]
        T, covered = table(part, b)
        assert part.nnz - 64 <= covered <= part.nnz
        if b % 256 == 0:  # block boundaries coincide: the shard's records are the whole matrix's
            assert np.array_equal(T[: (e - b) // 256], whole[b // 256: b // 256 + (e - b) // 256])
