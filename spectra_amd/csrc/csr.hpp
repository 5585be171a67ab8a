// Device-resident CSR row shard (the A behind SparseSymMatProd / SparseGenMatProd) and the SpMV launcher.
#pragma once
#include <vector>

#include "common.hpp"
#include "tiles.hpp"
#include "staged.hpp"

// x windows of the diagonal format: the offsets of a matrix cluster (here: ±1..3, ±1000/1001, ±100000/100001), and a
// 256-row block needs, per cluster, one contiguous piece of x of 256 + span entries.  Staged through LDS once per block.
struct mispec_dia_windows
{
    int nc = 0;        // clusters (0: the windowed kernel is not used)
    int total = 0;     // doubles of LDS
    int start[8];      // first offset of the cluster
    int len[8];        // 256 + span
    int base[8];       // position of the window in LDS
    int idx[32];       // per diagonal: base[c] + (offset - start[c]); thread t reads xs[idx[k] + t]
};

struct mispec_csr
{
    mispec_ctx* ctx = nullptr;
    int64_t n_rows = 0;     // global rows
    int64_t n_cols = 0;     // global columns (length of x)
    int64_t row_begin = 0;  // this shard holds rows [row_begin, row_end)
    int64_t row_end = 0;
    int64_t nnz = 0;        // local non-zeros
    mispec::DevBuf<int32_t> rowptr;  // local_rows + 1, offsets into colind/val (start at 0)
    mispec::DevBuf<int32_t> colind;  // nnz rounded up to 4, +8 slack, padded with column 0 / value 0
    mispec::DevBuf<double> val;
    // Offset-coded column indices (optional second index format, used by the SpMV when present): matrices whose
    // entries lie on at most 256 distinct diagonals (banded / stencil matrices) keep one byte per entry,
    // col = global_row + dict[code]; 9 instead of 12 bytes of HBM traffic per stored entry.  ndict == 0: absent.
    mispec::DevBuf<uint8_t> codes;   // same padding as colind, padded with code 0
    mispec::DevBuf<int32_t> dict;    // ndict distinct (col - global_row) values
    int ndict = 0;
    bool use_codes = true;           // mispec_csr_use_offset_codes: per-matrix switch (tests compare the two kernels)
    // Diagonal storage (third format, built from the codes when the dictionary is small and the diagonals are well
    // filled): dia[k * dia_ld + r] = A(r, r + dia_off[k]), diagonals sorted by offset, absent entries zero.  The SpMV
    // then needs no index at all and no gather: 8 bytes per stored slot, every load coalesced.
    mispec::DevBuf<double> dia;
    mispec::DevBuf<int32_t> dia_off;
    int64_t dia_ld = 0;
    int ndia = 0;
    mispec_dia_windows dia_win;
    // x windows of the int32 CSR kernel (csr.hip k_spmv_csr_win): per 256-row block a 32-int record naming at most 8 contiguous
    // ranges of x that hold the block's columns; built on the device at ingest, used by format 0 when most entries are covered.
    mispec::DevBuf<int32_t> wtab;
    int win_lds_doubles = 0;         // largest window total over the blocks (doubles of LDS the launch reserves)
    int64_t win_covered = 0;         // entries whose x comes from a window
    int64_t win_blocks = 0;          // blocks that have windows
    int use_windows = -1;            // mispec_csr_use_windows: -1 automatic, 0 / 1 per-matrix switch (tests compare the two kernels)
    // automatic: rows of fewer than 9 entries on average keep the gather kernel — a block then holds too few entries to pay for
    // its windows (7-point stencil after RCM, in the solver loop: 0.198-0.203 ms with windows, 0.194 ms without, profiles/r09o, r09z)
    bool windows_active() const
    {
        if (wtab.p == nullptr || win_lds_doubles <= 0 || use_windows == 0)
            return false;
        return use_windows == 1 || double(nnz) >= 9.0 * double(local_rows());
    }
    int forced_format = -1;          // mispec_csr_set_spmv_format: -1 automatic, 0 int32 indices, 1 offset codes, 2 diagonals, 3 tiles, 4 staged
    // Column-blocked tiles (fourth format, tiles.hip): built at ingest for matrices whose gathers are scattered over the
    // whole of x and that reordering does not localise; bit-identical products again.
    mispec::DevTiles tiles;
    // Staged format (fifth, staged.hip): the product in two streaming phases with x and y in LDS; built at ingest next to (and
    // preferred over) the tiles when MISPEC_SPMV_STAGED allows; bit-identical products again.
    mispec::DevStaged staged;
    // Symmetric reordering (reorder.hip): when perm is set, the arrays above hold B = P A P', B(i, j) = A(perm[i], perm[j]).
    // The public products (mispec_spmv*, operator(), downloads) keep the ORIGINAL index order (gather x, product, scatter
    // y); the eigensolvers work in the permuted order and un-permute what they return.  Unsharded square matrices only.
    mispec::DevBuf<int32_t> perm;          // new -> old, local_rows entries
    std::vector<int32_t> perm_host, inv_host;  // and old -> new
    int reorder_method = 0;                // 0 none, 1 reverse Cuthill-McKee
    double far_before = 0.0, far_after = 0.0;  // fraction of entries further than kFarWindow from the diagonal
    bool structurally_symmetric = false;   // set by the symmetric ingest paths (the ordering then skips A + A')
    mutable mispec::DevBuf<double> perm_x, perm_y;  // scratch of the order-preserving public product
    bool reordered() const { return perm.p != nullptr; }
    // staging for the host-pointer paths (allocated on first use)
    mutable mispec::DevBuf<double> stage_x, stage_y;

    int64_t local_rows() const { return row_end - row_begin; }
    // 12*nnz + 4*(rows+1) + 8*cols + 8*rows  
    double algorithmic_bytes() const
    {
        return 12.0 * double(nnz) + 4.0 * double(local_rows() + 1) + 8.0 * double(n_cols) + 8.0 * double(local_rows());
    }
    // what the SpMV actually has to move with the index format in use (compulsory traffic, x counted once)
    // 0: int32 column indices, 1: offset codes, 2: diagonals — what launch_spmv will use for this matrix
    int spmv_format() const;
    double stored_bytes_for(int format) const
    {
        if (format == 2)
            return 8.0 * double(ndia) * double(local_rows()) + 8.0 * double(n_cols) + 8.0 * double(local_rows());
        if (format == 4)  // both phases: values, indices, the product array's round trip, tables; x counted once like everywhere
            return staged.stored_bytes(local_rows(), n_cols);
        if (format == 3)  // 12 bytes per stored entry (padding included) + the chunk table; x counted once like everywhere
            return 12.0 * double(tiles.entries) + 8.0 * double(tiles.nchunks) + 12.0 * double(tiles.nseg) + 8.0 * double(n_cols) +
                   8.0 * double(local_rows());
        const double per_entry = format == 1 ? 9.0 : 12.0;
        return per_entry * double(nnz) + 4.0 * double(local_rows() + 1) + 8.0 * double(n_cols) + 8.0 * double(local_rows());
    }
    double stored_bytes() const
    {
        return stored_bytes_for(spmv_format());
    }
};

namespace mispec {

// Optional epilogue fused into the SpMV of a Lanczos step (Lanczos.h:131-142):
//   w = A v - h_prev * v_prev ;  alpha_partials[block] = sum_rows v[row] * w[row]
struct SpmvEpilogue
{
    const double* v_rows = nullptr;   // local rows of the input vector (the new basis column)
    const double* v_prev = nullptr;   // previous basis column, or nullptr when restarting (Lanczos.h:138)
    double h_prev = 0.0;              // H(i,i-1); used only when v_prev != nullptr
    const double* h_prev_dev = nullptr;  // if set, H(i,i-1) is read from device memory instead (device-driven steps)
    const int* status = nullptr;      // if set, the launch is a no-op unless *status == 0
    double* partials = nullptr;       // one double per block, deterministic second stage elsewhere
    int first_block = 0;              // set by the launcher: first 256-row block this launch covers
    // One-sweep Lanczos steps on diagonal storage (spmv_can_post_scale): x_dev is the un-normalised residual f, v_rows = f; the
    // kernel divides the row sums and v by beta = StepState::beta, uses beta as H(i,i-1), records it and takes the
    // beta < sqrt(eps) stop of k_scale_step (csr.hip k_spmv_dia_win<.., POST>).
    void* post_scale_state = nullptr;  // StepState* (krylov.hpp)
    int post_scale_step = 0;
    double post_scale_eps_sqrt = 0.0;
};
// true when launch_spmv_raw(A, ...) honours SpmvEpilogue::post_scale_state (else the caller scales with k_scale_step)
bool spmv_can_post_scale(const ::mispec_csr& A);
// Rows per SpMV workgroup (one thread per row in the reduction phase)
int spmv_rows_per_block();
inline int spmv_num_blocks(int64_t local_rows)
{
    const int r = spmv_rows_per_block();
    return int((local_rows + r - 1) / r);
}

// ev_start / ev_stop (both or none): HIP events bound to this dispatch's start and completion.
// launch_spmv keeps the caller's index order: for a reordered matrix it gathers x, runs the product on P A P', scatters y
// and applies the epilogue as a separate pass.  launch_spmv_raw is the product with the STORED matrix (the permuted one
// when reordered): what a solver that works in the permuted order calls.
void launch_spmv(const mispec_csr& A, const double* x_dev, double* y_dev, const SpmvEpilogue* epi, hipEvent_t ev_start = nullptr,
                 hipEvent_t ev_stop = nullptr);
// block_count >= 0: only the 256-row blocks [block_first, block_first + block_count) (their rows of y, their partial records)
void launch_spmv_raw(const mispec_csr& A, const double* x_dev, double* y_dev, const SpmvEpilogue* epi, hipEvent_t ev_start = nullptr,
                     hipEvent_t ev_stop = nullptr, int block_first = 0, int block_count = -1);
// For a row shard: the longest run of 256-row blocks whose entries reference only columns in [col_lo, col_hi) (the shard's own
// slice of x): rows that can be multiplied before the other ranks' parts of x have arrived.  Synchronises the stream.
void interior_blocks(const mispec_csr& A, int64_t col_lo, int64_t col_hi, int& first, int& count);
// dst[i] = src[perm[i]] (to the stored order) / dst[perm[i]] = src[i] (back to the caller's order), i < local_rows
void launch_to_stored_order(const mispec_csr& A, const double* src, double* dst);
void launch_from_stored_order(const mispec_csr& A, const double* src, double* dst);
constexpr int64_t kFarWindow = 131072;  // |col - row| beyond which an x gather is counted as "far" (1 MiB of x)

// For every rank p of a `world`-way row partition with `block` rows per rank: the smallest (lo[p]) and largest
// (hi[p]) column index this shard references inside p's rows; hi[p] = -1 when it references none.  Returns
// false (nothing computed) when world exceeds the kernel's peer limit.  Synchronises the stream.
bool column_ranges(const mispec_csr& A, int64_t block, int world, std::vector<int64_t>& lo, std::vector<int64_t>& hi);

}  // namespace mispec
