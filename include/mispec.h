/* =============================================================================
 *  mispec.h — C ABI of the MI355X-native implicitly-restarted Lanczos/Arnoldi
 *  hot path (libmispec.so, built from spectra_amd/csrc/ for gfx950).
 *
 *  The reference (yixuan/spectra v1.2.0) is header-only C++ and defines NO FFI:
 *  its plugin boundary is the duck-typed template concept
 *      OpType::Scalar, rows(), cols(), perform_op(const Scalar*, Scalar*)
 *  (include/Spectra/SymEigsSolver.h:43-51, MIGRATION.md:11-39).  This header is
 *  the "thin C-ABI shim" BASELINE.json's north_star asks for: plain pointers and
 *  sizes, no C++/torch types.  Each entry point names the reference routine it
 *  replaces (paths relative to /root/reference/include/Spectra/).  The
 *  Spectra-compatible C++ templates in include/Spectra/ are written on top of
 *  exactly these functions; INTEGRATION.md shows the binding a Spectra
 *  maintainer would add.
 *
 *  Conventions
 *    - every function returns int: 0 = ok, <0 = error class (below); the
 *      message is in mispec_last_error() (thread-local).
 *    - dense matrices are column-major (like Eigen's default), fp64; sparse
 *      indices are int32 (StorageIndex=int, SparseSymMatProd.h:30).
 *    - "_host" pointers are host memory owned by the caller; "_dev" pointers are
 *      device memory on the context's device.  Handles own everything behind them.
 *    - a context is bound to one device and one HIP stream; handles are not
 *      thread-safe (the reference's solver objects are not either).
 * ============================================================================= */
#ifndef MISPEC_H
#define MISPEC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MISPEC_OK 0
#define MISPEC_EINVAL (-1)   /* bad argument            -> std::invalid_argument (HermEigsBase.h:267-271, Arnoldi.h:148,209) */
#define MISPEC_ELOGIC (-2)   /* call-order violation    -> std::logic_error      (UpperHessenbergQR.h:207) */
#define MISPEC_ERUNTIME (-3) /* HIP/RCCL failure, failed small decomposition -> std::runtime_error (TridiagEigen.h:204) */

typedef struct mispec_ctx mispec_ctx; /* device + stream (+ communicator when row-sharded) */
typedef struct mispec_csr mispec_csr; /* device-resident CSR row shard: the A behind Sparse{Sym,Gen}MatProd */
typedef struct mispec_fac mispec_fac; /* Lanczos/Arnoldi factorisation A V = V H + f e' (LinAlg/Arnoldi.h:32-62) */

const char* mispec_last_error(void);
/* "x.y.z (gfx950)" */
const char* mispec_version(void);

/* Tuning switches and test hooks, by name (value NULL: back to the default).  The library reads no other global state; a name that
 * has not been set falls back to the environment variable MISPEC_<NAME IN UPPER CASE> — the test-only override the A/B tools
 * and the parity tests use.  Unknown names: MISPEC_EINVAL.  Names and values:
 *   orth            onesweep (default) | onesweep-eager | reference      control flow of the Lanczos steps at creation
 *   one_reduction   1 (default) | 0                                      one reduction per one-sweep step
 *   orth_kernel     dma (default, >= 131072 rows) | dma2 | dmac | dmap | reg   the one-sweep pass: LDS-DMA ring or registers
 *   host_turn       fast (default) | copy                                restart's host turn: pinned-memory kernels or hipMemcpy
 *   small           host (default) | host-serial | device                where the ncv x ncv work of a restart runs (host: the shifted
 *                                                                        QR sweeps as a skewed pipeline; host-serial: in the reference's order, same bits)
 *   restart_sync    0 (default) | 1                                      synchronising fused restart
 *   host_steps      0 (default) | 1                                      host-synchronous steps
 *   spec_corr       corrections enqueued speculatively per step (reference flow)
 *   overlap, exchange                                                    sharded product: 0 switches the overlap / the neighbour exchange off
 *   csr_win, csr_win_iters, csr_win_pf, csr_win_nt, dia2, spmv_tiles, spmv_staged, reorder, kernel_probe   SpMV format / kernel choice
 *   host_threads    upper bound on the host threads of the ingest / the shift solve's host-side factorisation (tests: results do not depend on it)
 *   vq              mfma: the f64-MFMA variant of V*Q
 *   shift           banded shift solve: key=value list — kernel variants that must agree (tests) and profile=1 (set_shift's phases on stderr)
 * The reference has no counterpart (its only switches are template parameters). */
int mispec_set_option(const char* name, const char* value);
const char* mispec_get_option(const char* name);

/* ---------------------------------------------------------------------------
 * Context
 * ------------------------------------------------------------------------- */
/* hip_stream: a hipStream_t to launch on (e.g. torch.cuda.current_stream().cuda_stream), or NULL to create one. */
int mispec_ctx_create(int device, void* hip_stream, mispec_ctx** out);
int mispec_ctx_destroy(mispec_ctx* ctx);
int mispec_ctx_sync(mispec_ctx* ctx);
void* mispec_ctx_stream(mispec_ctx* ctx);

/* Row sharding (the reference has no notion of it).  rank/world and two collectives over
 * the ranks: an all-gather of equal-sized blocks of doubles (the Krylov vector before each SpMV) and an
 * in-place sum all-reduce of a few doubles (alpha, |f|^2, V'f).
 * Contract of the callbacks: (i) the collective must be ORDERED ON `hip_stream` — the solver's stream, or a second stream of the
 * library while the exchange overlaps the product of the local rows (mispec_fac_overlap_info): a transport that runs on a
 * stream of its own has to wait for / signal that stream itself; (ii) the all-gather is called IN PLACE: send_dev is
 * recv_dev + rank * count_per_rank; (iii) return 0 on success, anything else aborts the solve with MISPEC_ERUNTIME. */
typedef struct mispec_comm
{
    int rank, world;
    int (*allgather)(void* user, const double* send_dev, double* recv_dev, int64_t count_per_rank, void* hip_stream);
    int (*allreduce_sum)(void* user, double* buf_dev, int64_t count, void* hip_stream);
    void* user;
    /* Optional (NULL = not provided): personalised exchange of sub-ranges.  Send send_count[p] doubles starting
     * at send_dev + send_off[p] to every peer p and receive recv_count[p] doubles from p into
     * recv_dev + recv_off[p]; the entries for p == rank are ignored; the offset/count arrays (host memory,
     * `world` entries) stay valid for the lifetime of the matrix.  When present, a matrix whose rows reference
     * only a small part of the other ranks' rows (banded / stencil matrices: neighbour halos) uses this instead
     * of the full all-gather; MISPEC_EXCHANGE=allgather forces the all-gather. */
    int (*exchange)(void* user, const double* send_dev, const int64_t* send_off, const int64_t* send_count,
                    double* recv_dev, const int64_t* recv_off, const int64_t* recv_count, void* hip_stream);
} mispec_comm;
int mispec_ctx_set_comm(mispec_ctx* ctx, const mispec_comm* comm);
/* Built-in communicator over RCCL (librccl.so.1 is dlopen'ed on first use).  unique_id is the 128-byte
 * ncclUniqueId produced by mispec_rccl_unique_id() on rank 0 and broadcast by the launcher. */
int mispec_rccl_unique_id(char out[128]);
int mispec_ctx_set_comm_rccl(mispec_ctx* ctx, int rank, int world, const char unique_id[128]);
/* Built-in in-process communicator for `world` host threads sharing one device (tests of the sharded
 * path on a 1-GPU box).  Returns a group handle; each thread calls _attach with its own ctx and rank. */
typedef struct mispec_loopback mispec_loopback;
int mispec_loopback_create(int world, mispec_loopback** out);
int mispec_loopback_attach(mispec_loopback* grp, mispec_ctx* ctx, int rank);
int mispec_loopback_destroy(mispec_loopback* grp);
/* Row range [begin,end) owned by `rank` when n rows are split over `world` ranks in equal blocks of
 * mispec_shard_block(n, world) rows (the last ranks may be short or empty). Pure host arithmetic. */
int64_t mispec_shard_block(int64_t n, int world);
int mispec_shard_range(int64_t n, int world, int rank, int64_t* begin, int64_t* end);

/* ---------------------------------------------------------------------------
 * Sparse matrix ingest — replaces the Eigen::SparseMatrix held by
 * SparseSymMatProd / SparseGenMatProd (MatOp/SparseSymMatProd.h:46-60).
 * The matrix is COPIED to HBM once (the reference keeps an Eigen::Ref).
 * With a sharded context only rows [begin,end) of mispec_shard_range are kept.
 * ------------------------------------------------------------------------- */
/* General CSR (SparseGenMatProd<double, RowMajor>).  rowptr[n_rows+1], colind/val[nnz], host memory. */
int mispec_csr_upload(mispec_ctx* ctx, int64_t n_rows, int64_t n_cols, const int32_t* rowptr_host,
                      const int32_t* colind_host, const double* val_host, mispec_csr** out);
/* General CSC (SparseGenMatProd<double, ColMajor>, the reference default): transposed to CSR on ingest. */
int mispec_csr_from_csc(mispec_ctx* ctx, int64_t n_rows, int64_t n_cols, const int32_t* colptr_host,
                        const int32_t* rowind_host, const double* val_host, mispec_csr** out);
/* Symmetric operator from ONE triangle of a compressed matrix: SparseSymMatProd<double, Uplo, Flags>
 * (MatOp/SparseSymMatProd.h:83-88, selfadjointView<Uplo>).  Entries in the other triangle are ignored
 * (test/SymEigs.cpp:27-28 relies on that); the triangle is mirrored into a full CSR on ingest.
 * uplo: 'L' or 'U'.  row_major: 0 = CSC input (Eigen::ColMajor), 1 = CSR input (Eigen::RowMajor). */
int mispec_csr_from_triangle(mispec_ctx* ctx, int64_t n, const int32_t* outer_host, const int32_t* inner_host,
                             const double* val_host, char uplo, int row_major, mispec_csr** out);
/* Synthetic benchmark matrices generated directly in HBM (SURVEY.md §8d "M-band"): row i holds columns
 * i+off for off in {0} U {+-offsets[k]} inside [0,n); value = counter hash of (seed, min(i,j), max(i,j))
 * (symmetric) or (seed, i, j) (non-symmetric) mapped to U(-0.5,0.5).  Bit-identical to oracle/synth_matrix.h. */
int mispec_csr_synth_band(mispec_ctx* ctx, int64_t n, uint64_t seed, const int64_t* offsets, int noff, int symmetric,
                          mispec_csr** out);
int mispec_csr_destroy(mispec_csr* A);
int64_t mispec_csr_rows(const mispec_csr* A);       /* global row count (rows() of the op) */
int64_t mispec_csr_cols(const mispec_csr* A);       /* global column count */
int64_t mispec_csr_local_rows(const mispec_csr* A); /* rows held by this shard */
int64_t mispec_csr_local_nnz(const mispec_csr* A);
/* Index format the SpMV uses for this shard: 0 = plain int32 column indices (12 bytes per stored entry), d > 0 =
 * offset codes, one byte per entry into a dictionary of d <= 256 distinct diagonals col - row (9 bytes per entry).
 * Built at construction whenever the dictionary fits; mispec_csr_set_spmv_format / mispec_csr_use_offset_codes select. */
int mispec_csr_offset_codes(const mispec_csr* A);
/* Per-matrix switch between the two index formats (both give bit-identical products); no effect when the matrix
 * has no dictionary. */
int mispec_csr_use_offset_codes(mispec_csr* A, int enable);
/* x windows of the int32 CSR kernel (format 0): at ingest every 256-row block gets the (at most 8) contiguous ranges of x that
 * hold its columns; when at least 75 % of the entries are covered the kernel stages those ranges through LDS with coalesced
 * loads instead of gathering x entry by entry (entries outside the windows keep the gather).  The matrix arrays stay the plain
 * int32 CSR; products and summation order are unchanged (bit-identical).  mispec_csr_use_windows switches the kernel per
 * matrix: 1 windows, 0 gathers, -1 automatic (the default: windows when the table was adopted and the rows hold at least 9
 * entries on average — shorter rows do not pay for them); mispec_csr_windows_info reports blocks with windows, entries served
 * from LDS and the LDS doubles reserved (0: not adopted).  Replaces the x access of MatOp/SparseSymMatProd.h:83-88 / SparseGenMatProd.h:72-77. */
int mispec_csr_use_windows(mispec_csr* A, int enable);
int mispec_csr_windows_info(const mispec_csr* A, int64_t* blocks, int64_t* covered_entries, int64_t* lds_doubles);
int mispec_csr_windows_in_use(const mispec_csr* A); /* 1: format 0 of this matrix runs k_spmv_csr_win (switch and automatic rule applied) */
/* The table itself, for tests: 32 ints per 256-row block of this shard — [0] windows | far flag << 8, [1] doubles of LDS, [2] entries
 * served from LDS, [4..11] first column of each window (0x3fffffff: unused), [12..19] LDS position minus first column, [20..27] one
 * past the last column. */
int mispec_csr_windows_table(const mispec_csr* A, int32_t* records_out, int64_t capacity);
/* Host-only test hook (no device): the table of local rows [0, n_rows) of a shard that starts at global row row_begin, from host
 * CSR arrays (rowptr starting at the shard's first entry) — the selection code the device builder runs, on the CPU.  records_out:
 * 32 ints per 256-row block. */
int mispec_csr_windows_host(int64_t n_rows, int64_t n_cols, int64_t row_begin, const int32_t* rowptr, const int32_t* colind,
                            int32_t* records_out);
/* Storage format the SpMV uses for this shard: 0 = CSR with int32 column indices, 1 = CSR with offset codes, 2 = diagonal
 * storage (values kept diagonal-major, no index and no gather; chosen when the dictionary has <= 32 diagonals that are
 * at least 3/4 full, rows sorted, no duplicate entries), 3 = column-blocked tiles (built at
 * ingest for unsharded matrices with more than a quarter of their entries further than 131072 columns from the diagonal
 * that reordering did not localise: 8192-row segments x 65536-column blocks, the segment's sums in LDS, the gathers of the
 * workgroups resident on an XCD sharing its L2; MISPEC_SPMV_TILES=0 turns the format off, =1 builds it for any matrix).  All give bit-identical products for finite x
 * (the diagonal format multiplies x by explicit zeros where the matrix has no entry).
 * mispec_csr_set_spmv_format forces a format for this matrix (-1 = automatic; a format that was not built falls back). */
int mispec_csr_spmv_format(const mispec_csr* A);
int mispec_csr_set_spmv_format(mispec_csr* A, int format);
/* Bytes one SpMV with this shard has to move, x counted once.  stored = 0: the CSR/int32 figure
 * 12 nnz + 4 (rows+1) + 8 cols + 8 rows that roofline numbers are quoted on; stored != 0: with the index format in use. */
double mispec_csr_spmv_bytes(const mispec_csr* A, int stored);
/* A(i,j) of the stored (mirrored) matrix; 0 when absent.  Replaces operator()(i,j) (SparseSymMatProd.h:101-104).
 * Only rows of this shard can be queried. */
int mispec_csr_coeff(const mispec_csr* A, int64_t i, int64_t j, double* out);
/* Download this shard as host CSR (rowptr[local_rows+1], colind/val[local_nnz]; any pointer may be NULL). */
int mispec_csr_download(const mispec_csr* A, int32_t* rowptr_host, int32_t* colind_host, double* val_host);

/* ---------------------------------------------------------------------------
 * y = A x — replaces perform_op (MatOp/SparseSymMatProd.h:83-88, SparseGenMatProd.h:82-87).
 * ------------------------------------------------------------------------- */
/* Device path: x_dev has cols() doubles (the FULL vector, also when row-sharded), y_dev local_rows(). */
int mispec_spmv(const mispec_csr* A, const double* x_dev, double* y_dev);
/* Literal perform_op contract: host pointers, staged through HBM (H2D + kernel + D2H).  Unsharded only. */
int mispec_spmv_host(const mispec_csr* A, const double* x_host, double* y_host);
/* Y = A X for a column-major n x k block: operator* (SparseSymMatProd.h:93-96). Host pointers. */
int mispec_spmm_host(const mispec_csr* A, const double* X_host, int64_t ldx, int k, double* Y_host, int64_t ldy);
/* Symmetric reordering of an unsharded square matrix (no reference counterpart: on a CPU the ordering of the rows costs
 * little; on the GPU scattered x gathers cost an order of magnitude).  The stored matrix becomes P A P' with P from reverse
 * Cuthill-McKee on the pattern of A + A' (host, once).  Everything the C ABI hands out keeps the CALLER's index order:
 * mispec_spmv / _host / mispec_csr_coeff / _download un-permute, the eigensolvers work in the permuted order and return
 * eigenvectors in the caller's order; eigenvalues are those of A.  method 1: always reorder; -1: only if more than a
 * quarter of the entries lie further than 131072 columns from the diagonal and the ordering at least halves that
 * fraction (this is also what ingest does by itself unless MISPEC_REORDER=none; MISPEC_REORDER=rcm forces it).
 * *applied = 1 when the matrix is reordered on return.  mispec_csr_reordering returns 0 (none) or 1 (RCM) and the
 * fraction of far entries before / after; mispec_csr_permutation writes perm[new] = old (identity when not reordered). */
int mispec_csr_reorder(mispec_csr* A, int method, int* applied);
int mispec_csr_reordering(const mispec_csr* A, double* far_before, double* far_after);
int mispec_csr_permutation(const mispec_csr* A, int32_t* perm_out);
/* What the tile format of this matrix looks like (segments = 0: not built): stored entries incl. padding, padding entries,
 * chunks (runs of <= 1024 entries of one tile, the unit between two barriers of the kernel). */
int mispec_csr_tiles_info(const mispec_csr* A, int64_t* segments, int64_t* entries, int64_t* padding, int64_t* chunks);
/* What the staged format of this matrix looks like (format 4: the product in two streaming kernels with x and y in LDS, built
 * for the same patterns as the tiles): row bins of 8192 rows (0: not built), phase-1 slots (stored entries + the padding of the
 * 8192-column blocks to even counts), batches (<= 1024 entries of one bin, the unit of phase 2) and the chunks they are made of
 * (runs that are contiguous in phase-1 order). */
int mispec_csr_staged_info(const mispec_csr* A, int64_t* bins, int64_t* slots, int64_t* batches, int64_t* chunks);
/* Wall-clock seconds of the host stages of the last mispec_csr_upload / mispec_csr_from_triangle on the calling thread:
 * [0] the whole call, [1] triangle -> full matrix, [2] validation + local row pointers, [3] index formats (offset codes, diagonal
 * storage) incl. the H2D copies of the CSR arrays, [4] far-gather statistics + reordering, [5] tile image on the host, [6] its
 * upload and split, [8] staged image on the host, [9] its upload.  count <= 10 values are written. */
int mispec_last_ingest_info(double* seconds_out, int count);
int mispec_ingest_threads(void); /* host threads the ingest stages use: the machine's hardware threads, at most 64 */
/* Host-only test hook (no device needed): the full symmetric CSR matrix (rows sorted by column) that mispec_csr_from_triangle
 * derives from one stored triangle — built by the library's host threads, the same bytes whatever their number.  rowptr_out has
 * n + 1 entries, colind_out / val_out `capacity` entries (twice the input's entries always suffice); *nnz_out = entries written. */
int mispec_mirror_triangle_host(int64_t n, const int32_t* outer, const int32_t* inner, const double* val, char uplo, int row_major,
                                int32_t* rowptr_out, int32_t* colind_out, double* val_out, int64_t capacity, int64_t* nnz_out);
/* Host image of the tile format and its summation order, for tests (no device needed): y = A x through the tiles of an
 * nrows x ncols CSR matrix; *built = 0 when the format does not apply (unsorted rows or duplicate entries, more than 65535
 * column blocks; a row with more than 7 entries inside one column block is emitted in several passes).  stats (optional): entries incl. padding, padding entries, chunks. */
int mispec_tiles_spmv_host(int64_t nrows, int64_t ncols, const int32_t* rowptr, const int32_t* colind, const double* val,
                           const double* x, double* y, int* built, int64_t* stats);
/* The same for the staged format (format 4): y = A x through its host image in the order of the two kernels; *built = 0 when the
 * format does not apply (unsorted rows, 2^32 stored entries or more, a (column block, row bin) table beyond 2^26 entries), 2 when
 * the image is correct but heavy rows leave its batches nearly empty — the automatic format choice at ingest then keeps the
 * tiles / CSR kernels.  stats (optional, 5 values): row bins, phase-1 slots, batches, chunks, the largest number of rank rounds
 * a batch needs. */
int mispec_staged_spmv_host(int64_t nrows, int64_t ncols, const int32_t* rowptr, const int32_t* colind, const double* val,
                            const double* x, double* y, int* built, int64_t* stats);
/* The ordering alone, on host arrays (no device needed): perm_out[new] = old for the pattern of an n x n CSR matrix;
 * *gave_up = 1 (identity returned) when the first breadth-first level structure is too wide for any ordering to help. */
int mispec_rcm_order(int64_t n, const int32_t* rowptr, const int32_t* colind, int symmetric_pattern, int32_t* perm_out,
                     int* gave_up, int64_t* widest_level);
/* Duration (ms, HIP events on the context stream) of the last `reps` back-to-back SpMV launches. */
int mispec_spmv_time(const mispec_csr* A, const double* x_dev, double* y_dev, int reps, float* ms_per_launch);

/* ---------------------------------------------------------------------------
 * Shift-and-invert operator  y = (A - sigma I)^{-1} x  for symmetric A — replaces SparseSymShiftSolve
 * (MatOp/SparseSymShiftSolve.h:36-111, which delegates to Eigen::SparseLU).  The factorisation is redone
 * by set_shift (once per shift); every solve runs on the device: a recursive partitioned banded LDL' when the
 * half-bandwidth is <= 8 (any n; top level factored on the device) or <= 64 with n > 4096 (longer chunks, levels
 * factored on the host; sigma may lie inside the spectrum: tiny pivots are boosted and set_shift
 * calibrates the iterative-refinement steps each solve then performs, mispec_symshift_refinement_info), a dense
 * LU inverse + GEMV when n <= 4096 (the reference's own test fixtures).  A matrix beyond the dense limit whose band is too
 * wide AS IT COMES is reordered at construction (reverse Cuthill-McKee on the pattern, as the reference's sparse factorisations
 * order theirs) and takes the banded path when the reordered half-bandwidth is <= 64 — a banded matrix in a scattering row
 * order, path- / ladder-like graphs; solves and eigenvectors keep the caller's index order (mispec_symshift_bandwidth reports both
 * widths; MISPEC_REORDER=none opts out).  Other sparsity patterns are rejected by set_shift (MISPEC_EINVAL).
 * Input: one triangle of a compressed matrix, as for mispec_csr_from_triangle.
 * ------------------------------------------------------------------------- */
typedef struct mispec_symshift mispec_symshift;
int mispec_symshift_create(mispec_ctx* ctx, int64_t n, const int32_t* outer_host, const int32_t* inner_host,
                           const double* val_host, char uplo, int row_major, mispec_symshift** out);
/* Pencil form — SymShiftInvert<double, Eigen::Sparse, Eigen::Sparse> (MatOp/SymShiftInvert.h:140-208): the operator
 * (A - sigma B)^{-1} for two sparse symmetric matrices given by one triangle each; same restrictions as above on
 * the pattern of A - sigma B (banded with half-bandwidth <= 64, or n <= 4096). */
int mispec_symshift_create_pencil(mispec_ctx* ctx, int64_t n, const int32_t* a_outer, const int32_t* a_inner,
                                  const double* a_val, char a_uplo, int a_row_major, const int32_t* b_outer,
                                  const int32_t* b_inner, const double* b_val, char b_uplo, int b_row_major,
                                  mispec_symshift** out);
/* General (non-symmetric) form — SparseGenRealShiftSolve (MatOp/SparseGenRealShiftSolve.h:33-99): every stored
 * entry of the CSC / CSR matrix is used; dense factorisation with partial pivoting, n <= 4096. */
int mispec_symshift_create_general(mispec_ctx* ctx, int64_t n, const int32_t* outer_host, const int32_t* inner_host,
                                   const double* val_host, int row_major, mispec_symshift** out);
int mispec_symshift_destroy(mispec_symshift* S);
int64_t mispec_symshift_rows(const mispec_symshift* S);
/* half-bandwidth of the matrix as given, of the matrix as stored (after the ordering, if one was adopted), and whether it is stored reordered */
int mispec_symshift_bandwidth(const mispec_symshift* S, int64_t* as_given, int64_t* stored, int* reordered);
/* The levels the banded path plans for an n x n matrix of this half-bandwidth (host arithmetic only): per level rows, half-bandwidth
 * (2b - 1 of the level above), chunk length, chunk count; the last level (one chunk) is the dense one.  Returns the number of
 * levels (arrays hold up to max_levels entries, may be NULL), 0 when the dense path is taken, MISPEC_EINVAL when unsupported. */
int mispec_symshift_level_plan(int64_t n, int64_t half_bandwidth, int max_levels, int64_t* rows, int64_t* bandwidth,
                               int64_t* chunk_rows, int64_t* chunks);
/* set_shift(sigma) (SparseSymShiftSolve.h:85-95): MISPEC_EINVAL "factorization failed with the given shift" on breakdown */
int mispec_symshift_set_shift(mispec_symshift* S, double sigma);
int mispec_symshift_solve(const mispec_symshift* S, const double* x_dev, double* y_dev);        /* device pointers */
/* What the last set_shift() of a banded operator found: iterative-refinement steps per solve (0 for a definite
 * A - sigma I), pivots boosted, smallest |pivot| relative to the level's scale, backward error of the probe solve.
 * Any output pointer may be NULL.  (No reference counterpart: Eigen::SparseLU pivots instead.) */
int mispec_symshift_refinement_info(const mispec_symshift* S, int* refine_steps, int64_t* boosted_pivots,
                                    double* min_pivot_ratio, double* probe_backward_error);
int mispec_symshift_solve_host(const mispec_symshift* S, const double* x_host, double* y_host); /* literal perform_op */

/* ---------------------------------------------------------------------------
 * Factorisation A V = V H + f e' kept in HBM — replaces LinAlg/Arnoldi.h + LinAlg/Lanczos.h.
 * V is local_rows x ncv column-major, H is ncv x ncv (host copy is authoritative), f is local_rows.
 * ------------------------------------------------------------------------- */
/* User operator with the reference's host-pointer contract (slow path: D2H x, call, H2D y per step).
 * Must return 0 on success. */
typedef int (*mispec_op_fn)(void* user, const double* x_in_host, double* y_out_host);

/* Exactly one of A / op must be given.  n = rows() of the operator.  symmetric=1: Lanczos (Lanczos.h),
 * symmetric=0: Arnoldi (Arnoldi.h). */
int mispec_fac_create(mispec_ctx* ctx, const mispec_csr* A, mispec_op_fn op, void* op_user, int64_t n, int ncv,
                      int symmetric, mispec_fac** out);
/* User operator on DEVICE pointers: y_dev = Op(x_dev) for n doubles each, enqueued on hip_stream (a hipStream_t; the
 * factorisation's own stream — do not synchronise it).  No staging: the Krylov vectors never leave HBM.  This is the
 * perform_op contract of the reference (SymEigsSolver.h:43-51) with the two pointers in device memory; x_dev may be a
 * column of V and must not be written.  Must return 0 on success.
 * The steps of a sweep are enqueued ahead of their execution (device-driven steps, since round 4 for these operators too):
 * the callback runs while earlier steps are still queued, so x_dev holds its data only in stream order — work issued on
 * another stream, or host code that reads x_dev, would see stale values.  A sweep that the device-side control flow stops
 * (a rare branch of Lanczos.h:99-181 / Arnoldi.h:228-291) has its remaining steps enqueued already; they turn into no-ops on
 * the library's side and the host path repeats them, so the callback can be invoked more often than num_operations() counts. */
typedef int (*mispec_device_op_fn)(void* user, const double* x_dev, double* y_dev, void* hip_stream);
int mispec_fac_create_device_op(mispec_ctx* ctx, mispec_device_op_fn op, void* op_user, int64_t n, int ncv, int symmetric,
                                mispec_fac** out);

/* ---------------------------------------------------------------------------
 * Generalized symmetric problem A x = lambda B x, regular-inverse mode
 * (SymGEigsSolver.h:224-238 with MatOp/SparseRegularInverse.h:55-127).
 * mispec_reginv is the B operator: the chosen triangle of B mirrored into a
 * CSR matrix in HBM; perform_op = B x, solve = B^{-1} x by conjugate gradient
 * (Jacobi preconditioner, tolerance epsilon, <= 2n iterations: the defaults
 * the reference inherits from Eigen::ConjugateGradient).
 * ------------------------------------------------------------------------- */
typedef struct mispec_reginv mispec_reginv;
int mispec_reginv_create(mispec_ctx* ctx, int64_t n, const int32_t* outer_host, const int32_t* inner_host,
                         const double* val_host, char uplo, int row_major, mispec_reginv** out);
int mispec_reginv_destroy(mispec_reginv* B);
int64_t mispec_reginv_rows(const mispec_reginv* B);
int mispec_reginv_perform_op_host(const mispec_reginv* B, const double* x_host, double* y_host); /* y = B x      */
int mispec_reginv_solve_host(const mispec_reginv* B, const double* x_host, double* y_host);      /* y = B^-1 x   */
int64_t mispec_reginv_last_iterations(const mispec_reginv* B);                                   /* of that solve */
/* Lanczos factorisation of y = B^{-1}(A x) in the B-inner product (MatOp/internal/SymGEigsRegInvOp.h:76-81 +
 * ArnoldiOp.h:68-101): every dot product / norm / V'f of Lanczos.h is taken as x'By.  Single GPU. */
int mispec_fac_create_geigs_reginv(mispec_ctx* ctx, const mispec_csr* A, const mispec_reginv* B, int ncv, mispec_fac** out);

/* Cholesky mode (SymGEigsSolver.h:142-208 with MatOp/SparseCholesky.h:36-128): B = G G' factored once — n <= 4096: dense
 * Cholesky factor, the triangular solves y = G^{-1} x / y = G^{-T} x as GEMVs with the explicit inverse factor in HBM;
 * n > 4096: B must be banded (half-bandwidth <= 8) and G is the factor of the partitioned band factorisation in its nested
 * order, the two solves being the two halves of the device band solve; other large patterns are rejected (MISPEC_EINVAL:
 * use the regular-inverse mode).  The operator of the standard problem is G^{-1} A G^{-T}.  mispec_cholesky_info: 0 = Successful, 3 = NumericalIssue
 * (B not positive definite), as SparseCholesky::info(). */
typedef struct mispec_cholesky mispec_cholesky;
int mispec_cholesky_create(mispec_ctx* ctx, int64_t n, const int32_t* outer_host, const int32_t* inner_host,
                           const double* val_host, char uplo, int row_major, mispec_cholesky** out);
int mispec_cholesky_destroy(mispec_cholesky* B);
int64_t mispec_cholesky_rows(const mispec_cholesky* B);
int mispec_cholesky_info(const mispec_cholesky* B);
int mispec_cholesky_lower_solve_host(const mispec_cholesky* B, const double* x_host, double* y_host); /* lower_triangular_solve */
int mispec_cholesky_upper_solve_host(const mispec_cholesky* B, const double* x_host, double* y_host); /* upper_triangular_solve */
int mispec_fac_create_geigs_cholesky(mispec_ctx* ctx, const mispec_csr* A, const mispec_cholesky* B, int ncv, mispec_fac** out);


/* Product operator y = A2 (A x) with A (p x n) and A2 (n x p) resident in HBM — the SVDTallMatOp (A2 = A') /
 * SVDWideMatOp (A = M', A2 = M) of contrib/PartialSVDSolver.h:36-110 as two chained SpMVs; symmetric (Lanczos).
 * Not available on a row-sharded context. */
int mispec_fac_create_product(mispec_ctx* ctx, const mispec_csr* A, const mispec_csr* A2, int ncv, mispec_fac** out);
int mispec_fac_create_shiftsolve(mispec_ctx* ctx, const mispec_symshift* S, int ncv, int symmetric, mispec_fac** out);
int mispec_fac_destroy(mispec_fac* fac);
/* Arnoldi::init (Arnoldi.h:136-195).  v0_host: n doubles (the GLOBAL vector; each shard takes its rows).
 * *nmatop is incremented once per operator application, like op_counter. */
int mispec_fac_init(mispec_fac* fac, const double* v0_host, int64_t* nmatop);
/* HermEigsBase::init() start vector (HermEigsBase.h:337-342): SimpleRandom(seed) generated on the
 * device by LCG jump-ahead (Util/SimpleRandom.h:30-123), then Arnoldi::init. */
int mispec_fac_init_random(mispec_fac* fac, uint64_t seed, int64_t* nmatop);
/* Lanczos::factorize_from / Arnoldi::factorize_from (Lanczos.h:62-187, Arnoldi.h:198-295). */
int mispec_fac_factorize(mispec_fac* fac, int from_k, int to_m, int64_t* nmatop);
int mispec_fac_subspace_dim(const mispec_fac* fac);          /* subspace_dim() */
int mispec_fac_f_norm(const mispec_fac* fac, double* beta);  /* f_norm() */
/* Orthogonalisation scheme of the symmetric (Lanczos) steps of a device-matrix factorisation.
 *   MISPEC_ORTH_REFERENCE (also MISPEC_ORTH=reference in the environment, which sets the default of every factorisation
 *     created afterwards): the reference's control flow, Lanczos.h:145-181 — V'f, then f -= V c with |f| and the V'f check:
 *     two passes over V per step.
 *   MISPEC_ORTH_ONESWEEP (the default since round 4: every parity gate holds in both modes): the operator is applied to the not yet
 *     corrected vector and the correction rides on the next step's pass — ONE pass over V per step.  Same decisions and the
 *     same fixed points (H follows from the Lanczos relation of the previous step), different rounding;
 *     every case the reference treats specially (second correction, breakdown clamp, tiny beta, restart heuristics) leaves
 *     the lagged path and continues with the reference's loop.  Effective for ncv <= 128 on standard problems over every operator that
 *     works on device pointers (device matrices incl. the product operator A'A / AA' of the SVD solver, the shift solve of
 *     SymEigsShiftSolver, dense matrices, user operators on device pointers, the Cholesky mode of the generalized problem);
 *     ignored otherwise (B-inner-product modes of the generalized problem, user operators with host pointers, wider bases).  For 64 < ncv <= 128 every sweep ends the reference's way (no fused restart).
 * mispec_fac_orth_info reports the mode in effect, the lagged steps executed, how often the lagged path was left because a
 * column needed a second correction (check_stops) or a correction could not be carried (state_stops), the largest accepted
 * |c|/|f| and the largest |V'v| measured after a lagged correction. */
enum { MISPEC_ORTH_REFERENCE = 0, MISPEC_ORTH_ONESWEEP = 1,
       /* flags, or-ed to MISPEC_ORTH_ONESWEEP: */
       MISPEC_ORTH_EAGER_LAST = 0x100,    /* apply the last correction of a full sweep at once (no fused restart, see below) */
       MISPEC_ORTH_TEST_RECORRECT = 0x200, /* test hook: every fused restart is followed by one more correction (see below) */
       MISPEC_ORTH_ONE_REDUCTION = 0x800,  /* one reduction per lagged step (below); MISPEC_ORTH_TWO_REDUCTIONS: alpha = <v, w> reduced on
                                              its own before the pass.  Neither flag: the library default — one reduction, MISPEC_ONE_REDUCTION=0 in the environment
                                              restores two */
       MISPEC_ORTH_TWO_REDUCTIONS = 0x1000,
       MISPEC_ORTH_TEST_RESTART_CHECK = 0x400 /* test hook: the device-side test of every fused restart reports "one correction was not
                                                enough", so none of the steps enqueued behind the restart runs and the host continues
                                                with the reference's loop before the sweep is enqueued again (see below) */ };
int mispec_fac_set_orth_mode(mispec_fac* fac, int mode);
/* One reduction per one-sweep step (MISPEC_ORTH_ONE_REDUCTION; the library's own operators — sparse and dense matrices, the SVD
 * product, the shift solve, the Cholesky mode —, ncv <= 128; CPU restatement: oracle/
 * onesweep_variant.hpp, flavour one-reduction).  The reference's step needs alpha = <v, w> before f = w - alpha v (Lanczos.h:142-145)
 * and beta = |f| before the next product (Lanczos.h:106): two global sums on the critical path, two all-reduces on a sharded run.
 * Here the product of step i + 1 runs on the UN-normalised residual f~ of step i, u = A f~, and its epilogue's partial sums of
 * <f~, u> are reduced by the kernel that reduces the record of step i's pass ([V, v_i]'f~, |f~|^2) — sharded: one all-reduce of
 * 2 i + 3 doubles; beta, alpha~ = <f~, u> / beta^2 - <f~, v_i> and w = u / beta - beta v_i follow from it (nothing cancels: both
 * norms are still measured).  The first step of a sweep and every step after the reference's own loop take the two-reduction
 * form.  mispec_fac_onered_steps counts the steps that took the one-reduction form. */
int mispec_fac_onered_steps(const mispec_fac* fac, int64_t* steps);
int mispec_fac_orth_info(const mispec_fac* fac, int* mode, int64_t* lagged_steps, int64_t* check_stops, int64_t* state_stops,
                         double* max_rel_c, double* max_chk);
/* One-sweep mode, end of a full sweep (factorize up to ncv): the correction of the LAST step stays pending as well, and
 * mispec_fac_restart_sym lets it ride on the restart's V*Q pass (one sweep over the basis instead of two) together with the
 * reference's test of the corrected residual (Lanczos.h:156).  If that test asks for a further correction (rare: max |V'f| a few
 * eps above the bar), the reference's loop continues on the compressed factorisation — V[:, :k]'f is measured again and f, H(k-2 : k-1,
 * k-1) corrected while the test fails.  Every other entry point that needs the residual (get_f, get_H, factorize, compress_V)
 * applies a pending correction first; f_norm() reports sqrt(|f~|^2 - |c|^2) until then.  mispec_fac_restart_info counts the
 * restarts that went the fused way and those that were followed by further corrections. */
int mispec_fac_restart_info(const mispec_fac* fac, int64_t* fused, int64_t* recorrected);
/* The host turns of the restarts (HermEigsBase.h:105-155 between two factorize_from calls): how many were timed, the host seconds
 * between "state of the finished sweep seen" and "restart enqueued" summed over them, and how often the pinned-memory hand-off
 * fell back to a copy (option host_turn). */
int mispec_fac_turn_info(const mispec_fac* fac, int64_t* turns, double* host_seconds, int64_t* fallbacks);
/* How a sharded device matrix moves the Krylov vector before each product: *halo = 1 if only the referenced
 * parts of the other ranks' slices are exchanged point-to-point (recv_doubles of them per product), 0 if the
 * full all-gather is used (or the context is not sharded). */
int mispec_fac_exchange_info(const mispec_fac* fac, int* halo, int64_t* recv_doubles);
/* Overlap of the exchange with the product on a row shard: the 256-row blocks [first_block, first_block + block_count) read
 * only the rank's own slice of the vector and are multiplied while the exchange is in flight on a second stream; the
 * others follow when it has landed.  block_count = 0: no overlap (unsharded, MISPEC_OVERLAP=0, or too few such blocks). */
int mispec_fac_overlap_info(const mispec_fac* fac, int* first_block, int* block_count, int* total_blocks);
/* matrix_H(), ncv x ncv col-major.  NOTE (one-sweep steps): when the last correction of a full sweep is still pending (it
 * would ride on the restart's V*Q pass), this getter, mispec_fac_get_f and mispec_fac_get_V MATERIALISE it first (kernel
 * launches, a stream synchronisation, f / H / beta updated) although the handle is const: the values returned are the finished
 * ones, and the restart that follows then takes the plain (two-pass) way instead of the fused one — same results to rounding,
 * not to the bit.  Not safe to call from two threads at once. */
int mispec_fac_get_H(const mispec_fac* fac, double* H_host);
int mispec_fac_set_H(mispec_fac* fac, const double* H_host, int k); /* after a host-side compress_H */
/* matrix_V().leftCols(ncols) / vector_f() of this shard to host (ld = local_rows). */
int mispec_fac_get_V(const mispec_fac* fac, int ncols, double* V_host);
int mispec_fac_get_f(const mispec_fac* fac, double* f_host);
const double* mispec_fac_V_dev(const mispec_fac* fac, int64_t* ld); /* device-resident accessor */
int64_t mispec_fac_local_rows(const mispec_fac* fac);

/* Ritz pairs of the projected matrix on the device: TridiagEigen::compute (LinAlg/TridiagEigen.h:121-210)
 * on H(0:ncv,0:ncv) in one workgroup (H and the eigenvector matrix live in LDS).
 * evals_host[ncv], evecs_host[ncv*ncv] col-major (may be NULL). Symmetric factorisations only. */
int mispec_fac_tridiag_eigen(mispec_fac* fac, double* evals_host, double* evecs_host);
/* The Ritz values and only the LAST ROW of their eigenvector matrix (m entries each): what the convergence test of an iteration
 * needs (HermEigsBase.h:158-175); bit-identical to the values and the last row mispec_fac_tridiag_eigen returns — the same
 * rotations, applied to one row instead of m. */
int mispec_fac_ritz_values(mispec_fac* fac, double* evals_host, double* last_row_host);
/* Implicit restart: for each shift mu (already ordered by the caller, HermEigsBase.h:118-121)
 *   TridiagQR::compute(H, mu); Q <- Q*Qi (apply_YQ); H <- Qi' H Qi (matrix_QtHQ)   (HermEigsBase.h:124-147,
 *   UpperHessenbergQR.h:515-598, :383-417, :627-693) — one workgroup, T and Q in LDS —
 * then Arnoldi::compress_V(Q) (Arnoldi.h:320-340): V[:, :k+1] <- V Q, f <- f Q(m-1,k-1) + V[:,k] H(k,k-1).
 * On return subspace_dim() == ncv - nshift and H (host copy) is the compressed H. */
int mispec_fac_restart_sym(mispec_fac* fac, const double* shifts_host, int nshift);
/* Host-side variant: the caller ran the shifted QR itself and hands over Q (ncv x ncv col-major) and the
 * compressed H / new k (compress_H already applied): only compress_V runs on the device. */
int mispec_fac_compress_V(mispec_fac* fac, const double* Q_host, const double* H_host, int new_k);
/* One implicit restart of a GENERAL factorisation entirely on the device (GenEigsBase.h:204-222, RestartArnoldi): the
 * shifts in order — kind[i] = 0: real shift a[i] (UpperHessenbergQR); kind[i] = 1: conjugate pair as the double shift
 * (s, t) = (a[i], b[i]) = (2 Re mu, |mu|^2) (DoubleShiftQR) — are applied to H by one LDS-resident kernel that also
 * accumulates Q; then V[:, :new_k+1] <- V Q and the update of f as in mispec_fac_compress_V.  ncv <= 96. */
int mispec_fac_restart_gen(mispec_fac* fac, const int* kind, const double* a, const double* b, int nshift, int new_k);
/* X = V * Y  (HermEigsBase.h:467, GenEigsBase.h:600).  Y_host: ncv x ncols col-major.
 * X_host (local_rows x ncols, may be NULL) and/or *X_dev (device buffer owned by fac, valid until the next call). */
int mispec_fac_ritz_vectors(mispec_fac* fac, const double* Y_host, int ncols, double* X_host, const double** X_dev);
/* max_j || A x_j - lambda_j x_j ||_2 / || x_j ||_2 over the ncols vectors of the last mispec_fac_ritz_vectors call,
 * evaluated on the device (parity check at sizes the CPU oracle cannot reach). resid_host[ncols]. */
int mispec_fac_residuals(mispec_fac* fac, const double* lambda_host, int ncols, double* resid_host);

/* Complex pairs of the general solver: x_j = V (Yre_j + i Yim_j), lambda_j = (re, im) interleaved;
 * resid_host[j] = || A x_j - lambda_j x_j ||_2 / || x_j ||_2, evaluated on the device. Y*: ncv x ncols col-major. */
int mispec_fac_residuals_complex(mispec_fac* fac, const double* Yre_host, const double* Yim_host, const double* lambda_host,
                                 int ncols, double* resid_host);

/* Profile of the factorisation so far: counts and accumulated HIP-event time (ms) per kernel family.
 * Timing is only collected between mispec_fac_profile(fac, level) and mispec_fac_profile(fac, 0); level 1 brackets
 * every kernel family with an event pair, level 3 the operator applications and the collectives of a sharded run, level 2 only the
 * operator applications (the SpMV roofline figure), level 4 only the passes over the basis (ms_vtf / bytes_vtf without the record reductions) —
 * the event records cost a few microseconds each, which matters when the shards are small. */
typedef struct mispec_profile
{
    int64_t n_spmv, n_vtf, n_gemv, n_scale, n_compress, n_small, n_host_sync;
    double ms_spmv, ms_vtf, ms_gemv, ms_scale, ms_compress, ms_small;
    double spmv_bytes; /* algorithmic bytes per SpMV launch of this shard: 12*nnz + 4*(rows+1) + 8*cols + 8*rows */
    /* algorithmic bytes of the length-n dense kernels, summed over their launches (8 bytes x local rows x vectors read + written):
     * vtf = the projection passes of the device-driven Lanczos steps (f = w - alpha v + V'f; one-sweep: the lagged pass),
     * compress = V <- V Q and X = V Y; gemv (the correction passes) is not counted — the device decides whether they run.  bytes / ms of the same family = the rate those kernels ran at (the family's event pair
     * also covers its record reduction, ~10 us per launch). */
    double bytes_vtf, bytes_gemv, bytes_compress;
    /* round 6 (appended): the merged record reduction behind the one-sweep passes timed on its own (level 1; it is also inside
     * ms_vtf), and — level 3, row-sharded runs — the wire: the exchange of the Krylov vector on its stream, the part of it the
     * product waits for once the interior row-blocks are done (1 - wait / exchange = the overlap achieved), the all-reduces. */
    int64_t n_reduce, n_exchange, n_exchange_wait, n_allreduce;
    double ms_reduce, ms_exchange, ms_exchange_wait, ms_allreduce;
} mispec_profile;
int mispec_fac_profile(mispec_fac* fac, int enable);
int mispec_fac_get_profile(const mispec_fac* fac, mispec_profile* out);

/* ---------------------------------------------------------------------------
 * Stand-alone small dense kernels (unit-test entry points mirroring test/QR.cpp, test/Eigen.cpp):
 * one workgroup, matrices in LDS.  All matrices n x n column-major host memory; outputs may be NULL.
 * ------------------------------------------------------------------------- */
int mispec_tridiag_qr(mispec_ctx* ctx, int n, const double* T_host, double shift, double* Q_host, double* QtHQ_host);
int mispec_tridiag_eigen(mispec_ctx* ctx, int n, const double* T_host, double* evals_host, double* evecs_host);
/* All nshift shifted QR sweeps of one restart (HermEigsBase.h:124-147: TridiagQR::compute, apply_YQ, matrix_QtHQ per shift;
 * UpperHessenbergQR.h:515-693) on the tridiagonal (diag_host[n], subd_host[n-1]), by variant — 0: host, the reference's serial
 * order; 1: host, the skewed pipeline of include/Spectra/internal/SmallDensePipelined.h (what a restart runs); 2: device,
 * k_restart_pipelined (n <= 64); 3: device, one wavefront.  Variants 0-2 give bit-identical results (variant 3 differs by
 * rounding: fused multiply-adds, its own hypot).  reps >= 1 calls are timed (host: best of reps; device: HIP events around reps
 * launches) -> *us_per_call.  diag_out[n], subd_out[n-1], Q_out[n*n] (column-major): Q'TQ and Q = Q_1 ... Q_p; any may be NULL.
 * ctx may be NULL for the host variants. */
int mispec_restart_sweeps(mispec_ctx* ctx, int n, const double* diag_host, const double* subd_host, const double* shifts_host,
                          int nshift, int variant, int reps, double* diag_out, double* subd_out, double* Q_out, double* us_per_call);

/* ---------------------------------------------------------------------------
 * Solver-level facade: Spectra::SymEigsSolver<Spectra::SparseSymMatProd<double>> (include/Spectra/)
 * instantiated inside the library, for bindings that cannot instantiate C++ templates (ctypes, cgo...).
 * Argument meaning and defaults as HermEigsBase.h:257-272, :309-342, :366-390, :395-478.
 * selection / sorting use the integer values of Spectra::SortRule (Util/SelectionRule.h:33-58).
 * ------------------------------------------------------------------------- */
typedef struct mispec_symeigs mispec_symeigs;
int mispec_symeigs_create(mispec_ctx* ctx, const mispec_csr* A, int64_t nev, int64_t ncv, mispec_symeigs** out);
int mispec_symeigs_create_op(mispec_ctx* ctx, mispec_op_fn op, void* op_user, int64_t n, int64_t nev, int64_t ncv,
                             mispec_symeigs** out);
/* The same for a user operator on device pointers (see mispec_device_op_fn); dense matrices: mispec_extras.h. */
int mispec_symeigs_create_device_op(mispec_ctx* ctx, mispec_device_op_fn op, void* op_user, int64_t n, int64_t nev, int64_t ncv,
                                    mispec_symeigs** out);
/* Spectra::SymEigsShiftSolver<Spectra::SparseSymShiftSolve<double>> (SymEigsShiftSolver.h:190-195): calls
 * set_shift(sigma) on S, iterates on (A - sigma I)^{-1} and maps the Ritz values back (lambda = 1/nu + sigma). */
/* SymGEigsSolver<SparseSymMatProd, SparseRegularInverse, GEigsMode::RegularInverse>. */
int mispec_symeigs_create_geigs_reginv(mispec_ctx* ctx, const mispec_csr* A, const mispec_reginv* B, int64_t nev, int64_t ncv,
                                       mispec_symeigs** out);
/* SymGEigsSolver<SparseSymMatProd, SparseCholesky, GEigsMode::Cholesky>; eigenvectors are back-transformed by L^{-T}. */
int mispec_symeigs_create_geigs_cholesky(mispec_ctx* ctx, const mispec_csr* A, const mispec_cholesky* B, int64_t nev, int64_t ncv,
                                         mispec_symeigs** out);
/* SymEigsSolver over the product operator of mispec_fac_create_product (PartialSVDSolver's inner solver). */
int mispec_symeigs_create_product(mispec_ctx* ctx, const mispec_csr* A, const mispec_csr* A2, int64_t nev, int64_t ncv,
                                  mispec_symeigs** out);
int mispec_symeigs_create_shift(mispec_ctx* ctx, mispec_symshift* S, int64_t nev, int64_t ncv, double sigma, mispec_symeigs** out);
int mispec_symeigs_destroy(mispec_symeigs* s);
int mispec_symeigs_init(mispec_symeigs* s, const double* v0_host /* NULL = init() */);
int mispec_symeigs_compute(mispec_symeigs* s, int selection, int64_t maxit, double tol, int sorting, int64_t* nconv);
int mispec_symeigs_info(const mispec_symeigs* s);            /* Spectra::CompInfo as int */
int64_t mispec_symeigs_num_iterations(const mispec_symeigs* s);
int64_t mispec_symeigs_num_operations(const mispec_symeigs* s);
int mispec_symeigs_eigenvalues(const mispec_symeigs* s, double* out_host, int64_t* count);
/* out_host: local_rows x min(nvec, nconv) col-major (may be NULL to keep the result on the device only). */
int mispec_symeigs_eigenvectors(mispec_symeigs* s, int64_t nvec, double* out_host, int64_t* ncols);
/* Residuals ||A x - lambda x|| / ||x|| of the converged pairs, computed on the device. */
int mispec_symeigs_residuals(mispec_symeigs* s, double* resid_host, int64_t* count);
int mispec_symeigs_get_profile(const mispec_symeigs* s, mispec_profile* out);
int mispec_symeigs_exchange_info(const mispec_symeigs* s, int* halo, int64_t* recv_doubles); /* see mispec_fac_exchange_info */
int mispec_symeigs_overlap_info(const mispec_symeigs* s, int* first_block, int* block_count, int* total_blocks);
int mispec_symeigs_profile(mispec_symeigs* s, int enable);
/* Orthogonalisation scheme of the Lanczos steps (see mispec_fac_set_orth_mode): MISPEC_ORTH_ONESWEEP (default) or
 * MISPEC_ORTH_REFERENCE, the reference's two-pass control flow.  Call before init() / compute(). */
int mispec_symeigs_set_orth_mode(mispec_symeigs* s, int mode);
int mispec_symeigs_orth_info(const mispec_symeigs* s, int* mode, int64_t* lagged_steps, int64_t* check_stops,
                             int64_t* state_stops, double* max_rel_c, double* max_chk);
int mispec_symeigs_restart_info(const mispec_symeigs* s, int64_t* fused, int64_t* recorrected); /* see mispec_fac_restart_info */
int mispec_symeigs_turn_info(const mispec_symeigs* s, int64_t* turns, double* host_seconds, int64_t* fallbacks); /* see mispec_fac_turn_info */
int mispec_symeigs_onered_steps(const mispec_symeigs* s, int64_t* steps);                        /* see mispec_fac_onered_steps */

/* ---------------------------------------------------------------------------
 * General (non-symmetric) solver: Spectra::GenEigsSolver<Spectra::SparseGenMatProd<double>> behind a handle.
 * Argument meaning and defaults as GenEigsBase.h:419-423, :442-476, :501-525, :548-610.
 * Complex results are interleaved (re, im) doubles.
 * ------------------------------------------------------------------------- */
typedef struct mispec_geneigs mispec_geneigs;
int mispec_geneigs_create(mispec_ctx* ctx, const mispec_csr* A, int64_t nev, int64_t ncv, mispec_geneigs** out);
int mispec_geneigs_create_op(mispec_ctx* ctx, mispec_op_fn op, void* op_user, int64_t n, int64_t nev, int64_t ncv,
                             mispec_geneigs** out);
/* GenEigsSolver over a user operator on device pointers (dense matrices, complex shifts: mispec_extras.h) */
int mispec_geneigs_create_device_op(mispec_ctx* ctx, mispec_device_op_fn op, void* op_user, int64_t n, int64_t nev, int64_t ncv,
                                    mispec_geneigs** out);
/* GenEigsRealShiftSolver<SparseGenRealShiftSolve> (GenEigsRealShiftSolver.h:36-82): Arnoldi on (A - sigma I)^{-1},
 * eigenvalues mapped back by lambda = 1/nu + sigma.  Calls set_shift(sigma) on S. */
int mispec_geneigs_create_shift(mispec_ctx* ctx, mispec_symshift* S, int64_t nev, int64_t ncv, double sigma, mispec_geneigs** out);
int mispec_geneigs_destroy(mispec_geneigs* s);
int mispec_geneigs_init(mispec_geneigs* s, const double* v0_host /* NULL = init() */);
int mispec_geneigs_compute(mispec_geneigs* s, int selection, int64_t maxit, double tol, int sorting, int64_t* nconv);
int mispec_geneigs_info(const mispec_geneigs* s);
int64_t mispec_geneigs_num_iterations(const mispec_geneigs* s);
int64_t mispec_geneigs_num_operations(const mispec_geneigs* s);
int mispec_geneigs_eigenvalues(const mispec_geneigs* s, double* out_host /* 2*count */, int64_t* count);
/* out_host: local_rows x ncols complex, column-major, interleaved (2 * local_rows * ncols doubles) */
int mispec_geneigs_eigenvectors(mispec_geneigs* s, int64_t nvec, double* out_host, int64_t* ncols);
/* || A x - lambda x || / || x || of the converged pairs, evaluated on the device (device matrices only) */
int mispec_geneigs_residuals(mispec_geneigs* s, double* resid_host, int64_t* count);
int mispec_geneigs_get_profile(const mispec_geneigs* s, mispec_profile* out);
int mispec_geneigs_profile(mispec_geneigs* s, int enable);

/* ---------------------------------------------------------------------------
 * Host-side ncv x ncv kernels of the general restart (include/Spectra/internal/SmallDenseGen.h), exported
 * for parity tests and non-C++ callers.  They need no GPU.  All matrices n x n column-major.
 *   mispec_hess_qr_host          UpperHessenbergQR: H - sI = QR ; Q, Q'HQ = RQ + sI     (UpperHessenbergQR.h:136-255)
 *   mispec_double_shift_qr_host  DoubleShiftQR: H^2 - sH + tI = QR ; Q, Q'HQ            (DoubleShiftQR.h:358-467)
 *   mispec_hess_schur_host       UpperHessenbergSchur: H = U T U'                       (UpperHessenbergSchur.h:354-421)
 *   mispec_hess_eigen_host       UpperHessenbergEigen: complex eigenpairs, interleaved  (UpperHessenbergEigen.h:231-327)
 * ------------------------------------------------------------------------- */
int mispec_hess_qr_host(int n, const double* H, double shift, double* Q, double* QtHQ);
/* The two sweeps of the general restart as HIP kernels (one wavefront, H and Q in LDS, 3 <= n <= 96): the device
 * counterparts of mispec_hess_qr_host / mispec_double_shift_qr_host (UpperHessenbergQR.h:136-255, DoubleShiftQR.h:334-467);
 * host arrays in and out.  The *_lanes_host forms run the kernels' source (internal/SmallDenseGenLanes.h) with one lane on
 * the host.  Inside a solve the whole shift list of a restart is one launch: mispec_fac_restart_gen. */
int mispec_hess_qr(mispec_ctx* ctx, int n, const double* H_host, double shift, double* Q_host, double* QtHQ_host);
int mispec_double_shift_qr(mispec_ctx* ctx, int n, const double* H_host, double s, double t, double* Q_host, double* QtHQ_host);
int mispec_hess_qr_lanes_host(int n, const double* H, double shift, double* Q, double* QtHQ);
int mispec_double_shift_qr_lanes_host(int n, const double* H, double s, double t, double* Q, double* QtHQ);
int mispec_double_shift_qr_host(int n, const double* H, double s, double t, double* Q, double* QtHQ);
int mispec_hess_schur_host(int n, const double* H, double* T, double* U);
int mispec_hess_eigen_host(int n, const double* H, double* evals, double* evecs);

#ifdef __cplusplus
}
#endif
#endif /* MISPEC_H */
