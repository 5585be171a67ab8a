// y = (A - sigma I)^{-1} x on the GPU for symmetric A — the operator behind SymEigsShiftSolver
// (replaces MatOp/SparseSymShiftSolve.h:85-109, which delegates to Eigen::SparseLU).
//
// Two factorisations, chosen from the half-bandwidth b of A - sigma I at set_shift():
//   * banded (b <= 8 at any n, b <= 64 beyond n = 4096 — as given or after a reverse Cuthill-McKee ordering): a recursive "partition + Schur complement" LDL' — the parallel form of a band
//     solve.  The rows are cut into chunks of L rows; the last b rows of every chunk form a separator,
//     the rest (the interior) of different chunks are decoupled.  Factor (once per shift; b <= 8: top levels on the device,
//     k_chunk_factor; wider bands: on the host's cores, chunk-parallel): banded
//     LDL' of every interior block, the spikes W = M_II^{-1} M_IS, and the Schur complement of the
//     separators, which is again banded (half-bandwidth 2b-1) and is factored the same way, recursively,
//     until it fits one chunk.  Solve (device, every Lanczos step), per level three kernels:
//        k_chunk_solve  one thread per chunk: forward/backward substitution on its interior block
//                       (factors stored chunk-interleaved => coalesced across the threads of a wave);
//                       k_chunk_solve_lds: the same, LDS-staged, for b <= 8 (the measured path, C5);
//                       k_chunk_solve_wave (round 6): one WAVEFRONT per chunk for b = 9...64 on a row-major factor
//        k_sep_rhs      g_S = f_S - M_SI y_I
//        k_back_subst   x_I = y_I - W x_S   (one thread per row, fully parallel)
//     The sequential depth per level is L rows instead of n.  No pivoting inside the chunks: exact when A - sigma I
//     is definite (sigma outside the spectrum, config 5).  For a shift INSIDE the spectrum (the usual use of
//     shift-and-invert, test/SymEigsShift.cpp:119-184) a chunk's leading minors may be (nearly) singular although
//     A - sigma I is not, so the factorisation is made robust the way partitioned band solvers are (SPIKE's "diagonal
//     boosting"): a pivot below sqrt(eps) * scale is replaced by +-sqrt(eps) * scale (the factors are then those of a
//     slightly perturbed matrix), the last level is a band LU with partial pivoting, and set_shift() CALIBRATES the
//     number of iterative-refinement steps (r = x - (A - sigma I) y on the device from the resident band, y += solve(r))
//     that bring the backward error of a probe solve to rounding level; every later solve runs that many steps
//     (0 for a definite matrix).  A shift for which refinement does not converge is reported as a failed
//     factorisation, like the reference does for a singular shift (SparseSymShiftSolve.h:93-94).
//   * dense (n <= 4096, any sparsity — the reference's own test fixtures are of this kind): LU with partial
//     pivoting of the dense A - sigma I on the host, explicit inverse, and a dense GEMV kernel per step.
// Anything else (large n with large bandwidth) is rejected: a general sparse LU on the GPU is out of scope.
#include "shiftsolve.hpp"
#include "reorder.hpp"
#include "dense.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

using namespace mispec;

namespace {

constexpr int kThreads = 256;

// ---------------------------------------------------------------------------------------------------
// host-side band matrix (symmetric, lower band incl. diagonal): a(i, d) = M(i, i - d), 0 <= d <= b
// ---------------------------------------------------------------------------------------------------
struct HostBand
{
    int64_t n = 0;
    int b = 0;
    std::vector<double> a;  // n x (b + 1), row-major
    // read-only view of an unshifted band kept elsewhere (the operator's resident copy): M = *view - shift * I.
    // Used for the top level when it is factored on the device, so that no shifted host copy has to be built.
    const std::vector<double>* view = nullptr;
    const double* view_dev = nullptr;  // the same unshifted band in device memory
    const std::vector<double>* viewB = nullptr;  // pencil: M = *view - shift * *viewB (same layout); nullptr: B = I
    const double* viewB_dev = nullptr;
    double shift = 0.0;
    double& at(int64_t i, int d) { return a[size_t(i) * (b + 1) + d]; }
    double at(int64_t i, int d) const
    {
        if (view)
            return (*view)[size_t(i) * (b + 1) + d] - shift * (viewB ? (*viewB)[size_t(i) * (b + 1) + d] : (d == 0 ? 1.0 : 0.0));
        return a[size_t(i) * (b + 1) + d];
    }
    double get(int64_t i, int64_t j) const  // symmetric access, 0 outside the band
    {
        if (i < j)
            std::swap(i, j);
        const int64_t d = i - j;
        return d <= b ? at(i, int(d)) : 0.0;
    }
};

// Where the factor of chunk p, row k lives.  Chunks are stored in groups of 64 — one wavefront of the solve kernels — and
// lane-interleaved inside a group: entries (2j, d) and (2j + 1, d) of the 64 chunks of group g are one 1024-byte line at
// ((g * R0 / 2 + j) * b + d) * 128, so a wavefront's loads are coalesced and consecutive (j, d) are a compile-time stride apart.
// The last chunk (no separator; it also takes the remainder of N / L and may be almost twice as long) is lane 0 of a final
// group with Rt rows.  R0 / Rt are the interior lengths rounded up to the batch length of the sweeps plus a margin, and the
// padding is ZERO: a sweep may run whole batches past the end of a chunk (z = 0 there) and read L(k + d + 1, d) without a test.
struct ChunkStore
{
    int64_t P = 1, R0 = 0, Rt = 0, groups = 0;  // groups: full-length groups, = ceil((P - 1) / 64)
    int b = 0;
    // wide != 0 (round 6, half-bandwidths 9...64 solved by one WAVEFRONT per chunk, k_chunk_solve_wave): plain row-major
    // storage, chunk p's row k at row p * R0 + k (the last chunk has Rt rows), a row's b entries contiguous
    int wide = 0;
    __host__ __device__ int64_t row_base(int64_t p) const { return (p == P - 1 ? groups : (p >> 6)) * R0; }  // R0, Rt even
    __host__ __device__ int lane(int64_t p) const { return p == P - 1 ? 0 : int(p & 63); }
    // rows are stored in PAIRS: a lane's entries of rows 2j and 2j + 1 are adjacent, one 16-byte load (see k_chunk_solve_lds)
    __host__ __device__ size_t lf(int64_t p, int64_t k, int d) const
    {
        if (wide)
            return (size_t(p) * size_t(R0) + size_t(k)) * size_t(b) + size_t(d);
        return (size_t((row_base(p) + k) >> 1) * b + d) * 128 + size_t(lane(p)) * 2 + size_t(k & 1);
    }
    __host__ __device__ size_t dinv(int64_t p, int64_t k) const
    {
        if (wide)
            return size_t(p) * size_t(R0) + size_t(k);
        return size_t((row_base(p) + k) >> 1) * 128 + size_t(lane(p)) * 2 + size_t(k & 1);
    }
    __host__ __device__ size_t rows() const { return size_t(wide ? P - 1 : groups) * R0 + Rt; }
    // (wide: 192 rows of slack behind the last chunk — the wave kernel loads its coefficients two batches ahead without a clamp)
    __host__ __device__ size_t lf_size() const { return (rows() + (wide ? 192 : 0)) * (b > 0 ? b : 1) * (wide ? 1 : 64); }
    __host__ __device__ size_t dinv_size() const { return rows() * (wide ? 1 : 64); }
};
constexpr int kSweepBatch = 32;  // longest batch of the sweep kernels: the row padding of ChunkStore
inline ChunkStore make_chunk_store(int64_t N, int b, int64_t L, int64_t P, bool wide = false)
{
    ChunkStore cs;
    cs.P = P;
    cs.b = b;
    cs.wide = wide ? 1 : 0;
    cs.groups = (P - 1 + 63) / 64;
    const auto padded = [](int64_t m) { return (m + kSweepBatch - 1) / kSweepBatch * kSweepBatch + 16; };
    cs.R0 = P > 1 ? padded(L - b) : 0;
    cs.Rt = padded(N - (P - 1) * L);
    return cs;
}

// ---- kernels ------------------------------------------------------------------------------------------
// Chunk p owns rows [p*L, min((p+1)*L, N)); its interior is the chunk minus the last b rows (the last chunk
// has no separator).  Lf(k, d, p) multiplies z_{k-d-1}; everything chunk-interleaved: ChunkStore above.
//
// One thread per chunk.  The recurrence is sequential in k, so the only latency that may sit on the critical
// path is the FMA chain itself: the last B unknowns live in registers (never re-read from memory), and the
// factor entries / right-hand sides of the next U rows are loaded as one batch before they are needed.
//
// Every wavefront runs chunks of ONE interior length: the last chunk has a workgroup of its own.  Row counters and trip counts
// are therefore wave-uniform, and the factor addresses are `per-lane base + compile-time stride` (ChunkStore).
//
// What bounds these kernels (measured, profiles/rounds_1_2/r03e-r03g): with one wavefront per SIMD nothing overlaps the instruction
// stream, so the time is the NUMBER OF INSTRUCTIONS per row.  The first version (per-lane interior length in the loop bounds,
// factor index (k*b + d)*P + p) needed five 64-bit vector integer instructions and a branch per load — 540 cycles per row,
// unchanged by deeper batches, more wavefronts, or staging the vector in LDS.
constexpr int kChunkThreads = 64;  // one wavefront per workgroup: the chunks spread over all CUs, 512 VGPRs per lane
// General kernel (any b <= B, Cholesky halves):
// mode 0: y = M_II^{-1} f (forward, diagonal, backward); with u_out also u = D^{-1/2} L^{-1} f from the forward sweep — the
//         interior part of G^{-1} f for the Cholesky-like factor G of a positive definite band (cholesky.hip);
// mode 2: y = L^{-T} D^{-1/2} f (backward sweep only): the interior part of G^{-T}.
template <int B, int U>
__global__ __launch_bounds__(kChunkThreads) void k_chunk_solve(int64_t N, int64_t L, ChunkStore cs, const double* __restrict__ Lf,
                                                           const double* __restrict__ Dinv, const double* __restrict__ f,
                                                           double* __restrict__ y, int mode, double* __restrict__ u_out)
{
    const int64_t P = cs.P;
    const int b = cs.b;
    const bool tail_block = blockIdx.x == gridDim.x - 1;  // the last chunk alone
    const int64_t p = tail_block ? P - 1 : int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (tail_block ? threadIdx.x != 0 : p >= P - 1)
        return;
    const int m = int(tail_block ? N - (P - 1) * L : L - b);  // interior rows of this wavefront's chunks
    const double* const fp = f + p * L;
    double* const yp = y + p * L;
    double* const up = u_out ? u_out + p * L : nullptr;
    const double* const lfw = Lf + cs.lf(p, 0, 0);    // entry (k, d): lfw[((k >> 1) * b + d) * 128 + (k & 1)]
    const double* const dvw = Dinv + cs.dinv(p, 0);  // row k: dvw[(k >> 1) * 128 + (k & 1)]
    double hist[B];  // hist[d] = unknown k-d-1 (forward) / k+d+1 (backward)
#pragma unroll
    for (int d = 0; d < B; d++)
        hist[d] = 0.0;
    // forward: z_k = f_k - sum_d Lf(k,d) z_{k-d-1}
    for (int k0 = 0; k0 < (mode == 2 ? 0 : m); k0 += U)
    {
        double fk[U], lf[U][B], ds[U];
#pragma unroll
        for (int u = 0; u < U; u++)
        {
            const int k = (k0 + u < m) ? k0 + u : m - 1;
            fk[u] = fp[k];
            ds[u] = u_out ? sqrt(fabs(dvw[int64_t(k >> 1) * 128 + (k & 1)])) : 0.0;
#pragma unroll
            for (int d = 0; d < B; d++)
                lf[u][d] = (d < b) ? lfw[(int64_t(k >> 1) * b + d) * 128 + (k & 1)] : 0.0;  // rows k < b: zeros for the missing neighbours
        }
#pragma unroll
        for (int u = 0; u < U; u++)
        {
            if (k0 + u < m)
            {
                double acc = fk[u];
#pragma unroll
                for (int d = B - 1; d >= 0; d--)  // the most recent unknown (d = 0) enters last
                    acc = fma(-lf[u][d], hist[d], acc);  // explicit: both solve kernels round identically
#pragma unroll
                for (int d = B - 1; d > 0; d--)
                    hist[d] = hist[d - 1];
                hist[0] = acc;
                yp[k0 + u] = acc;
                if (u_out)
                    up[k0 + u] = acc * ds[u];
            }
        }
    }
    // diagonal and backward: y_k = z_k / D_k - sum_d Lf(k+d+1, d) y_{k+d+1}   (mode 2: z_k |D_k|^{-1/2} from f instead)
#pragma unroll
    for (int d = 0; d < B; d++)
        hist[d] = 0.0;
    for (int k0 = m - 1; k0 >= 0; k0 -= U)
    {
        double zk[U], di[U], lf[U][B];
#pragma unroll
        for (int u = 0; u < U; u++)
        {
            const int k = (k0 - u >= 0) ? k0 - u : 0;
            zk[u] = (mode == 2) ? fp[k] : yp[k];
            di[u] = (mode == 2) ? sqrt(fabs(dvw[int64_t(k >> 1) * 128 + (k & 1)])) : dvw[int64_t(k >> 1) * 128 + (k & 1)];
#pragma unroll
            for (int d = 0; d < B; d++)
                lf[u][d] = (d < b && k + d + 1 < m) ? lfw[(int64_t((k + d + 1) >> 1) * b + d) * 128 + ((k + d + 1) & 1)] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; u++)
        {
            if (k0 - u >= 0)
            {
                double acc = __dmul_rn(zk[u], di[u]);
#pragma unroll
                for (int d = B - 1; d >= 0; d--)
                    acc = fma(-lf[u][d], hist[d], acc);  // explicit: both solve kernels round identically
#pragma unroll
                for (int d = B - 1; d > 0; d--)
                    hist[d] = hist[d - 1];
                hist[0] = acc;
                yp[k0 - u] = acc;
            }
        }
    }
}


// ---- one WAVEFRONT per chunk: half-bandwidths 9...64 (round 6) --------------------------------------------------------------
// One lane per chunk (k_chunk_solve above) leaves a level of wide-band chunks — 32 b rows each, P = N / (32 b) of them — with
// P / 64 wavefronts on the whole device, each walking b products per row in sequence: 26 ms per solve at n = 1e6, b = 32.  Here a
// wavefront owns ONE chunk and its lane l owns the rows r = l (mod 64).  Both sweeps run in scatter form: as soon as the unknown of
// row k is final (lane k mod 64 holds it; every lane reads it with one v_readlane), every lane adds its product to the row it is
// working on — forward: row r = k + d + 1 gets -L(r, d) z_k, backward: row r = k - d - 1 gets -L(k, d) y_k, d = the distance of the
// lane's row —, and the lane that has just finished takes its next row (64 further on), whose start value it loaded a block ago.
// A row thus receives its b products in exactly the order of k_chunk_solve's chain (forward: oldest unknown first, backward: the
// farthest first; the start value f_k resp. z_k D_k^{-1} first, every step one fma): the same bits, and the critical path of a
// step is readlane -> fma.  Coefficients come from the row-major layout (ChunkStore::wide), one coalesced piece of b doubles per
// step: the backward sweep reads row k of L, the forward sweep row k of the SHIFTED copy LT(k, d) = L(k + d + 1, d) that
// k_shift_factor builds on the device once per factorisation (reading L down its anti-diagonals instead — 64 different lines per
// load — ran the first version into the L1: 0.4 us per step).  Modes as k_chunk_solve (0: solve, + u_out; 2: backward half only).
constexpr int kWaveChunks = 4;  // chunks (wavefronts) per workgroup
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// LT(k, d) = L(k + d + 1, d) for the interior rows of every chunk (zero where row k + d + 1 lies beyond the interior): the forward
// sweep's coefficients, row-contiguous.  One thread per entry.
__global__ __launch_bounds__(256) void k_shift_factor(int64_t N, int64_t L, ChunkStore cs, const double* __restrict__ Lf, double* __restrict__ LfT)
{
    const int64_t P = cs.P;
    const int b = cs.b;
    const int64_t p = blockIdx.y;
    const int64_t m = (p == P - 1) ? N - (P - 1) * L : L - b;
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= m * b)
        return;
    const int64_t k = e / b;
    const int d = int(e - k * b);
    const int64_t r = k + d + 1;
    LfT[cs.lf(p, k, d)] = (r < m) ? Lf[cs.lf(p, r, d)] : 0.0;
}
__global__ __launch_bounds__(64 * kWaveChunks) void k_chunk_solve_wave(int64_t N, int64_t L, ChunkStore cs, const double* __restrict__ Lf,
                                                                      const double* __restrict__ LfT, const double* __restrict__ Dinv,
                                                                      const double* __restrict__ f,
                                                                      double* __restrict__ y, int mode, double* __restrict__ u_out)
{
    // The coefficients of a batch of U = 32 steps are loaded TWO batches ahead of their use into one of three register sets whose
    // roles rotate (no copies: a copy would wait for the loads it moves): a step is ~50 cycles of readlane -> fma, an uncached
    // load 1-2 us.  Everything that steers the loops is made wave-uniform explicitly (readfirstlane): with a per-lane `m` the
    // compiler predicates every step on exec and waits for ALL outstanding loads in each (first version: 0.4 us per step).
    constexpr int U = 32;
    const int lane = threadIdx.x & 63;
    const int64_t P = cs.P;
    const int64_t p = int64_t(blockIdx.x) * kWaveChunks + __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    if (p >= P)
        return;
    const int b = cs.b;
    const int m = __builtin_amdgcn_readfirstlane(int(p == P - 1 ? N - (P - 1) * L : L - b));  // interior rows of this chunk
    if (m <= 0)
        return;
    const double* const fp = f + p * L;
    double* const yp = y + p * L;
    double* const up = u_out ? u_out + p * L : nullptr;
    const double* const lf = Lf + cs.lf(p, 0, 0);    // L(k, d) = lf[k * b + d]: what y_k contributes to row k - d - 1 (backward)
    const double* const lft = LfT + cs.lf(p, 0, 0);  // L(k + d + 1, d) = lft[k * b + d]: what z_k contributes to row k + d + 1 (forward)
    const double* const dv = Dinv + cs.dinv(p, 0);  // 1 / D_k
    const int nblk = (m + 63) >> 6;
    const int nbat = 2 * nblk;  // batches of 32 steps: two per 64-row block
    double c0[U], c1[U], c2[U];

    if (mode != 2)
    {
        // ---- forward: z_k = f_k - sum_d L(k, d) z_{k-d-1}; lane l works on row blk * 64 + l, then on the next block's ----
        double A = (lane < m) ? fp[lane] : 0.0;
        double fnext = (64 + lane < m) ? fp[64 + lane] : 0.0;  // start values of the rows of the next block
        double f2 = 0.0, zbuf = 0.0;
        // the coefficients of the batch that starts at step kb: what z_k contributes to this lane's row k + d + 1.  Unconditional
        // loads at 32-bit offsets from one base per batch (rows beyond the chunk are the next chunk's or the slack behind the
        // last: valid memory, masked to 0) — no branch, no 64-bit multiply per load
        const auto load_batch = [&](double (&tgt)[U], int kb) {
            const double* const rowp = lft + int64_t(kb) * b;
            const int d0 = (lane - kb - 1) & 63;
#pragma unroll
            for (int u = 0; u < U; u++)
            {
                const int d = (d0 - u) & 63;
                tgt[u] = rowp[u * b + (d < b ? d : b - 1)];  // raw: masked where it is USED (a select here would wait for the load at once)
            }
        };
        // batch `bat` with the coefficients in `cur`; the loads of batch bat + 2 go to `tgt`
        const auto batch = [&](int bat, const double (&cur)[U], double (&tgt)[U]) {
            if (bat >= nbat)
                return;
            const int k0 = bat * U;
            const int blk = bat >> 1;
            if ((bat & 1) == 0)
                f2 = (int64_t(blk + 2) * 64 + lane < m) ? fp[(blk + 2) * 64 + lane] : 0.0;  // two blocks ahead
            load_batch(tgt, k0 + 2 * U);
            const int kk0 = k0 & 63;  // 0 or 32
#pragma unroll
            for (int u = 0; u < U; u++)
            {
                // (no branch inside a batch — straight-line code lets the compiler count the loads in flight; a step beyond
                // the last row is harmless: its rows start from 0 and every coefficient there is 0)
                const double z = readlane_f64(A, kk0 + u);
                if (lane == kk0 + u)
                {
                    zbuf = z;
                    A = fnext;  // this lane's next row: k + 64
                }
                const int d = (lane - kk0 - u - 1) & 63;
                const double c = ((k0 + u < m) & (d < b)) ? cur[u] : 0.0;
                A = fma(-c, z, A);
            }
            if (bat & 1)
            {
                const int64_t r = int64_t(blk) * 64 + lane;
                if (r < m)
                {
                    yp[r] = zbuf;
                    if (up)
                        up[r] = zbuf * sqrt(fabs(dv[r]));
                }
                fnext = f2;
            }
        };
        load_batch(c0, 0);
        load_batch(c1, U);
        for (int bat = 0; bat < nbat; bat += 3)
        {
            batch(bat, c0, c2);
            batch(bat + 1, c1, c0);
            batch(bat + 2, c2, c1);
        }
    }
    // ---- diagonal and backward: y_k = z_k / D_k - sum_d L(k+d+1, d) y_{k+d+1}  (mode 2: z_k |D_k|^{-1/2} from f instead) ----
    {
        const auto start = [&](int64_t r) -> double {  // start value of row r (the wavefront's own writes above are visible to it)
            if (r < 0 || r >= m)
                return 0.0;
            return (mode == 2) ? __dmul_rn(fp[r], sqrt(fabs(dv[r]))) : __dmul_rn(yp[r], dv[r]);
        };
        const int top = nblk - 1;
        // lanes beyond the last row of the top block start in the block below (they finish no row while the top block is swept)
        double A = (int64_t(top) * 64 + lane < m) ? start(int64_t(top) * 64 + lane) : start(int64_t(top - 1) * 64 + lane);
        double snext = start(int64_t(top - 1) * 64 + lane);  // what a lane takes when it finishes its row of the block being swept
        double s2a = 0.0, s2b = 0.0, ybuf = 0.0;  // raw operands of the start values two blocks down (combined when they are taken)
        int64_t s2r = -1;
        // the coefficients of batch bt (steps k = bt * 32 + 31 - u): what y_k contributes to this lane's row k - d - 1
        const auto load_batch = [&](double (&tgt)[U], int bt) {
            const int kb = (bt > 0 ? bt : 0) * U;  // (a batch below the first: loads of batch 0, masked)
            const double* const rowp = lf + int64_t(kb) * b;
            const int e0 = ((kb + U - 1) & 63) - lane - 1;
#pragma unroll
            for (int u = 0; u < U; u++)
            {
                const int d = (e0 - u) & 63;
                tgt[u] = rowp[(U - 1 - u) * b + (d < b ? d : b - 1)];  // raw: masked where it is used
            }
        };
        // batch `bat` (descending from nbat - 1): steps k = bat * 32 + 31 - u
        const auto batch = [&](int bat, const double (&cur)[U], double (&tgt)[U]) {
            if (bat < 0)
                return;
            const int blk = bat >> 1;
            if (bat & 1)
            {
                // loaded a block ahead of their use, RAW (arithmetic on them here would wait for every load in flight)
                s2r = int64_t(blk - 2) * 64 + lane;
                const int64_t rc = (s2r >= 0 && s2r < m) ? s2r : 0;
                s2a = (mode == 2) ? fp[rc] : yp[rc];
                s2b = dv[rc];
            }
            load_batch(tgt, bat - 2);
            const int kk1 = (bat * U + U - 1) & 63;  // 63 or 31
#pragma unroll
            for (int u = 0; u < U; u++)
            {
                // (no branch: see above; a step above the last row must not finish a lane's row — `live` —, its coefficient is 0)
                const bool live = bat * U + U - 1 - u < m;
                const double yk = readlane_f64(A, kk1 - u);
                if ((lane == kk1 - u) & live)
                {
                    ybuf = yk;
                    A = snext;  // this lane's next row: k - 64
                }
                const int d = (kk1 - u - lane - 1) & 63;
                const double c = (live & (d < b)) ? cur[u] : 0.0;
                A = fma(-c, yk, A);
            }
            if ((bat & 1) == 0)
            {
                const int64_t r = int64_t(blk) * 64 + lane;
                if (r < m)
                    yp[r] = ybuf;
                snext = (s2r >= 0 && s2r < m) ? ((mode == 2) ? __dmul_rn(s2a, sqrt(fabs(s2b))) : __dmul_rn(s2a, s2b)) : 0.0;
            }
        };
        load_batch(c0, nbat - 1);
        load_batch(c1, nbat - 2);
        for (int bat = nbat - 1; bat >= 0; bat -= 3)
        {
            batch(bat, c0, c2);
            batch(bat - 1, c1, c0);
            batch(bat - 2, c2, c1);
        }
    }
}

// The plain solve of a level (the hot one), half-bandwidth exactly B: as few instructions per row as the recurrence allows.
//  * the wavefront's piece of the vector is staged in LDS: its chunks are contiguous rows, copied in and out with coalesced
//    accesses (one lane per chunk would make every access to f / y a 64-line gather); row stride in LDS odd, no bank conflicts.
//    The copies are done by four wavefronts (one alone has 32 x 512 bytes in flight per round trip to HBM: 14.6 of the
//    kernel's 39.7 us, measured by skipping it), the sweeps by the first;
//  * every batch is a full one: the sweeps run to the interior length rounded up to U over the zero padding of ChunkStore
//    (z = 0, y = 0 there), so there is no clamp, no guard and no branch between the loads;
//  * B is a template parameter: no `d < b` tests, and consecutive factor entries are 512 bytes apart at compile time.
//  * a scheduling barrier separates the loads of a batch from its recurrence: left alone, the compiler sinks each load next to
//    its use to save registers and keeps five or six in flight (s_waitcnt vmcnt(5) between the FMAs in the ISA), which makes
//    every row wait for half a memory latency — that, not bandwidth or address arithmetic, was the 540 cycles per row.
// Same operations in the same order as k_chunk_solve: bit-identical results.
constexpr int kStageThreads = 256;  // the workgroup: the sweeps run on its first wavefront, all four copy the vector in and out
template <int B, int U>
__global__ __launch_bounds__(kStageThreads) void k_chunk_solve_lds(int64_t N, int64_t L, ChunkStore cs, int ldl, int chunks_per_block,
                                                                   const double* __restrict__ Lf, const double* __restrict__ Dinv,
                                                                   const double* __restrict__ f, double* __restrict__ y)
{
    extern __shared__ double seg[];  // chunks_per_block chunks x ldl
    const int64_t P = cs.P;
    const bool tail_block = blockIdx.x == gridDim.x - 1;  // the last chunk alone
    const int64_t first = tail_block ? P - 1 : int64_t(blockIdx.x) * chunks_per_block;
    const int nch = tail_block ? 1 : int(min(int64_t(chunks_per_block), P - 1 - first));
    const int m = int(tail_block ? N - (P - 1) * L : L - B);  // interior rows of this wavefront's chunks
    const int mr = (m + U - 1) / U * U;                        // ... in whole batches
    const int lane = threadIdx.x, nl = blockDim.x;
    {
        // copy in: the wavefront's rows are one contiguous piece of f; 32 loads per lane in flight, then the LDS writes
        // (row i of the piece is row k = i mod L of chunk c = i div L, followed incrementally; separator rows are skipped)
        const double* src = f + first * L;
        const int64_t total = tail_block ? m : int64_t(nch) * L;
        constexpr int kInFlight = 32;
        int c = 0;
        int64_t k = lane;
        while (k >= L && !tail_block)
        {
            k -= L;
            c++;
        }
        for (int64_t i0 = lane; i0 < total; i0 += int64_t(kInFlight) * nl)
        {
            double t[kInFlight];
#pragma unroll
            for (int q = 0; q < kInFlight; q++)
                t[q] = (i0 + int64_t(q) * nl < total) ? src[i0 + int64_t(q) * nl] : 0.0;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < kInFlight; q++)
            {
                if (i0 + int64_t(q) * nl < total && k < m)
                    seg[c * ldl + k] = t[q];
                k += nl;
                while (k >= L && !tail_block)
                {
                    k -= L;
                    c++;
                }
            }
        }
        for (int kz = m + lane; kz < mr; kz += nl)
            for (int cz = 0; cz < nch; cz++)
                seg[cz * ldl + kz] = 0.0;
    }
    __syncthreads();
    if (lane < nch)
    {
        const int64_t p = first + lane;
        double* const zp = seg + lane * ldl;
        const double* const lfw = Lf + cs.lf(p, 0, 0);
        const double* const dvw = Dinv + cs.dinv(p, 0);
        double hist[B];
        // one batch of the forward / backward recurrence on factor entries already in registers
        const auto forward = [&](const double (&lf)[U][B], int k0) {
#pragma unroll
            for (int u = 0; u < U; u++)
            {
                double acc = zp[k0 + u];
#pragma unroll
                for (int d = B - 1; d >= 0; d--)
                    acc = fma(-lf[u][d], hist[d], acc);  // explicit: both solve kernels round identically
#pragma unroll
                for (int d = B - 1; d > 0; d--)
                    hist[d] = hist[d - 1];
                hist[0] = acc;
                zp[k0 + u] = acc;
            }
        };
        const auto backward = [&](const double (&lf)[U][B], const double (&di)[U], int k0) {  // rows k0 + U - 1 down to k0
#pragma unroll
            for (int u = U - 1; u >= 0; u--)
            {
                double acc = __dmul_rn(zp[k0 + u], di[u]);
#pragma unroll
                for (int d = B - 1; d >= 0; d--)
                    acc = fma(-lf[u][d], hist[d], acc);  // explicit: both solve kernels round identically
#pragma unroll
                for (int d = B - 1; d > 0; d--)
                    hist[d] = hist[d - 1];
                hist[0] = acc;
                zp[k0 + u] = acc;
            }
        };
        // 16-byte loads: rows 2j and 2j + 1 of one entry (k0 and U are even).  A wavefront may have 63 loads outstanding,
        // whatever their width, and with one wavefront per CU that count times the bytes per load is all the memory
        // parallelism there is: 8-byte loads ran these sweeps at 4 TB/s.
        const auto load_forward = [&](double (&lf)[U][B], int k0) {
            const double* const lfk = lfw + int64_t(k0 >> 1) * (B * 128);
#pragma unroll
            for (int j = 0; j < U / 2; j++)
#pragma unroll
                for (int d = 0; d < B; d++)
                {
                    const double2 v = *reinterpret_cast<const double2*>(lfk + (j * B + d) * 128);
                    lf[2 * j][d] = v.x;
                    lf[2 * j + 1][d] = v.y;
                }
        };
        const auto load_backward = [&](double (&lf)[U][B], double (&di)[U], int k0) {  // entry d of row k0 + u + d + 1
            const double* const lfk = lfw + int64_t(k0 >> 1) * (B * 128);
            const double* const dvk = dvw + int64_t(k0 >> 1) * 128;
#pragma unroll
            for (int j = U / 2 - 1; j >= 0; j--)
            {
                const double2 v = *reinterpret_cast<const double2*>(dvk + j * 128);
                di[2 * j] = v.x;
                di[2 * j + 1] = v.y;
            }
#pragma unroll
            for (int d = 0; d < B; d++)
#pragma unroll
                for (int j = (U + d) / 2; j >= (d + 1) / 2; j--)
                {
                    const double2 v = *reinterpret_cast<const double2*>(lfk + (j * B + d) * 128);
                    if (2 * j - d - 1 >= 0 && 2 * j - d - 1 < U)
                        lf[2 * j - d - 1][d] = v.x;
                    if (2 * j - d >= 0 && 2 * j - d < U)
                        lf[2 * j - d][d] = v.y;
                }
        };
#pragma unroll
        for (int d = 0; d < B; d++)
            hist[d] = 0.0;
        for (int k0 = 0; k0 < mr; k0 += U)
        {
            double lf[U][B];
            load_forward(lf, k0);
            __builtin_amdgcn_sched_barrier(0);  // all loads of the batch are issued before the first use
            forward(lf, k0);
        }
#pragma unroll
        for (int d = 0; d < B; d++)
            hist[d] = 0.0;
        for (int k0 = mr - U; k0 >= 0; k0 -= U)
        {
            double di[U], lf[U][B];
            load_backward(lf, di, k0);
            __builtin_amdgcn_sched_barrier(0);
            backward(lf, di, k0);
        }
    }
    __syncthreads();
    {
        double* dst = y + first * L;
        for (int k = lane; k < m; k += nl)
        {
#pragma unroll 8
            for (int c = 0; c < nch; c++)
                dst[int64_t(c) * L + k] = seg[c * ldl + k];
        }
    }
}

// ---- explicit inverses of the chunk interiors (levels below the top one) ------------------------------------------------
// A lower level has few chunks (C5: 366 at the second level = 6 wavefronts of k_chunk_solve), so its two sequential sweeps are
// pure latency: 116 us for 47 000 rows.  Its interiors are small enough to invert explicitly (m x m each, 48 MB together):
// M_II^{-1} f becomes a batched GEMV that reads every inverse once with all lanes busy.
//
// k_chunk_inverse: lane (p, j) solves M_II(p) c = e_j with the chunk's banded factor and stores c as COLUMN j of the row-major
// m x m block (ld = longest interior) — coalesced over j; the inverse is symmetric, so the GEMV below may read it either way.
template <int B>
__global__ __launch_bounds__(64) void k_chunk_inverse(int64_t N, int64_t L, ChunkStore cs, int64_t ld, const double* __restrict__ Lf,
                                                      const double* __restrict__ Dinv, double* __restrict__ inv)
{
    const int64_t P = cs.P;
    const int b = cs.b;
    const int64_t p = blockIdx.y;
    const int64_t j = int64_t(blockIdx.x) * 64 + threadIdx.x;
    const int64_t row0 = p * L;
    const int64_t m = ((p == P - 1) ? N : (row0 + L - b)) - row0;
    if (j >= m)
        return;
    double* col = inv + size_t(p) * ld * ld + j;  // element k of the column: col[k * ld]
    double hist[B];
#pragma unroll
    for (int d = 0; d < B; d++)
        hist[d] = 0.0;
    for (int64_t k = 0; k < m; k++)  // every lane walks all rows (zeros above row j): the factor loads are wave-uniform
    {
        double acc = (k == j) ? 1.0 : 0.0;
#pragma unroll
        for (int d = B - 1; d >= 0; d--)
            acc -= ((d < b) ? Lf[cs.lf(p, k, d)] : 0.0) * hist[d];
#pragma unroll
        for (int d = B - 1; d > 0; d--)
            hist[d] = hist[d - 1];
        hist[0] = acc;
        col[k * ld] = acc;
    }
#pragma unroll
    for (int d = 0; d < B; d++)
        hist[d] = 0.0;
    for (int64_t k = m - 1; k >= 0; k--)
    {
        double acc = col[k * ld] * Dinv[cs.dinv(p, k)];
#pragma unroll
        for (int d = B - 1; d >= 0; d--)
            acc -= ((d < b && k + d + 1 < m) ? Lf[cs.lf(p, k + d + 1, d)] : 0.0) * hist[d];
#pragma unroll
        for (int d = B - 1; d > 0; d--)
            hist[d] = hist[d - 1];
        hist[0] = acc;
        col[k * ld] = acc;
    }
}

// y_I(p) = inv(p) f_I(p): workgroup (p, column tile of 64), four wavefronts that each take every fourth row k of the block
// (lane j reads inv[k][j]: 512 contiguous bytes per wavefront and row); partial sums meet in LDS in a fixed order.
constexpr int kBlockGemvWaves = 4;
__global__ __launch_bounds__(64 * kBlockGemvWaves) void k_block_gemv(int64_t N, int b, int64_t L, int64_t P, int64_t ld,
                                                                      const double* __restrict__ inv, const double* __restrict__ f,
                                                                      double* __restrict__ y)
{
    __shared__ double fs[256];
    __shared__ double part[kBlockGemvWaves][64];
    const int64_t p = blockIdx.y;
    const int64_t row0 = p * L;
    const int64_t m = ((p == P - 1) ? N : (row0 + L - b)) - row0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t j = int64_t(blockIdx.x) * 64 + lane;
    for (int64_t k = threadIdx.x; k < m; k += 64 * kBlockGemvWaves)
        fs[k] = f[row0 + k];
    __syncthreads();
    const double* blk = inv + size_t(p) * ld * ld;
    double acc = 0.0;
    if (j < m)
    {
        int64_t k = w;
        for (; k + 3 * kBlockGemvWaves < m; k += 4 * kBlockGemvWaves)
        {
            const double a0 = blk[k * ld + j], a1 = blk[(k + kBlockGemvWaves) * ld + j], a2 = blk[(k + 2 * kBlockGemvWaves) * ld + j],
                         a3 = blk[(k + 3 * kBlockGemvWaves) * ld + j];
            acc += a0 * fs[k];
            acc += a1 * fs[k + kBlockGemvWaves];
            acc += a2 * fs[k + 2 * kBlockGemvWaves];
            acc += a3 * fs[k + 3 * kBlockGemvWaves];
        }
        for (; k < m; k += kBlockGemvWaves)
            acc += blk[k * ld + j] * fs[k];
    }
    part[w][lane] = acc;
    __syncthreads();
    if (w == 0 && j < m)
    {
        double s = part[0][lane];
#pragma unroll
        for (int q = 1; q < kBlockGemvWaves; q++)
            s += part[q][lane];
        y[row0 + j] = s;
    }
}

// separator rows: s = p*b + c  <->  global row (p+1)*L - b + c.  g[s] = f[r] - sum over interior neighbours M(r,j) y[j]
// (all 4b operands are loaded before the first product: the kernel is a few thousand threads of pure latency)
template <int B>
__global__ __launch_bounds__(kThreads) void k_sep_rhs(int64_t N, int b, int64_t L, int64_t P, const double* __restrict__ band,
                                                       const double* __restrict__ f, const double* __restrict__ y,
                                                       double* __restrict__ g)
{
    const int64_t s = int64_t(blockIdx.x) * kThreads + threadIdx.x;
    if (s >= (P - 1) * b)
        return;
    const int64_t p = s / b;
    const int64_t sep0 = (p + 1) * L - b, r = sep0 + (s % b);
    double mu[B], yu[B], ml[B], yl[B];
#pragma unroll
    for (int d = 1; d <= B; d++)
    {
        const int64_t ju = r - d;  // above: interior of chunk p unless still inside this separator
        const bool up = d <= b && ju >= 0 && ju < sep0;
        mu[d - 1] = up ? band[r * (b + 1) + d] : 0.0;
        yu[d - 1] = up ? y[ju] : 0.0;
        const int64_t jl = r + d;  // below: interior of chunk p+1 unless still inside this separator
        const bool lo = d <= b && jl < N && jl >= sep0 + b;
        ml[d - 1] = lo ? band[jl * (b + 1) + d] : 0.0;
        yl[d - 1] = lo ? y[jl] : 0.0;
    }
    double acc = f[r];
#pragma unroll
    for (int d = 1; d <= B; d++)
    {
        if (d <= b && r - d >= 0 && r - d < sep0)
            acc -= mu[d - 1] * yu[d - 1];
        if (d <= b && r + d < N && r + d >= sep0 + b)
            acc -= ml[d - 1] * yl[d - 1];
    }
    g[s] = acc;
}

void launch_sep_rhs(hipStream_t stream, int64_t N, int b, int64_t L, int64_t P, const double* band, const double* f, const double* y,
                    double* g)
{
    const dim3 grid(unsigned(((P - 1) * b + kThreads - 1) / kThreads));
    if (b <= 4)
        hipLaunchKernelGGL(k_sep_rhs<4>, grid, dim3(kThreads), 0, stream, N, b, L, P, band, f, y, g);
    else if (b <= 8)
        hipLaunchKernelGGL(k_sep_rhs<8>, grid, dim3(kThreads), 0, stream, N, b, L, P, band, f, y, g);
    else if (b <= 16)
        hipLaunchKernelGGL(k_sep_rhs<16>, grid, dim3(kThreads), 0, stream, N, b, L, P, band, f, y, g);
    else
        hipLaunchKernelGGL(k_sep_rhs<64>, grid, dim3(kThreads), 0, stream, N, b, L, P, band, f, y, g);
}

// x_I = y_I - W [x_S(p-1); x_S(p)], x_S copied into place.  W: 2b columns of N (column c of row r at c*N + r; zero in
// separator rows): one workgroup per chunk, so the 2b separator unknowns are wave-uniform (scalar loads) and every W / y / x
// access is a coalesced line.  (The first version — one thread per row over a row-major W, chunk number by a 64-bit division —
// ran at 3.6 TB/s.)
constexpr int kBackThreads = 128;
constexpr int64_t kBackPiece = 2048;  // rows per workgroup
__global__ __launch_bounds__(kBackThreads) void k_back_subst(int64_t N, int b, int64_t L, int64_t P, const double* __restrict__ W,
                                                             const double* __restrict__ y, const double* __restrict__ xs,
                                                             double* __restrict__ x)
{
    const int64_t p = blockIdx.x;
    const int64_t row0 = p * L;
    const int64_t end = (p == P - 1) ? N : row0 + L;
    const int64_t sep0 = (p == P - 1) ? N : row0 + L - b;
    const double* const xprev = xs + (p - 1) * b;  // p > 0
    const double* const xnext = xs + p * b;        // p < P - 1
    // blockIdx.y: pieces of kBackPiece rows of a long chunk (wide bands: up to 31250 rows in at most 32 chunks — one workgroup per
    // chunk streamed its 2b columns of W through 128 threads: 12 ms at n = 1e6, b = 64)
    const int64_t piece0 = row0 + int64_t(blockIdx.y) * kBackPiece;
    const int64_t piece1 = (piece0 + kBackPiece < end) ? piece0 + kBackPiece : end;
    for (int64_t r = piece0 + threadIdx.x; r < piece1; r += kBackThreads)
    {
        if (r >= sep0)
        {
            x[r] = xnext[r - sep0];
            continue;
        }
        double acc = y[r];
        // eight columns of W in flight per thread (a plain loop over the columns waits for every load in turn: 0.8 ms for the
        // 512 MB of W at n = 1e6, b = 32); the products enter in column order as before
        const auto side = [&](const double* Wc, const double* xv) {
            int c = 0;
            for (; c + 8 <= b; c += 8)
            {
                double w[8];
#pragma unroll
                for (int i = 0; i < 8; i++)
                    w[i] = Wc[int64_t(c + i) * N];
#pragma unroll
                for (int i = 0; i < 8; i++)
                    acc -= w[i] * xv[c + i];
            }
            for (; c < b; c++)
                acc -= Wc[int64_t(c) * N] * xv[c];
        };
        if (p > 0)
            side(W + r, xprev);
        if (p < P - 1)
            side(W + int64_t(b) * N + r, xnext);
        x[r] = acc;
    }
}

// dst = src with sigma subtracted from the diagonal entries (column 0 of the n x (b+1) row-major band)
// (pencil: dst = src - sigma * srcB entry by entry, the bands share one layout)
__global__ __launch_bounds__(kThreads) void k_band_shift(int64_t total, int bw, double sigma, const double* __restrict__ src,
                                                          const double* __restrict__ srcB, double* __restrict__ dst)
{
    for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < total; i += int64_t(gridDim.x) * kThreads)
        dst[i] = src[i] - sigma * (srcB ? srcB[i] : ((i % bw) == 0 ? 1.0 : 0.0));
}

// ---- factorisation of the top level on the device -------------------------------------------------------
// One lane per chunk, the three steps of the host routine below (factor_level) in the same order of operations:
//   1. banded LDL' of the chunk's interior block (the last B rows of L and D kept in registers),
//   2. the 2b spikes  W = M_II^{-1} M_IS  (forward/backward substitution with the factor just written),
//   3. the chunk's (2b x 2b) contribution  M_SI W  to the Schur complement of its two separators.
// band: N x (b+1) row-major, band[i*(b+1)+d] = M(i, i-d).  W (2b columns of N: column c of row r at c*N + r) and C (P x 2b x 2b) must be
// zero on entry.  stats[0] counts boosted pivots (|d| <= tiny replaced by +-tiny), stats[1] is the smallest |pivot|.
template <int B>
__global__ __launch_bounds__(kChunkThreads) void k_chunk_factor(int64_t N, int64_t L, ChunkStore cs, double tiny,
                                                                 const double* __restrict__ band, double* __restrict__ Lf,
                                                                 double* __restrict__ Dinv, double* __restrict__ W,
                                                                 double* __restrict__ C, unsigned long long* __restrict__ stats)
{
    const int64_t P = cs.P;
    const int b = cs.b;
    const int64_t p = int64_t(blockIdx.x) * kChunkThreads + threadIdx.x;
    if (p >= P)
        return;
    const int64_t row0 = p * L;
    const int64_t m = ((p == P - 1) ? N : (row0 + L - b)) - row0;  // interior rows
    const int bw = b + 1;
    // ---- 1. LDL' ---------------------------------------------------------------------------------------
    {
        double minpiv = 1.7976931348623157e308;
        unsigned long long negs = 0;
        double Lw[B][B], Dw[B];  // Lw[i][d] = L(k-1-i, k-1-i-d-1), Dw[i] = D(k-1-i)
#pragma unroll
        for (int i = 0; i < B; i++)
        {
            Dw[i] = 1.0;
#pragma unroll
            for (int d = 0; d < B; d++)
                Lw[i][d] = 0.0;
        }
        for (int64_t k = 0; k < m; k++)
        {
            const int dk = int(k < b ? k : b);
            const double* mrow = band + (row0 + k) * bw;
            double lrow[B];
#pragma unroll
            for (int d = B - 1; d >= 0; d--)
            {
                lrow[d] = 0.0;
                if (d < dk)
                {
                    double v = mrow[d + 1];
#pragma unroll
                    for (int e = B - 1; e > d; e--)
                        if (e < dk)
                            v -= lrow[e] * Dw[e] * Lw[d][e - d - 1];
                    lrow[d] = v / Dw[d];
                }
            }
            double dv = mrow[0];
#pragma unroll
            for (int d = 0; d < B; d++)
                if (d < dk)
                    dv -= lrow[d] * lrow[d] * Dw[d];
            // A pivot that is tiny RELATIVE TO ITS OWN ROW of the matrix (levels below the first mix magnitudes: Schur
            // complements next to an ill-conditioned chunk are huge) is boosted; `tiny` is the relative threshold.
            double rowmax = fabs(mrow[0]);
#pragma unroll
            for (int d = 0; d < B; d++)
                if (d < dk)
                    rowmax = fmax(rowmax, fabs(mrow[d + 1]));
            // stats[1]: smallest |pivot| / row scale seen (bit pattern of a non-negative double orders like an integer)
            const double thr = tiny * rowmax + 1e-300;
            minpiv = fmin(minpiv, rowmax > 0.0 ? fabs(dv) / rowmax : 0.0);
            if (!(fabs(dv) > thr))
            {
                atomicAdd(&stats[0], 1ull);  // boosted pivots
                dv = (dv < 0.0) ? -thr : thr;
            }
            if (dv < 0.0)
                negs++;
            Dinv[cs.dinv(p, k)] = 1.0 / dv;
#pragma unroll
            for (int d = 0; d < B; d++)
                if (d < b)
                    Lf[cs.lf(p, k, d)] = lrow[d];
#pragma unroll
            for (int i = B - 1; i > 0; i--)
            {
                Dw[i] = Dw[i - 1];
#pragma unroll
                for (int d = 0; d < B; d++)
                    Lw[i][d] = Lw[i - 1][d];
            }
            Dw[0] = dv;
#pragma unroll
            for (int d = 0; d < B; d++)
                Lw[0][d] = lrow[d];
        }
        if (minpiv == minpiv)  // a NaN pivot was boosted and counted above
            atomicMin(&stats[1], (unsigned long long) __double_as_longlong(minpiv));
        if (negs)
            atomicAdd(&stats[2], negs);
    }
    if (P == 1)
        return;
    // ---- 2. spikes: side 0 couples to separator p-1 (rows row0-b..row0-1), side 1 to separator p --------
    const int w2 = 2 * b;
    for (int side = 0; side < 2; side++)
    {
        if ((side == 0 && p == 0) || (side == 1 && p == P - 1))
            continue;
        double hist[B][B];  // hist[c][d] = unknown k-d-1 (forward) / k+d+1 (backward) of right-hand side c
#pragma unroll
        for (int c = 0; c < B; c++)
#pragma unroll
            for (int d = 0; d < B; d++)
                hist[c][d] = 0.0;
        for (int64_t k = 0; k < m; k++)
        {
            double lf[B];
#pragma unroll
            for (int d = 0; d < B; d++)
                lf[d] = (d < b) ? Lf[cs.lf(p, k, d)] : 0.0;
            const int dk = int(k < b ? k : b);
#pragma unroll
            for (int c = 0; c < B; c++)
            {
                if (c >= b)
                    continue;
                double acc = 0.0;
                if (side == 0)
                {
                    if (k <= c)
                        acc = band[(row0 + k) * bw + (k + b - c)];
                }
                else if (k >= m + c - b)
                    acc = band[(row0 + m + c) * bw + (m + c - k)];
#pragma unroll
                for (int d = B - 1; d >= 0; d--)
                    if (d < dk)
                        acc -= lf[d] * hist[c][d];
#pragma unroll
                for (int d = B - 1; d > 0; d--)
                    hist[c][d] = hist[c][d - 1];
                hist[c][0] = acc;
                W[int64_t(side * b + c) * N + row0 + k] = acc;
            }
        }
#pragma unroll
        for (int c = 0; c < B; c++)
#pragma unroll
            for (int d = 0; d < B; d++)
                hist[c][d] = 0.0;
        for (int64_t k = m - 1; k >= 0; k--)
        {
            const double di = Dinv[cs.dinv(p, k)];
            double lf[B];
#pragma unroll
            for (int d = 0; d < B; d++)
                lf[d] = (d < b && k + d + 1 < m) ? Lf[cs.lf(p, k + d + 1, d)] : 0.0;
#pragma unroll
            for (int c = 0; c < B; c++)
            {
                if (c >= b)
                    continue;
                double acc = W[int64_t(side * b + c) * N + row0 + k] * di;
#pragma unroll
                for (int d = B - 1; d >= 0; d--)
                    acc -= lf[d] * hist[c][d];
#pragma unroll
                for (int d = B - 1; d > 0; d--)
                    hist[c][d] = hist[c][d - 1];
                hist[c][0] = acc;
                W[int64_t(side * b + c) * N + row0 + k] = acc;
            }
        }
    }
    // ---- 3. C(s1, s2) = sum_k M(separator row s1, interior k) * spike_s2[k] --------------------------------
    double* Cp = C + p * int64_t(w2) * w2;
    for (int side1 = 0; side1 < 2; side1++)
    {
        if ((side1 == 0 && p == 0) || (side1 == 1 && p == P - 1))
            continue;
        for (int c1 = 0; c1 < b; c1++)
            for (int s2 = 0; s2 < w2; s2++)
            {
                const int side2 = s2 / b;
                if ((side2 == 0 && p == 0) || (side2 == 1 && p == P - 1))
                    continue;
                double acc = 0.0;
                if (side1 == 0)
                {
                    const int64_t kend = (b < m) ? b : m;
                    for (int64_t k = 0; k < kend; k++)
                        if (k <= c1)
                            acc += band[(row0 + k) * bw + (k + b - c1)] * W[int64_t(s2) * N + row0 + k];
                }
                else
                {
                    for (int64_t k = (m - b > 0 ? m - b : 0); k < m; k++)
                        if (k >= m + c1 - b)
                            acc += band[(row0 + m + c1) * bw + (m + c1 - k)] * W[int64_t(s2) * N + row0 + k];
                }
                Cp[(side1 * b + c1) * w2 + s2] = acc;
            }
    }
}

// column-major host inverse -> row-major device copy
void upload_row_major(const std::vector<double>& inv, int64_t n, DevBuf<double>& dst)
{
    std::vector<double> rm(inv.size());
    // (by the host's cores: at n = 1825 — C5's last level — the serial transposition was 10 of set_shift's 65 ms)
    parallel_ranges(n, n >= 512 ? ingest_threads() : 1, [&](int, int64_t r0, int64_t r1) {
        for (int64_t c = 0; c < n; c++)
            for (int64_t r = r0; r < r1; r++)
                rm[size_t(r) * n + c] = inv[size_t(c) * n + r];
    });
    dst.alloc(rm.size());
    MISPEC_HIP(hipMemcpy(dst.p, rm.data(), rm.size() * sizeof(double), hipMemcpyHostToDevice));
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// one level of the recursive factorisation
// ---------------------------------------------------------------------------------------------------
namespace {
// Explicit inverse (column-major) of a symmetric band matrix by LU with PARTIAL PIVOTING within the band (row k is
// exchanged with one of the rows k..k+b, U fills to width 2b): the dgbtrf scheme on a row-major window
// R(i, j - i + b), i - b <= j <= i + 2b.  Throws when the matrix is singular to working precision.
void band_lu_inverse(const HostBand& M, std::vector<double>& inv)
{
    const int64_t N = M.n;
    const int b = M.b, w = 3 * b + 1;
    std::vector<double> R(size_t(N) * w, 0.0);
    auto at = [&](int64_t i, int64_t j) -> double& { return R[size_t(i) * w + size_t(j - i + b)]; };
    double scale = 0.0;
    for (int64_t i = 0; i < N; i++)
        for (int64_t j = std::max<int64_t>(i - b, 0); j <= std::min<int64_t>(i + b, N - 1); j++)
        {
            at(i, j) = M.get(i, j);
            scale = std::max(scale, std::fabs(at(i, j)));
        }
    std::vector<int64_t> piv(static_cast<size_t>(N));
    for (int64_t k = 0; k < N; k++)
    {
        const int64_t rmax = std::min<int64_t>(k + b, N - 1), cmax = std::min<int64_t>(k + 2 * b, N - 1);
        int64_t pr = k;
        double best = std::fabs(at(k, k));
        for (int64_t i = k + 1; i <= rmax; i++)
            if (std::fabs(at(i, k)) > best)
            {
                best = std::fabs(at(i, k));
                pr = i;
            }
        if (!(best > 0.0))  // exactly singular (or NaN); anything else is judged by the calibration of the refinement
            throw Error(MISPEC_EINVAL, "SparseSymShiftSolve: factorization failed with the given shift (singular last level: column " +
                                           std::to_string(k) + " of " + std::to_string(N) + ")");
        piv[size_t(k)] = pr;
        if (pr != k)
            for (int64_t j = k; j <= cmax; j++)
                std::swap(at(k, j), at(pr, j));
        const double d = at(k, k);
        for (int64_t i = k + 1; i <= rmax; i++)
        {
            const double l = at(i, k) / d;
            at(i, k) = l;
            if (l != 0.0)
                for (int64_t j = k + 1; j <= cmax; j++)
                    at(i, j) -= l * at(k, j);
        }
    }
    inv.assign(size_t(N) * N, 0.0);
    // the N columns of the inverse are independent solves with the factor above: spread over the host's cores (each column is
    // computed exactly as before — same operations, same order —, so the inverse does not depend on the number of threads).
    // C5's last level (N = 1825, b = 9): 50 of set_shift's 65 ms on one core.
    parallel_ranges(N, N >= 256 ? ingest_threads() : 1, [&](int, int64_t c_begin, int64_t c_end) {
    std::vector<double> e(static_cast<size_t>(N));
    for (int64_t c = c_begin; c < c_end; c++)
    {
        std::fill(e.begin(), e.end(), 0.0);
        e[size_t(c)] = 1.0;
        for (int64_t k = 0; k < N; k++)  // forward sweep with the interchanges interleaved (as dgbtrs)
        {
            if (piv[size_t(k)] != k)
                std::swap(e[size_t(k)], e[size_t(piv[size_t(k)])]);
            const double ek = e[size_t(k)];
            if (ek != 0.0)
                for (int64_t i = k + 1; i <= std::min<int64_t>(k + b, N - 1); i++)
                    e[size_t(i)] -= at(i, k) * ek;
        }
        for (int64_t k = N - 1; k >= 0; k--)
        {
            double acc = e[size_t(k)];
            for (int64_t j = k + 1; j <= std::min<int64_t>(k + 2 * b, N - 1); j++)
                acc -= at(k, j) * e[size_t(j)];
            e[size_t(k)] = acc / at(k, k);
        }
        std::copy(e.begin(), e.end(), inv.begin() + size_t(c) * N);
    }
    });
}

// ---- iterative refinement on the device ---------------------------------------------------------------------
// r = x - (A - sigma B) y from the resident unshifted band(s) (n x (b+1) row-major, band[i*(b+1)+d] = A(i, i-d)); B = I
// when bandB is null.  One thread per row.
__global__ __launch_bounds__(kThreads) void k_band_resid(int64_t n, int b, double sigma, const double* __restrict__ band,
                                                          const double* __restrict__ bandB, const double* __restrict__ x,
                                                          const double* __restrict__ y, double* __restrict__ r)
{
    const int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x;
    if (i >= n)
        return;
    const int bw = b + 1;
    double acc = x[i];
    for (int d = b; d >= 1; d--)
        if (i - d >= 0)
        {
            const double m = band[i * bw + d] - (bandB ? sigma * bandB[i * bw + d] : 0.0);
            acc -= m * y[i - d];
        }
    acc -= (band[i * bw] - sigma * (bandB ? bandB[i * bw] : 1.0)) * y[i];
    for (int d = 1; d <= b; d++)
        if (i + d < n)
        {
            const double m = band[(i + d) * bw + d] - (bandB ? sigma * bandB[(i + d) * bw + d] : 0.0);
            acc -= m * y[i + d];
        }
    r[i] = acc;
}
__global__ __launch_bounds__(kThreads) void k_add_inplace(int64_t n, double* __restrict__ y, const double* __restrict__ dy)
{
    const int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x;
    if (i < n)
        y[i] += dy[i];
}
// out[slot] = max(out[slot], max_i |v_i|) through the integer order of non-negative doubles
__global__ __launch_bounds__(kThreads) void k_absmax(int64_t n, const double* __restrict__ v, unsigned long long* __restrict__ out, int slot)
{
    __shared__ double red[kThreads];
    double m = 0.0;
    for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < n; i += int64_t(gridDim.x) * kThreads)
    {
        const double a = fabs(v[i]);
        m = (a > m || a != a) ? a : m;  // a NaN wins
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1)
    {
        if (int(threadIdx.x) < s)
        {
            const double o = red[threadIdx.x + s];
            if (o > red[threadIdx.x] || o != o)
                red[threadIdx.x] = o;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        const double r = red[0];
        atomicMax(&out[slot], r == r ? (unsigned long long) __double_as_longlong(r) : 0x7ff8000000000000ull);
    }
}
// deterministic probe right-hand side in (-0.5, 0.5) (splitmix64 of the index)
__global__ __launch_bounds__(kThreads) void k_probe_fill(int64_t n, double* __restrict__ v)
{
    const int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x;
    if (i >= n)
        return;
    unsigned long long z = (unsigned long long) i + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    v[i] = double(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
}
}  // namespace

// What the factorisation saw (all levels): decides whether set_shift() has to calibrate iterative refinement.
struct FactorStats
{
    long long boosts = 0;          // pivots replaced by +-sqrt(eps)*scale
    double min_pivot_ratio = 1.0;  // smallest |pivot| / scale of its level
    long long negative = 0;        // negative pivots (a positive definite matrix has none at any level)
    bool want_cholesky = false;    // also keep the triangular factor of the last level (G G' = M for SparseCholesky)
};

// option shift=profile=1: the phases of set_shift with their host wall time, on stderr (tools/bench_configs.py c5 reads them)
struct PhaseTimer
{
    const char* what;
    int64_t n;
    int b;
    bool on;
    std::chrono::steady_clock::time_point t0;
    PhaseTimer(const char* w, int64_t n_, int b_);
    ~PhaseTimer()
    {
        if (on)
            std::fprintf(stderr, "[mispec set_shift] %-28s N %9lld b %3d  %9.3f ms\n", what, (long long) n, b,
                         1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
};

struct mispec::BandLevel
{
    int64_t N = 0, L = 0, P = 1;
    int b = 0;
    ChunkStore cs;  // layout of Lf / Dinv
    DevBuf<double> Lf, Dinv, W, band, y, g, xs;
    DevBuf<double> LfT;  // wide levels (cs.wide): LT(k, d) = L(k + d + 1, d), the forward sweep's coefficients (k_shift_factor)
    DevBuf<double> inv;  // last level only: explicit inverse (N x N), applied by a dense GEMV
    DevBuf<double> binv;  // partitioned level with few chunks: explicit inverses of the interiors, P blocks of binv_ld x binv_ld
    int64_t binv_ld = 0;
    DevBuf<double> linv, linvt;  // last level, Cholesky use only: C^{-1} and C^{-T} of the level's M = C C' (row-major)
    std::unique_ptr<BandLevel> next;
};

namespace {

// defaults of ChunkPlan below: 128 rows per chunk; a level of at most 2048 rows is not partitioned any further: its inverse
// is formed explicitly (band LU solves of the unit vectors) and applied as a GEMV

void throw_singular()
{
    throw Error(MISPEC_EINVAL, "SparseSymShiftSolve: factorization failed with the given shift");
}

// Chunk-length bias of the factorisation attempt in progress (set_shift): when an attempt meets chunk interiors that are
// singular although the matrix is not — tridiag(-1, 0, -1) cut into odd pieces is the textbook case — the next attempt
// moves every separator by using chunks one row longer.
thread_local int g_chunk_bias = 0;
// Last resort of set_shift for a matrix of at most kPivotedLimit rows whose chunk interiors stay singular wherever the
// separators are put (tridiag(-1, 0, -1) is one): no partition at all, one band LU with partial pivoting + explicit inverse.
thread_local bool g_single_chunk = false;
constexpr int64_t kPivotedLimit = 8192;

// chunk length and chunk count of a level: 128-row chunks, levels of at most 2048 rows inverted densely.  The solve kernels run
// one lane per chunk, so a level needs many chunks to pull bandwidth: short chunks, paid for with more separator rows (the next
// level).  Other lengths measured slower (profiles/rounds_1_2/r03j_*: 0.137 / 0.182 / 0.205 ms per solve at 160 / 192 / 256 rows, 0.139 at 96).
struct ChunkPlan
{
    int64_t big = 128, small = 128, n_switch = 0, n_dense = 2048;
};
const ChunkPlan& chunk_plan()
{
    static const ChunkPlan plan = [] {
        ChunkPlan c;
        return c;
    }();
    return plan;
}

// A top level wider than kNarrowBandwidth (round 4: half-bandwidths of up to 64 beyond the dense limit) plans longer chunks at
// every level — 32 b rows instead of 128 — so that a level keeps 1/32 of its rows as separators whatever b is and the chain
// n -> n/32 -> n/1024 ... reaches the dense last level before the band (2b - 1 per level) outgrows the chunk kernels.
thread_local bool g_wide_top = false;

void plan_level(int64_t N, int b, int64_t& L, int64_t& P)
{
    const ChunkPlan& cp = chunk_plan();
    L = std::max<int64_t>((N > cp.n_switch ? cp.big : cp.small) + g_chunk_bias, 4 * int64_t(b));
    const bool last = N <= std::max<int64_t>(cp.n_dense, 8 * int64_t(b)) || (g_single_chunk && N <= kPivotedLimit);
    if (g_wide_top)
    {
        L = std::max<int64_t>(L, 32 * int64_t(b) + g_chunk_bias);
        if (!last && N / L < 2)
            L = N / 2;  // too large for the dense level, too small for two long chunks: two chunks of half the rows
    }
    P = last ? 1 : N / L;
    // The Schur complement of a level of half-bandwidth b has half-bandwidth 2b - 1.  Where that is more than the chunk kernels
    // take, the next level has to be the last (dense) one: at most n_dense separator rows, i.e. fewer and longer chunks.
    if (P > 1 && 2 * int64_t(b) - 1 > kMaxBandwidth)
    {
        const int64_t pmax = cp.n_dense / std::max(b, 1) + 1;
        if (P > pmax)
        {
            // (the retry bias of mispec_symshift_set_shift — "separators moved by one row each attempt" — has to survive the
            // clamp, or every attempt factors the identical partition: ADVICE r04)
            L = (N + pmax - 1) / pmax + g_chunk_bias;
            P = std::max<int64_t>(1, N / L);
        }
    }
    if (P < 2)
    {
        P = 1;
        L = N;
    }
}

// MISPEC_SHIFT="key=value,...": ONE test hook for the kernel variants of the banded solve, so that tests can require that they
// agree (tests/test_gpu_shift.py): lds=0 (always the general sweep kernel), batch=8|16|32 (rows per batch of the staged sweeps),
// lanes=8|16|32|64 (chunks per wavefront), block_inverse=<MiB> (size limit of the explicit chunk inverses; 0: sweeps everywhere),
// factor=host (the top level factored on the host).  Not a tuning interface: the defaults are the measured best.
long long shift_option(const char* key, long long dflt)
{
    const char* spec_c = option("shift");
    const std::string spec = spec_c ? spec_c : "";
    const std::string k = std::string(key) + "=";
    size_t pos = 0;
    while (pos < spec.size())
    {
        size_t end = spec.find(',', pos);
        if (end == std::string::npos)
            end = spec.size();
        if (spec.compare(pos, k.size(), k) == 0)
        {
            const std::string v = spec.substr(pos + k.size(), end - pos - k.size());
            if (v == "host")
                return 1;
            if (v == "device")
                return 0;
            return atoll(v.c_str());
        }
        pos = end + 1;
    }
    return dflt;
}

}  // namespace
PhaseTimer::PhaseTimer(const char* w, int64_t n_, int b_) : what(w), n(n_), b(b_), on(shift_option("profile", 0) != 0)
{
    if (on)
        t0 = std::chrono::steady_clock::now();
}
namespace {

// whether a level of this shape is factored by k_chunk_factor (MISPEC_SHIFT=factor=host keeps everything on the host)
bool factored_on_device(int64_t N, int b)
{
    const bool host_only = shift_option("factor", 0) != 0;
    int64_t L, P;
    plan_level(N, b, L, P);
    return P > 1 && b <= 8 && !host_only;
}

// Whether the chunk interiors of a partitioned level are also inverted explicitly (k_chunk_inverse / k_block_gemv): levels
// whose chunks would not fill the device with one lane each, as long as the blocks stay small next to the matrix
// (MISPEC_SHIFT=block_inverse=0 keeps the sweeps; =<MiB> moves the size limit, default 256 MiB per level).
bool wants_block_inverse(int64_t P, int64_t mmax, int b)
{
    const long long limit_mib = shift_option("block_inverse", 256);
    if (limit_mib <= 0 || mmax > 256 || b > 16)
        return false;
    const double bytes = double(P) * double(mmax) * double(mmax) * 8.0;
    return P <= 65535 && bytes <= double(limit_mib) * 1048576.0;  // P: grid.y of the two kernels
}

void build_block_inverses(const mispec_ctx& ctx, BandLevel& lev, int64_t mmax)
{
    lev.binv_ld = mmax;
    lev.binv.alloc(size_t(lev.P) * mmax * mmax);
    MISPEC_HIP(hipMemsetAsync(lev.binv.p, 0, lev.binv.n * sizeof(double), ctx.stream));
    const dim3 grid(unsigned((mmax + 63) / 64), unsigned(lev.P));
    if (lev.b <= 4)
        hipLaunchKernelGGL((k_chunk_inverse<4>), grid, dim3(64), 0, ctx.stream, lev.N, lev.L, lev.cs, mmax, lev.Lf.p, lev.Dinv.p,
                           lev.binv.p);
    else if (lev.b <= 8)
        hipLaunchKernelGGL((k_chunk_inverse<8>), grid, dim3(64), 0, ctx.stream, lev.N, lev.L, lev.cs, mmax, lev.Lf.p, lev.Dinv.p,
                           lev.binv.p);
    else
        hipLaunchKernelGGL((k_chunk_inverse<16>), grid, dim3(64), 0, ctx.stream, lev.N, lev.L, lev.cs, mmax, lev.Lf.p, lev.Dinv.p,
                           lev.binv.p);
    MISPEC_HIP(hipGetLastError());
}

// Factor the band matrix M (destroyed) into `lev`, recursively.
void factor_level(mispec_ctx* ctx, HostBand& M, BandLevel& lev, FactorStats& stats)
{
    const int64_t N = M.n;
    const int b = M.b;
    lev.N = N;
    lev.b = b;
    int64_t L, P;
    plan_level(N, b, L, P);
    // the last level (P == 1) is a band LU with an explicit inverse: any width
    MISPEC_REQUIRE(P == 1 || b <= kMaxBandwidth, "internal: band wider than the chunk kernel supports");
    lev.L = L;
    lev.P = P;
    const int64_t mmax = (P == 1) ? N : std::max<int64_t>(L - b, N - (P - 1) * L);  // longest interior
    // pivots at or below tiny * (largest entry of their matrix row) are boosted to that magnitude: see the header comment
    const double tiny = 1.4901161193847656e-08;  // sqrt(eps)

    const bool on_device = factored_on_device(N, b);
    MISPEC_REQUIRE(on_device || !M.view, "internal: a band view is only valid for a level factored on the device");
    // host-factored levels of half-bandwidth 9...64 whose interiors get no explicit inverse: the row-major layout of the
    // wave-per-chunk solve (option shift=wave=0: the lane-per-chunk kernels of rounds 4-5 on the interleaved layout)
    const bool wide = !on_device && b > 8 && P > 1 && shift_option("wave", 1) != 0 &&
                      !(!stats.want_cholesky && wants_block_inverse(P, mmax, b));
    const ChunkStore cs = make_chunk_store(N, b, L, P, wide);
    lev.cs = cs;
    const size_t lf_size = cs.lf_size(), dinv_size = cs.dinv_size(), w_size = size_t(N) * 2 * std::max(b, 1);
    // host images of the factor (host path only).  Hundreds of MB at the top level of a wide band (W: 512 MB at n = 1e6, b = 32):
    // allocated without initialisation and zeroed — first touched — by the host's cores (one core took 0.15 s for it)
    struct HostImage
    {
        std::unique_ptr<double[]> p;
        size_t n = 0;
        void zeros(size_t count)
        {
            n = count;
            p.reset(new double[std::max<size_t>(count, 1)]);
            double* q = p.get();
            parallel_ranges(int64_t(count), count >= (size_t(1) << 22) ? ingest_threads() : 1,
                            [q](int, int64_t b0, int64_t b1) { std::fill(q + b0, q + b1, 0.0); });
        }
        double& operator[](size_t i) { return p[i]; }
    };
    HostImage Lf, Dinv, W;
    if (!on_device)
    {
        PhaseTimer pt("level: host images zeroed", N, b);
        Lf.zeros(lf_size);
        Dinv.zeros(dinv_size);
        W.zeros(w_size);
    }
    const int64_t nsep = (P - 1) * b;
    HostBand S;  // Schur complement of the separators
    S.n = nsep;
    S.b = (P > 1) ? std::min<int64_t>(2 * b - 1, std::max<int64_t>(nsep - 1, 0)) : 0;
    S.a.assign(size_t(std::max<int64_t>(nsep, 1)) * (S.b + 1), 0.0);

    // per-chunk (2b x 2b) contributions M_SI W to the Schur complement, [P][2b][2b] — written by the device kernel or by the host
    // threads below, assembled afterwards in chunk order (one order of additions whoever produced the blocks)
    std::vector<double> Cc;
    // ---- the top level of a large matrix is factored on the device (one lane per chunk, k_chunk_factor); the
    // ---- Schur complement comes back as per-chunk blocks and is assembled below, in the order of the host loop
    if (on_device)
    {
        PhaseTimer pt("level: device factor", N, b);
        ctx->make_current();
        const int w2 = 2 * b;
        lev.band.alloc(size_t(N) * (b + 1));
        if (M.view_dev)
        {
            const int64_t total = N * (b + 1);
            hipLaunchKernelGGL(k_band_shift, dim3(unsigned(std::min<int64_t>((total + kThreads - 1) / kThreads, 4096))), dim3(kThreads),
                               0, ctx->stream, total, b + 1, M.shift, M.view_dev, M.viewB_dev, lev.band.p);
            MISPEC_HIP(hipGetLastError());
        }
        else
            MISPEC_HIP(hipMemcpyAsync(lev.band.p, M.a.data(), M.a.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        lev.Lf.alloc(lf_size);
        lev.Dinv.alloc(dinv_size);
        lev.W.alloc(w_size);
        DevBuf<double> Cdev;
        Cdev.alloc(size_t(P) * w2 * w2);
        DevBuf<unsigned long long> dstats;
        dstats.alloc(3);
        MISPEC_HIP(hipMemsetAsync(lev.Lf.p, 0, lf_size * sizeof(double), ctx->stream));
        MISPEC_HIP(hipMemsetAsync(lev.Dinv.p, 0, dinv_size * sizeof(double), ctx->stream));
        MISPEC_HIP(hipMemsetAsync(lev.W.p, 0, w_size * sizeof(double), ctx->stream));
        MISPEC_HIP(hipMemsetAsync(Cdev.p, 0, Cdev.n * sizeof(double), ctx->stream));
        {
            const double huge = 1.7976931348623157e308;
            unsigned long long init[3] = {0ull, 0ull, 0ull};
            std::memcpy(&init[1], &huge, sizeof(double));
            MISPEC_HIP(hipMemcpyAsync(dstats.p, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
            MISPEC_HIP(hipStreamSynchronize(ctx->stream));  // init is a local
        }
        const dim3 grid(unsigned((P + kChunkThreads - 1) / kChunkThreads));
        if (b <= 4)
            hipLaunchKernelGGL((k_chunk_factor<4>), grid, dim3(kChunkThreads), 0, ctx->stream, N, L, cs, tiny, lev.band.p,
                               lev.Lf.p, lev.Dinv.p, lev.W.p, Cdev.p, dstats.p);
        else
            hipLaunchKernelGGL((k_chunk_factor<8>), grid, dim3(kChunkThreads), 0, ctx->stream, N, L, cs, tiny, lev.band.p,
                               lev.Lf.p, lev.Dinv.p, lev.W.p, Cdev.p, dstats.p);
        MISPEC_HIP(hipGetLastError());
        Cc.resize(Cdev.n);
        unsigned long long hstats[3] = {0ull, 0ull, 0ull};
        MISPEC_HIP(hipMemcpyAsync(Cc.data(), Cdev.p, Cc.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        MISPEC_HIP(hipMemcpyAsync(hstats, dstats.p, sizeof(hstats), hipMemcpyDeviceToHost, ctx->stream));
        MISPEC_HIP(hipStreamSynchronize(ctx->stream));
        {
            double minpiv;
            std::memcpy(&minpiv, &hstats[1], sizeof(double));
            stats.boosts += (long long) hstats[0];
            stats.negative += (long long) hstats[2];
            stats.min_pivot_ratio = std::min(stats.min_pivot_ratio, minpiv);
        }
    }

    if (P == 1)
    {
        // last level: band LU with partial pivoting (this is where an indefinite matrix needs it most: the Schur
        // complement of all separators), explicit inverse applied as a GEMV
        PhaseTimer pt("last level: inverse + upload", N, b);
        std::vector<double> inv;
        band_lu_inverse(M, inv);
        ctx->make_current();
        upload_row_major(inv, N, lev.inv);
        if (stats.want_cholesky)
        {
            // M = C C' (banded Cholesky), C^{-1} by forward substitution on the unit vectors; both C^{-1} and C^{-T} row-major
            std::vector<double> Cb(size_t(N) * (b + 1), 0.0);  // Cb[i*(b+1)+d] = C(i, i-d)
            bool spd = true;
            for (int64_t i = 0; i < N && spd; i++)
                for (int d = std::min<int64_t>(b, i); d >= 0; d--)
                {
                    const int64_t j = i - d;
                    double v = static_cast<const HostBand&>(M).at(i, d);
                    for (int64_t t = std::max<int64_t>(std::max<int64_t>(i - b, j - b), 0); t < j; t++)
                        v -= Cb[size_t(i) * (b + 1) + (i - t)] * Cb[size_t(j) * (b + 1) + (j - t)];
                    if (d == 0)
                    {
                        if (!(v > 0.0))
                        {
                            spd = false;
                            break;
                        }
                        Cb[size_t(i) * (b + 1)] = std::sqrt(v);
                    }
                    else
                        Cb[size_t(i) * (b + 1) + d] = v / Cb[size_t(j) * (b + 1)];
                }
            if (!spd)
                stats.negative++;
            else
            {
                std::vector<double> X(size_t(N) * N, 0.0), Xt(size_t(N) * N, 0.0), col(static_cast<size_t>(N));
                for (int64_t c = 0; c < N; c++)
                {
                    std::fill(col.begin(), col.end(), 0.0);
                    for (int64_t i = c; i < N; i++)
                    {
                        double acc = (i == c) ? 1.0 : 0.0;
                        for (int64_t t = std::max<int64_t>(std::max<int64_t>(i - b, c), 0); t < i; t++)
                            acc -= Cb[size_t(i) * (b + 1) + (i - t)] * col[size_t(t)];
                        col[size_t(i)] = acc / Cb[size_t(i) * (b + 1)];
                        X[size_t(i) * N + c] = col[size_t(i)];
                        Xt[size_t(c) * N + i] = col[size_t(i)];
                    }
                }
                lev.linv.alloc(X.size());
                lev.linvt.alloc(Xt.size());
                MISPEC_HIP(hipMemcpy(lev.linv.p, X.data(), X.size() * sizeof(double), hipMemcpyHostToDevice));
                MISPEC_HIP(hipMemcpy(lev.linvt.p, Xt.data(), Xt.size() * sizeof(double), hipMemcpyHostToDevice));
            }
        }
        lev.y.alloc(size_t(N));
        MISPEC_HIP(hipStreamSynchronize(ctx->stream));
        M.a.clear();
        M.a.shrink_to_fit();
        return;
    }
    // ---- host path (half-bandwidth 9...64, and every lower level): the chunks are independent — interior LDL', spikes and the
    // ---- chunk's Schur block only read M and write their own slices of Lf / Dinv / W / Cc — so they are spread over the host's
    // ---- cores (parallel_ranges, as the ingest is); the pivot statistics are sums and a minimum, the Schur complement is
    // ---- assembled from the blocks below: the result does not depend on the number of threads.
    const int w2h = 2 * b;
    std::unique_ptr<PhaseTimer> pt_host(on_device ? nullptr : new PhaseTimer("level: host factor (threads)", N, b));
    if (!on_device)
        Cc.assign(size_t(P) * w2h * w2h, 0.0);
    const int parts = on_device ? 1 : int(std::min<int64_t>(std::max<int64_t>(1, (N * int64_t(b + 1) * (b + 1)) / 2000000), ingest_threads()));
    std::vector<FactorStats> part_stats((size_t) parts);
    const HostBand& Mc = M;
    parallel_ranges(on_device ? 0 : P, parts, [&](int part, int64_t p_begin, int64_t p_end) {
    FactorStats& pst = part_stats[size_t(part)];
    std::vector<double> D, Lc, rhs;
    for (int64_t p = p_begin; p < p_end; p++)
    {
        const int64_t row0 = p * L;
        const int64_t m = ((p == P - 1) ? N : (row0 + L - b)) - row0;
        // ---- banded LDL' of the interior block (no pivoting) --------------------------------------
        D.assign(size_t(m), 0.0);
        Lc.assign(size_t(m) * std::max(b, 1), 0.0);  // Lc[k*b + d] = L(k, k-d-1)
        for (int64_t k = 0; k < m; k++)
        {
            const int dk = int(std::min<int64_t>(k, b));
            // row k of L: for j = k-dk .. k-1
            for (int d = dk - 1; d >= 0; d--)
            {
                const int64_t j = k - d - 1;
                double v = Mc.at(row0 + k, d + 1);
                // subtract sum_{t<j} L(k,t) D_t L(j,t), t within both bands
                const int64_t tlo = std::max<int64_t>(std::max<int64_t>(k - b, j - b), 0);
                for (int64_t t = tlo; t < j; t++)
                    v -= Lc[size_t(k) * b + (k - t - 1)] * D[size_t(t)] * Lc[size_t(j) * b + (j - t - 1)];
                Lc[size_t(k) * b + d] = v / D[size_t(j)];
            }
            double dv = Mc.at(row0 + k, 0);
            for (int d = 0; d < dk; d++)
                dv -= Lc[size_t(k) * b + d] * Lc[size_t(k) * b + d] * D[size_t(k - d - 1)];
            double rowmax = std::fabs(Mc.at(row0 + k, 0));
            for (int d = 0; d < dk; d++)
                rowmax = std::max(rowmax, std::fabs(Mc.at(row0 + k, d + 1)));
            const double thr = tiny * rowmax + 1e-300;
            pst.min_pivot_ratio = std::min(pst.min_pivot_ratio, rowmax > 0.0 ? std::fabs(dv) / rowmax : 0.0);
            if (!(std::fabs(dv) > thr))
            {
                pst.boosts++;
                dv = (dv < 0.0) ? -thr : thr;
            }
            if (dv < 0.0)
                pst.negative++;
            D[size_t(k)] = dv;
        }
        for (int64_t k = 0; k < m; k++)
        {
            Dinv[cs.dinv(p, k)] = 1.0 / D[size_t(k)];
            for (int d = 0; d < b; d++)
                Lf[cs.lf(p, k, d)] = Lc[size_t(k) * b + d];
        }
        // ---- spikes W = M_II^{-1} M_IS and their contribution to the Schur complement -------------------
        auto solve_block = [&](std::vector<double>& v) {
            for (int64_t k = 0; k < m; k++)
            {
                double acc = v[size_t(k)];
                const int dk = int(std::min<int64_t>(k, b));
                for (int d = dk - 1; d >= 0; d--)
                    acc -= Lc[size_t(k) * b + d] * v[size_t(k - d - 1)];
                v[size_t(k)] = acc;
            }
            for (int64_t k = m - 1; k >= 0; k--)
            {
                double acc = v[size_t(k)] / D[size_t(k)];
                const int dk = int(std::min<int64_t>(m - 1 - k, b));
                for (int d = dk - 1; d >= 0; d--)
                    acc -= Lc[size_t(k + d + 1) * b + d] * v[size_t(k + d + 1)];
                v[size_t(k)] = acc;
            }
        };
        // side 0: separator p-1 (rows row0-b .. row0-1); side 1: separator p (rows row0+m .. row0+m+b-1)
        std::vector<std::vector<double>> spike(size_t(2 * b));
        for (int side = 0; side < 2; side++)
        {
            if ((side == 0 && p == 0) || (side == 1 && p == P - 1))
                continue;
            for (int c = 0; c < b; c++)
            {
                const int64_t sr = (side == 0) ? (row0 - b + c) : (row0 + m + c);
                rhs.assign(size_t(m), 0.0);
                bool any = false;
                for (int64_t k = (side == 0 ? 0 : std::max<int64_t>(m - b, 0)); k < (side == 0 ? std::min<int64_t>(b, m) : m); k++)
                {
                    const double e = Mc.get(row0 + k, sr);
                    rhs[size_t(k)] = e;
                    any = any || (e != 0.0);
                }
                if (any)
                    solve_block(rhs);
                spike[size_t(side * b + c)] = rhs;
                for (int64_t k = 0; k < m; k++)
                    W[size_t(side * b + c) * N + size_t(row0 + k)] = rhs[size_t(k)];
            }
        }
        // S(s1, s2) -= sum_k M(sep row s1, interior k) * spike_s2[k]
        for (int side1 = 0; side1 < 2; side1++)
        {
            if ((side1 == 0 && p == 0) || (side1 == 1 && p == P - 1))
                continue;
            for (int c1 = 0; c1 < b; c1++)
            {
                const int64_t sr1 = (side1 == 0) ? (row0 - b + c1) : (row0 + m + c1);
                const int64_t s1 = (side1 == 0 ? (p - 1) : p) * b + c1;
                for (int side2 = 0; side2 < 2; side2++)
                {
                    if ((side2 == 0 && p == 0) || (side2 == 1 && p == P - 1))
                        continue;
                    for (int c2 = 0; c2 < b; c2++)
                    {
                        const int64_t s2 = (side2 == 0 ? (p - 1) : p) * b + c2;
                        if (s2 > s1)
                            continue;  // lower triangle only
                        const std::vector<double>& sp = spike[size_t(side2 * b + c2)];
                        double acc = 0.0;
                        for (int64_t k = (side1 == 0 ? 0 : std::max<int64_t>(m - b, 0)); k < (side1 == 0 ? std::min<int64_t>(b, m) : m);
                             k++)
                            acc += Mc.get(row0 + k, sr1) * sp[size_t(k)];
                        Cc[(size_t(p) * w2h + size_t(side1 * b + c1)) * w2h + size_t(side2 * b + c2)] = acc;
                    }
                }
            }
        }
    }
    });
    pt_host.reset();
    for (const FactorStats& pst : part_stats)
    {
        stats.boosts += pst.boosts;
        stats.negative += pst.negative;
        stats.min_pivot_ratio = std::min(stats.min_pivot_ratio, pst.min_pivot_ratio);
    }
    // ---- Schur complement of the separators: S -= (block of chunk p), chunk by chunk, entry by entry in this fixed order
    if (P > 1)
    for (int64_t p = 0; p < P; p++)
        for (int side1 = 0; side1 < 2; side1++)
        {
            if ((side1 == 0 && p == 0) || (side1 == 1 && p == P - 1))
                continue;
            for (int c1 = 0; c1 < b; c1++)
            {
                const int64_t s1 = (side1 == 0 ? (p - 1) : p) * b + c1;
                for (int side2 = 0; side2 < 2; side2++)
                {
                    if ((side2 == 0 && p == 0) || (side2 == 1 && p == P - 1))
                        continue;
                    for (int c2 = 0; c2 < b; c2++)
                    {
                        const int64_t s2 = (side2 == 0 ? (p - 1) : p) * b + c2;
                        if (s2 > s1)
                            continue;  // lower triangle only
                        const double acc = Cc[(size_t(p) * w2h + size_t(side1 * b + c1)) * w2h + size_t(side2 * b + c2)];
                        if (acc != 0.0)
                        {
                            MISPEC_REQUIRE(s1 - s2 <= S.b, "internal: Schur complement wider than expected");
                            S.at(s1, int(s1 - s2)) -= acc;
                        }
                    }
                }
            }
        }
    if (P > 1)
    {
        // + M_SS itself (entries inside one separator; different separators are more than b rows apart)
        for (int64_t p = 0; p < P - 1; p++)
            for (int c1 = 0; c1 < b; c1++)
                for (int c2 = 0; c2 <= c1; c2++)
                    S.at(p * b + c1, c1 - c2) += static_cast<const HostBand&>(M).at((p + 1) * L - b + c1, c1 - c2);
    }

    // ---- upload this level ---------------------------------------------------------------------------
    ctx->make_current();
    auto up = [&](DevBuf<double>& dst, const double* src, size_t count) {
        dst.alloc(std::max<size_t>(count, 1));
        if (count)
            MISPEC_HIP(hipMemcpyAsync(dst.p, src, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    };
    std::unique_ptr<PhaseTimer> pt_up(on_device ? nullptr : new PhaseTimer("level: factor uploaded", N, b));
    if (!on_device)
    {
        up(lev.Lf, Lf.p.get(), Lf.n);
        up(lev.Dinv, Dinv.p.get(), Dinv.n);
        if (cs.wide)
        {
            lev.LfT.alloc(std::max<size_t>(Lf.n, 1));
            MISPEC_HIP(hipMemsetAsync(lev.LfT.p, 0, lev.LfT.n * sizeof(double), ctx->stream));  // (the padding rows of the layout)
            const int64_t per_chunk = std::max<int64_t>(L - b, N - (P - 1) * L) * b;
            hipLaunchKernelGGL(k_shift_factor, dim3(unsigned((per_chunk + 255) / 256), unsigned(P)), dim3(256), 0, ctx->stream, N, L, cs,
                               lev.Lf.p, lev.LfT.p);
            MISPEC_HIP(hipGetLastError());
        }
    }
    lev.y.alloc(size_t(N));
    if (P > 1)
    {
        if (!on_device)
        {
            up(lev.W, W.p.get(), W.n);
            up(lev.band, M.a.data(), M.a.size());
        }
        lev.g.alloc(size_t(nsep));
        lev.xs.alloc(size_t(nsep));
    }
    if (pt_up)
    {
        MISPEC_HIP(hipStreamSynchronize(ctx->stream));  // (the host images go out of scope at the end of this function anyway)
        pt_up.reset();
    }
    {
        PhaseTimer pt("level: uploads + block inverses", N, b);
        if (P > 1 && !stats.want_cholesky && wants_block_inverse(P, mmax, b))
            build_block_inverses(*ctx, lev, mmax);
        MISPEC_HIP(hipStreamSynchronize(ctx->stream));
    }
    M.a.clear();
    M.a.shrink_to_fit();
    if (P > 1)
    {
        lev.next = std::make_unique<BandLevel>();
        factor_level(ctx, S, *lev.next, stats);
    }
}

// Chunks per wavefront of the solve kernels (test hook MISPEC_SHIFT=lanes=64|32|16|8).  One lane per chunk makes a level with P
// chunks run P / 64 wavefronts — 244 at the top level of C5, 6 at the second; fewer chunks per wavefront (the other lanes idle)
// means more wavefronts in flight, but measured (profiles/rounds_1_2/r03a_*) 0.265 / 0.276 / 0.278 / 0.464 ms per solve at 64 / 32 / 16 / 8:
// the sweeps are bound by their dependency chain, not by the number of wavefronts.  A software-pipelined variant (next batch of
// rows in flight during the recurrence) measured 0.281 against 0.265 ms (profiles/rounds_1_2/r03b_*) and was removed again.
int solve_lanes(int64_t P)
{
    const int knob = int(shift_option("lanes", 0));
    if (knob == 64 || knob == 32 || knob == 16 || knob == 8)
        return knob;
    (void) P;
    return kChunkThreads;
}

void launch_chunk_solve(const mispec_ctx& ctx, const BandLevel& lev, dim3 grid_unused, const double* f, double* y, int mode = 0,
                        double* u_out = nullptr)
{
    (void) grid_unused;
    if (lev.cs.wide)  // half-bandwidth 9...64 on the row-major layout: one wavefront per chunk
    {
        hipLaunchKernelGGL(k_chunk_solve_wave, dim3(unsigned((lev.P + kWaveChunks - 1) / kWaveChunks)), dim3(64 * kWaveChunks), 0, ctx.stream,
                           lev.N, lev.L, lev.cs, lev.Lf.p, lev.LfT.p, lev.Dinv.p, f, y, mode, u_out);
        MISPEC_HIP(hipGetLastError());
        return;
    }
    const int lanes = solve_lanes(lev.P);
    const dim3 grid(unsigned((lev.P - 1 + lanes - 1) / lanes) + 1);  // + the last chunk's own workgroup
    const bool chol = mode != 0 || u_out != nullptr;
    // plain solves of a level whose wavefront segments fit the LDS: the staged kernel (MISPEC_SHIFT=lds=0: always the general one)
    const bool lds_off = shift_option("lds", 1) == 0;
    const int batch = int(shift_option("batch", 0));
    if (!chol && !lds_off && lev.P > 1 && lev.b >= 1 && lev.b <= 8)
    {
        const int U = (batch == 8 || batch == 16 || batch == 32) ? batch : (lev.b <= 4 ? 32 : 16);
        const auto whole = [U](int64_t m) { return (m + U - 1) / U * U; };
        // LDS row stride of a chunk: the padded interior; the last chunk's (its workgroup holds it alone) may spread over all rows
        const int64_t ldl = std::max<int64_t>(whole(lev.L - lev.b), (whole(lev.N - (lev.P - 1) * lev.L) + lanes - 1) / lanes) | 1;
        const size_t lds_bytes = size_t(lanes) * size_t(ldl) * sizeof(double);
        if (lds_bytes <= 160 * 1024)
        {
            const auto launch = [&](auto kernel) {
                MISPEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               int(lds_bytes)));
                hipLaunchKernelGGL(kernel, grid, dim3(kStageThreads), lds_bytes, ctx.stream, lev.N, lev.L, lev.cs, int(ldl), lanes, lev.Lf.p,
                                   lev.Dinv.p, f, y);
            };
#define MISPEC_LDS_CASE(B)                       \
    case B:                                      \
        if (U == 32)                             \
            launch(&k_chunk_solve_lds<B, 32>);   \
        else if (U == 16)                        \
            launch(&k_chunk_solve_lds<B, 16>);   \
        else                                     \
            launch(&k_chunk_solve_lds<B, 8>);    \
        break
            switch (lev.b)
            {
                MISPEC_LDS_CASE(1);
                MISPEC_LDS_CASE(2);
                MISPEC_LDS_CASE(3);
                MISPEC_LDS_CASE(4);
                MISPEC_LDS_CASE(5);
                MISPEC_LDS_CASE(6);
                MISPEC_LDS_CASE(7);
                MISPEC_LDS_CASE(8);
            }
#undef MISPEC_LDS_CASE
            MISPEC_HIP(hipGetLastError());
            return;
        }
    }
    (void) chol;
#define MISPEC_CHUNK(B, U) \
    hipLaunchKernelGGL((k_chunk_solve<B, U>), grid, dim3(unsigned(lanes)), 0, ctx.stream, lev.N, lev.L, lev.cs, lev.Lf.p, lev.Dinv.p, f, y, mode, u_out)
    // U rows of factor entries are in flight per lane and batch ((B + 2) * U doubles): as deep as the register file allows
    if (lev.b <= 4)
        MISPEC_CHUNK(4, 32);
    else if (lev.b <= 8)
        MISPEC_CHUNK(8, 16);
    else if (lev.b <= 16)
        MISPEC_CHUNK(16, 8);
    else if (lev.b <= 32)
        MISPEC_CHUNK(32, 4);
    else
        MISPEC_CHUNK(64, 1);
#undef MISPEC_CHUNK
    MISPEC_HIP(hipGetLastError());
}

void solve_level(const mispec_ctx& ctx, const BandLevel& lev, const double* f, double* x)
{
    if (lev.P == 1)
    {
        if (lev.inv.p)
        {
            // explicit inverse, row-major: one wavefront per row (dense.hip) — n waves in flight instead of n/256
            // workgroups that each walk all the columns (the first version: 233 us at n = 1830, a third of a banded solve)
            launch_row_gemv(ctx, lev.inv.p, lev.N, lev.N, lev.N, f, x);
        }
        else
            launch_chunk_solve(ctx, lev, dim3(1), f, x);
        return;
    }
    if (lev.binv.p)
    {
        hipLaunchKernelGGL(k_block_gemv, dim3(unsigned((lev.binv_ld + 63) / 64), unsigned(lev.P)), dim3(64 * kBlockGemvWaves), 0,
                           ctx.stream, lev.N, lev.b, lev.L, lev.P, lev.binv_ld, lev.binv.p, f, lev.y.p);
        MISPEC_HIP(hipGetLastError());
    }
    else
        launch_chunk_solve(ctx, lev, dim3(unsigned((lev.P + kChunkThreads - 1) / kChunkThreads)), f, lev.y.p);
    launch_sep_rhs(ctx.stream, lev.N, lev.b, lev.L, lev.P, lev.band.p, f, lev.y.p, lev.g.p);
    MISPEC_HIP(hipGetLastError());
    solve_level(ctx, *lev.next, lev.g.p, lev.xs.p);
    {
        const int64_t longest = std::max<int64_t>(lev.L, lev.N - (lev.P - 1) * lev.L);
        hipLaunchKernelGGL(k_back_subst, dim3(unsigned(lev.P), unsigned((longest + kBackPiece - 1) / kBackPiece)), dim3(kBackThreads), 0,
                           ctx.stream, lev.N, lev.b, lev.L, lev.P, lev.W.p, lev.y.p, lev.xs.p, x);
    }
    MISPEC_HIP(hipGetLastError());
}

// ---- G^{-1} and G^{-T} for the factor G G' = M of a positive definite band (SparseCholesky beyond the dense limit) -------
// In the nested order (chunk interiors, then separators, recursively) M = G G' with
//   G = [ L_II D^{1/2}  0 ; M_SI L_II^{-T} D^{-1/2}  G_S ],   G_S G_S' = Schur complement of the separators,
// so   u = G^{-1} f :  u_I = D^{-1/2} L_II^{-1} f_I ,  u_S = G_S^{-1} (f_S - M_SI M_II^{-1} f_I)
//      x = G^{-T} u :  x_S = G_S^{-T} u_S ,            x_I = L_II^{-T} D^{-1/2} u_I - W x_S     (W = M_II^{-1} M_IS)
// — the same kernels and factors as the solve, split in two halves.
__global__ __launch_bounds__(kThreads) void k_sep_move(int64_t nsep, int b, int64_t L, const double* __restrict__ src, double* __restrict__ dst,
                                                        int to_rows)
{
    const int64_t s = int64_t(blockIdx.x) * kThreads + threadIdx.x;
    if (s >= nsep)
        return;
    const int64_t row = (s / b + 1) * L - b + (s % b);
    if (to_rows)
        dst[row] = src[s];
    else
        dst[s] = src[row];
}

void chol_forward(const mispec_ctx& ctx, const BandLevel& lev, const double* f, double* u)
{
    const auto blocks = [](int64_t n) { return dim3(unsigned((n + kThreads - 1) / kThreads)); };
    if (lev.P == 1)
    {
        launch_row_gemv(ctx, lev.linv.p, lev.N, lev.N, lev.N, f, u);
        return;
    }
    const int64_t nsep = (lev.P - 1) * lev.b;
    launch_chunk_solve(ctx, lev, dim3(unsigned((lev.P + kChunkThreads - 1) / kChunkThreads)), f, lev.y.p, 0, u);
    launch_sep_rhs(ctx.stream, lev.N, lev.b, lev.L, lev.P, lev.band.p, f, lev.y.p, lev.g.p);
    MISPEC_HIP(hipGetLastError());
    chol_forward(ctx, *lev.next, lev.g.p, lev.xs.p);
    hipLaunchKernelGGL(k_sep_move, blocks(nsep), dim3(kThreads), 0, ctx.stream, nsep, lev.b, lev.L, lev.xs.p, u, 1);
    MISPEC_HIP(hipGetLastError());
}

void chol_backward(const mispec_ctx& ctx, const BandLevel& lev, const double* u, double* x)
{
    const auto blocks = [](int64_t n) { return dim3(unsigned((n + kThreads - 1) / kThreads)); };
    if (lev.P == 1)
    {
        launch_row_gemv(ctx, lev.linvt.p, lev.N, lev.N, lev.N, u, x);
        return;
    }
    const int64_t nsep = (lev.P - 1) * lev.b;
    hipLaunchKernelGGL(k_sep_move, blocks(nsep), dim3(kThreads), 0, ctx.stream, nsep, lev.b, lev.L, u, lev.g.p, 0);
    MISPEC_HIP(hipGetLastError());
    chol_backward(ctx, *lev.next, lev.g.p, lev.xs.p);
    launch_chunk_solve(ctx, lev, dim3(unsigned((lev.P + kChunkThreads - 1) / kChunkThreads)), u, lev.y.p, 2, nullptr);
    {
        const int64_t longest = std::max<int64_t>(lev.L, lev.N - (lev.P - 1) * lev.L);
        hipLaunchKernelGGL(k_back_subst, dim3(unsigned(lev.P), unsigned((longest + kBackPiece - 1) / kBackPiece)), dim3(kBackThreads), 0,
                           ctx.stream, lev.N, lev.b, lev.L, lev.P, lev.W.p, lev.y.p, lev.xs.p, x);
    }
    MISPEC_HIP(hipGetLastError());
}

// dense LU with partial pivoting -> explicit inverse (column-major), host
template <typename T>
void dense_inverse(int n, std::vector<T>& A, std::vector<T>& inv)
{
    std::vector<int> piv(static_cast<size_t>(n));
    auto a = [&](int i, int j) -> T& { return A[size_t(j) * n + i]; };
    for (int k = 0; k < n; k++)
    {
        int pr = k;
        double best = std::abs(a(k, k));
        for (int i = k + 1; i < n; i++)
            if (std::abs(a(i, k)) > best)
            {
                best = std::abs(a(i, k));
                pr = i;
            }
        if (!(best > 0.0))
            throw_singular();
        piv[size_t(k)] = pr;
        if (pr != k)
            for (int j = 0; j < n; j++)
                std::swap(a(k, j), a(pr, j));
        const T d = a(k, k);
        for (int i = k + 1; i < n; i++)
            a(i, k) /= d;
        for (int j = k + 1; j < n; j++)
        {
            const T akj = a(k, j);
            if (akj == T(0))
                continue;
            T* col = &A[size_t(j) * n];
            const T* lk = &A[size_t(k) * n];
            for (int i = k + 1; i < n; i++)
                col[i] -= lk[i] * akj;
        }
    }
    inv.assign(size_t(n) * n, T(0));
    std::vector<T> e(static_cast<size_t>(n));
    for (int c = 0; c < n; c++)
    {
        std::fill(e.begin(), e.end(), T(0));
        e[size_t(c)] = T(1);
        for (int k = 0; k < n; k++)
            std::swap(e[size_t(k)], e[size_t(piv[size_t(k)])]);
        for (int k = 0; k < n; k++)  // L y = P e
        {
            const T ek = e[size_t(k)];
            if (ek != T(0))
                for (int i = k + 1; i < n; i++)
                    e[size_t(i)] -= a(i, k) * ek;
        }
        for (int k = n - 1; k >= 0; k--)  // U x = y
        {
            e[size_t(k)] /= a(k, k);
            const T ek = e[size_t(k)];
            if (ek != T(0))
                for (int i = 0; i < k; i++)
                    e[size_t(i)] -= a(i, k) * ek;
        }
        std::copy(e.begin(), e.end(), inv.begin() + size_t(c) * n);
    }
}

}  // namespace

namespace {
inline dim3 row_blocks(int64_t n) { return dim3(unsigned((n + kThreads - 1) / kThreads)); }

// y += (A - sigma B)^{-1}_approx (x - (A - sigma B) y): one step of iterative refinement with the factorisation at hand
void refine_once(const mispec_symshift& S, const double* x_dev, double* y_dev)
{
    hipStream_t st = S.ctx->stream;
    hipLaunchKernelGGL(k_band_resid, row_blocks(S.n), dim3(kThreads), 0, st, S.n, S.band_b, S.sigma, S.band0_dev.p,
                       S.pencil ? S.bandB0_dev.p : nullptr, x_dev, y_dev, S.ref_r.p);
    MISPEC_HIP(hipGetLastError());
    solve_level(*S.ctx, *S.top, S.ref_r.p, S.ref_dy.p);
    hipLaunchKernelGGL(k_add_inplace, row_blocks(S.n), dim3(kThreads), 0, st, S.n, y_dev, S.ref_dy.p);
    MISPEC_HIP(hipGetLastError());
}

// After a banded factorisation: solve a probe system, measure the backward error
//   omega = |x - M y|_inf / (|M|_inf |y|_inf + |x|_inf)
// and add refinement steps until it reaches rounding level.  A definite matrix passes at once (no step, no cost per
// solve); an indefinite one with boosted / small pivots typically needs 1-2; no convergence = singular shift.
void calibrate_refinement(mispec_symshift& S, const FactorStats& fs)
{
    S.boosted_pivots = fs.boosts;
    S.min_pivot_ratio = fs.min_pivot_ratio;
    S.refine_steps = 0;
    S.probe_backward_error = 0.0;
    const int64_t n = S.n;
    const int bw = S.band_b + 1;
    hipStream_t st = S.ctx->stream;
    if (!S.band0_dev.p)
    {
        S.band0_dev.alloc(S.band0.size());
        MISPEC_HIP(hipMemcpyAsync(S.band0_dev.p, S.band0.data(), S.band0.size() * sizeof(double), hipMemcpyHostToDevice, st));
        if (S.pencil)
        {
            S.bandB0_dev.alloc(S.bandB0.size());
            MISPEC_HIP(hipMemcpyAsync(S.bandB0_dev.p, S.bandB0.data(), S.bandB0.size() * sizeof(double), hipMemcpyHostToDevice, st));
        }
    }
    if (S.ref_r.n < size_t(n))
    {
        S.ref_r.alloc(size_t(n));
        S.ref_dy.alloc(size_t(n));
    }
    // |M|_inf from the host band (rows of the symmetric matrix: |M(i, i-d)| counts for rows i and i-d)
    double mnorm = 0.0;
    {
        std::vector<double> rs(static_cast<size_t>(n), 0.0);
        for (int64_t i = 0; i < n; i++)
            for (int d = 0; d < bw; d++)
            {
                if (i - d < 0)
                    continue;
                const double a = S.band0[size_t(i) * bw + d] - S.sigma * (S.pencil ? S.bandB0[size_t(i) * bw + d] : (d == 0 ? 1.0 : 0.0));
                rs[size_t(i)] += std::fabs(a);
                if (d > 0)
                    rs[size_t(i - d)] += std::fabs(a);
            }
        for (double v : rs)
            mnorm = std::max(mnorm, v);
    }
    DevBuf<double> px, py;
    DevBuf<unsigned long long> norms;
    px.alloc(size_t(n));
    py.alloc(size_t(n));
    norms.alloc(3);
    hipLaunchKernelGGL(k_probe_fill, row_blocks(n), dim3(kThreads), 0, st, n, px.p);
    MISPEC_HIP(hipGetLastError());
    solve_level(*S.ctx, *S.top, px.p, py.p);
    const unsigned grid = unsigned(std::min<int64_t>((n + kThreads - 1) / kThreads, 1024));
    constexpr int kMaxRefine = 12;
    constexpr double kTarget = 4.0e-15;  // a few eps: what a backward-stable solve of a band matrix gives
    double omega = 0.0, prev = 1e300;
    for (int it = 0;; it++)
    {
        MISPEC_HIP(hipMemsetAsync(norms.p, 0, 3 * sizeof(unsigned long long), st));
        hipLaunchKernelGGL(k_band_resid, row_blocks(n), dim3(kThreads), 0, st, n, S.band_b, S.sigma, S.band0_dev.p,
                           S.pencil ? S.bandB0_dev.p : nullptr, px.p, py.p, S.ref_r.p);
        hipLaunchKernelGGL(k_absmax, dim3(grid), dim3(kThreads), 0, st, n, S.ref_r.p, norms.p, 0);
        hipLaunchKernelGGL(k_absmax, dim3(grid), dim3(kThreads), 0, st, n, py.p, norms.p, 1);
        hipLaunchKernelGGL(k_absmax, dim3(grid), dim3(kThreads), 0, st, n, px.p, norms.p, 2);
        MISPEC_HIP(hipGetLastError());
        unsigned long long hb[3];
        MISPEC_HIP(hipMemcpyAsync(hb, norms.p, sizeof(hb), hipMemcpyDeviceToHost, st));
        MISPEC_HIP(hipStreamSynchronize(st));
        double nr, ny, nx;
        std::memcpy(&nr, &hb[0], 8);
        std::memcpy(&ny, &hb[1], 8);
        std::memcpy(&nx, &hb[2], 8);
        omega = nr / (mnorm * ny + nx);
        if (!(omega == omega) || !(ny == ny))
            throw Error(MISPEC_EINVAL, "SparseSymShiftSolve: factorization failed with the given shift (the probe solve is not finite; " +
                                           std::to_string(fs.boosts) + " boosted pivots)");
        if (omega <= kTarget)
            break;
        if (it == kMaxRefine || (it >= 2 && omega > 0.9 * prev))
        {
            // stagnation: accept when the probe is still solved to 1e-13 (the error then sits below the eigensolver's
            // tolerances), otherwise the shift is (numerically) singular for this factorisation
            if (omega <= 1e-13)
                break;
            throw Error(MISPEC_EINVAL, "SparseSymShiftSolve: factorization failed with the given shift (iterative refinement stalls at backward error " +
                                           std::to_string(omega) + " after " + std::to_string(it) + " steps; " + std::to_string(fs.boosts) +
                                           " boosted pivots, smallest pivot ratio " + std::to_string(fs.min_pivot_ratio) + ")");
        }
        prev = omega;
        S.refine_steps = it + 1;
        refine_once(S, px.p, py.p);
    }
    S.probe_backward_error = omega;
    // The step count was calibrated on ONE probe right-hand side.  With boosted pivots the factors are those of a perturbed
    // matrix, and a Lanczos vector with large components along the nearly singular directions of a boosted chunk can need one
    // step more than the probe did: every later solve therefore runs one step beyond the calibrated count.  A
    // definite shift (no boosted pivot: every configuration of BASELINE.json) keeps its count, normally 0.
    if (S.boosted_pivots > 0)
        S.refine_steps += 1;
}
}  // namespace

namespace mispec {

namespace {
__global__ __launch_bounds__(256) void k_ss_gather(int64_t n, const int32_t* __restrict__ perm, const double* __restrict__ src,
                                                   double* __restrict__ dst)
{
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n)
        dst[i] = src[perm[i]];
}
__global__ __launch_bounds__(256) void k_ss_scatter(int64_t n, const int32_t* __restrict__ perm, const double* __restrict__ src,
                                                    double* __restrict__ dst)
{
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n)
        dst[perm[i]] = src[i];
}
void to_stored_order(const mispec_symshift& S, const double* src, double* dst)
{
    hipLaunchKernelGGL(k_ss_gather, dim3(unsigned((S.n + 255) / 256)), dim3(256), 0, S.ctx->stream, S.n, S.perm_dev.p, src, dst);
    MISPEC_HIP(hipGetLastError());
}
void from_stored_order(const mispec_symshift& S, const double* src, double* dst)
{
    hipLaunchKernelGGL(k_ss_scatter, dim3(unsigned((S.n + 255) / 256)), dim3(256), 0, S.ctx->stream, S.n, S.perm_dev.p, src, dst);
    MISPEC_HIP(hipGetLastError());
}
void perm_scratch(const mispec_symshift& S)
{
    if (S.perm_x.n < size_t(S.n))
    {
        S.perm_x.alloc(size_t(S.n));
        S.perm_y.alloc(size_t(S.n));
    }
}
}  // namespace

void launch_shiftsolve(const mispec_symshift& S, const double* x_dev, double* y_dev)
{
    if (!S.factored)
        throw Error(MISPEC_ELOGIC, "SparseSymShiftSolve: need to call set_shift() first");
    if (S.dense)
        launch_row_gemv(*S.ctx, S.inverse.p, S.n, S.n, S.n, x_dev, y_dev);
    else
    {
        const double* x = x_dev;
        double* y = y_dev;
        if (S.reordered())  // the caller's index order is kept: x -> stored order, solve with P (A - sigma B) P', y back
        {
            perm_scratch(S);
            to_stored_order(S, x_dev, S.perm_x.p);
            x = S.perm_x.p;
            y = S.perm_y.p;
        }
        solve_level(*S.ctx, *S.top, x, y);
        for (int it = 0; it < S.refine_steps; it++)
            refine_once(S, x, y);
        if (S.reordered())
            from_stored_order(S, S.perm_y.p, y_dev);
    }
}

}  // namespace mispec

namespace mispec {
void launch_band_cholesky_solve(const mispec_symshift& S, bool upper, const double* x_dev, double* y_dev)
{
    if (!S.factored || S.dense || !S.cholesky_ready)
        throw Error(MISPEC_ELOGIC, "SparseCholesky (banded): the factorisation is not available");
    if (S.reordered())
    {
        // G G' = P B P', so B = (P'G)(P'G)': the factor the generalized solver works with is P'G — (P'G)^{-1} x = G^{-1} (P x),
        // (P'G)^{-T} x = P' (G^{-T} x)
        perm_scratch(S);
        if (upper)
        {
            chol_backward(*S.ctx, *S.top, x_dev, S.perm_y.p);
            from_stored_order(S, S.perm_y.p, y_dev);
        }
        else
        {
            to_stored_order(S, x_dev, S.perm_x.p);
            chol_forward(*S.ctx, *S.top, S.perm_x.p, y_dev);
        }
        return;
    }
    if (upper)
        chol_backward(*S.ctx, *S.top, x_dev, y_dev);
    else
        chol_forward(*S.ctx, *S.top, x_dev, y_dev);
}
}  // namespace mispec

mispec_symshift::~mispec_symshift() {}

// =================================================================================================
// C ABI
// =================================================================================================
namespace {
struct TriangleInput
{
    const int32_t* outer;
    const int32_t* inner;
    const double* val;
    bool lower;
    bool row_major;
};

// calls fn(row >= col, value) for every entry of the selected triangle, like selfadjointView<Uplo>
template <typename Fn>
void for_each_entry(const TriangleInput& T, int64_t n, Fn&& fn)
{
    for (int64_t o = 0; o < n; o++)
        for (int32_t p = T.outer[o]; p < T.outer[o + 1]; p++)
        {
            const int64_t in = T.inner[p];
            MISPEC_REQUIRE(in >= 0 && in < n, "mispec_symshift_create: index out of range");
            const int64_t r = T.row_major ? o : in, c = T.row_major ? in : o;
            if (T.lower ? (r >= c) : (r <= c))
                fn(r >= c ? r : c, r >= c ? c : r, T.val[p]);
        }
}

int symshift_create_impl(mispec_ctx* ctx, int64_t n, const TriangleInput& A, const TriangleInput* B, mispec_symshift** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && out && A.outer && n >= 1, "mispec_symshift_create: bad argument");
        MISPEC_REQUIRE(ctx->comm.allgather == nullptr, "mispec_symshift_create: shift-and-invert operators cannot be row-sharded");
        auto S = std::make_unique<mispec_symshift>();
        S->ctx = ctx;
        S->n = n;
        S->pencil = (B != nullptr);
        // first the bandwidth of the triangle(s) ...
        int64_t countA = 0, countB = 0;
        for_each_entry(A, n, [&](int64_t r, int64_t c, double) {
            S->half_bandwidth = std::max<int64_t>(S->half_bandwidth, r - c);
            countA++;
        });
        if (B)
            for_each_entry(*B, n, [&](int64_t r, int64_t c, double) {
                S->half_bandwidth = std::max<int64_t>(S->half_bandwidth, r - c);
                countB++;
            });
        S->half_bandwidth_as_given = S->half_bandwidth;
        std::vector<int32_t> inv;  // old -> new when the matrix is reordered
        // (the adjacency below holds every off-diagonal entry of A and of B twice behind int32 row pointers: a pencil with more
        // entries than that is not reordered — it ends as the clean "unsupported pattern" error instead — ADVICE r05)
        if (!band_path(n, S->half_bandwidth) && n > kMaxDense && n < (int64_t(1) << 31) &&
            2 * (countA + countB) < (int64_t(1) << 31) && !option_is("reorder", "none"))
        {
            // Too wide for the band kernels as it comes and too large for the dense path: try a bandwidth-reducing ordering
            // (reverse Cuthill-McKee on the pattern of A (+ B)) before giving up — the reference's SparseLU / SimplicialLDLT
            // order their matrix too.  Adopted only if the reordered band fits the kernels.
            std::vector<int32_t> rp(size_t(n) + 1, 0);
            const auto count = [&](int64_t r, int64_t c, double) {
                if (r != c)
                {
                    rp[size_t(r) + 1]++;
                    rp[size_t(c) + 1]++;
                }
            };
            for_each_entry(A, n, count);
            if (B)
                for_each_entry(*B, n, count);
            for (int64_t i = 0; i < n; i++)
                rp[size_t(i) + 1] += rp[size_t(i)];
            std::vector<int32_t> ci(static_cast<size_t>(rp[size_t(n)]));
            std::vector<int32_t> fill(rp.begin(), rp.end() - 1);
            const auto place = [&](int64_t r, int64_t c, double) {
                if (r != c)
                {
                    ci[size_t(fill[size_t(r)]++)] = int32_t(c);
                    ci[size_t(fill[size_t(c)]++)] = int32_t(r);
                }
            };
            for_each_entry(A, n, place);
            if (B)
                for_each_entry(*B, n, place);
            std::vector<int32_t> perm;
            ReorderStats st;
            if (rcm_order(n, rp.data(), ci.data(), true, 0.0, perm, &st) && int64_t(perm.size()) == n)
            {
                inv.assign(size_t(n), 0);
                for (int64_t i = 0; i < n; i++)
                    inv[size_t(perm[size_t(i)])] = int32_t(i);
                int64_t hb = 0;
                const auto width = [&](int64_t r, int64_t c, double) { hb = std::max<int64_t>(hb, std::llabs(int64_t(inv[size_t(r)]) - int64_t(inv[size_t(c)]))); };
                for_each_entry(A, n, width);
                if (B)
                    for_each_entry(*B, n, width);
                if (band_path(n, hb))
                {
                    S->half_bandwidth = hb;
                    S->perm_host = perm;
                }
                else
                    inv.clear();
            }
        }
        if (band_path(n, S->half_bandwidth))
        {
            // ... then, for a band, the band itself (assembled once; every set_shift() starts from it)
            S->band_b = int(std::max<int64_t>(1, std::min<int64_t>(S->half_bandwidth, n - 1)));  // a diagonal matrix: width 1, zeros
            const size_t bw = size_t(S->band_b) + 1;
            S->band0.assign(size_t(n) * bw, 0.0);
            // (r >= c in the caller's order; in the stored order the larger index is the row)
            const auto at = [&](int64_t r, int64_t c) {
                if (!inv.empty())
                {
                    const int64_t R = inv[size_t(r)], C = inv[size_t(c)];
                    r = std::max(R, C);
                    c = std::min(R, C);
                }
                return size_t(r) * bw + size_t(r - c);
            };
            for_each_entry(A, n, [&](int64_t r, int64_t c, double v) { S->band0[at(r, c)] += v; });
            if (B)
            {
                S->bandB0.assign(size_t(n) * bw, 0.0);
                for_each_entry(*B, n, [&](int64_t r, int64_t c, double v) { S->bandB0[at(r, c)] += v; });
            }
            if (!S->perm_host.empty())
            {
                ctx->make_current();
                S->perm_dev.alloc(size_t(n));
                MISPEC_HIP(hipMemcpy(S->perm_dev.p, S->perm_host.data(), size_t(n) * sizeof(int32_t), hipMemcpyHostToDevice));
            }
            if (factored_on_device(n, S->band_b))
            {
                ctx->make_current();
                S->band0_dev.alloc(S->band0.size());
                MISPEC_HIP(hipMemcpy(S->band0_dev.p, S->band0.data(), S->band0.size() * sizeof(double), hipMemcpyHostToDevice));
                if (B)
                {
                    S->bandB0_dev.alloc(S->bandB0.size());
                    MISPEC_HIP(hipMemcpy(S->bandB0_dev.p, S->bandB0.data(), S->bandB0.size() * sizeof(double), hipMemcpyHostToDevice));
                }
            }
        }
        else
        {
            // ... or (row >= col) triplets for the dense path
            S->rows.reserve(size_t(countA));
            S->cols.reserve(size_t(countA));
            S->vals.reserve(size_t(countA));
            for_each_entry(A, n, [&](int64_t r, int64_t c, double v) {
                S->rows.push_back(r);
                S->cols.push_back(c);
                S->vals.push_back(v);
            });
            if (B)
                for_each_entry(*B, n, [&](int64_t r, int64_t c, double v) {
                    S->rowsB.push_back(r);
                    S->colsB.push_back(c);
                    S->valsB.push_back(v);
                });
        }
        *out = S.release();
    });
}

bool parse_uplo(char uplo, bool& lower)
{
    lower = (uplo == 'L' || uplo == 'l');
    return lower || uplo == 'U' || uplo == 'u';
}
}  // namespace

extern "C" int mispec_symshift_create(mispec_ctx* ctx, int64_t n, const int32_t* outer, const int32_t* inner, const double* val,
                                      char uplo, int row_major, mispec_symshift** out)
{
    bool lower;
    if (!parse_uplo(uplo, lower))
    {
        set_last_error("mispec_symshift_create: uplo must be 'L' or 'U'");
        return MISPEC_EINVAL;
    }
    return symshift_create_impl(ctx, n, TriangleInput{outer, inner, val, lower, row_major != 0}, nullptr, out);
}

// SparseGenRealShiftSolve (MatOp/SparseGenRealShiftSolve.h:33-99): y = (A - sigma I)^{-1} x for a general sparse A.
// The reference factors with Eigen::SparseLU; here the dense path is used (LU with partial pivoting, explicit
// inverse, GEMV), so n <= 4096.
extern "C" int mispec_symshift_create_general(mispec_ctx* ctx, int64_t n, const int32_t* outer, const int32_t* inner, const double* val,
                                              int row_major, mispec_symshift** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && out && outer && n >= 1, "mispec_symshift_create_general: bad argument");
        MISPEC_REQUIRE(n <= kMaxDense,
                       "SparseGenRealShiftSolve: only n <= 4096 is supported on the GPU (the reference uses a general sparse LU)");
        MISPEC_REQUIRE(ctx->comm.allgather == nullptr, "mispec_symshift_create_general: shift-and-invert operators cannot be row-sharded");
        auto S = std::make_unique<mispec_symshift>();
        S->ctx = ctx;
        S->n = n;
        S->general = true;
        S->half_bandwidth = n;  // never the banded (symmetric) path
        for (int64_t o = 0; o < n; o++)
            for (int32_t p = outer[o]; p < outer[o + 1]; p++)
            {
                const int64_t in = inner[p];
                MISPEC_REQUIRE(in >= 0 && in < n, "mispec_symshift_create_general: index out of range");
                S->rows.push_back(row_major ? o : in);
                S->cols.push_back(row_major ? in : o);
                S->vals.push_back(val[p]);
            }
        *out = S.release();
    });
}

extern "C" int mispec_symshift_create_pencil(mispec_ctx* ctx, int64_t n, const int32_t* a_outer, const int32_t* a_inner,
                                             const double* a_val, char a_uplo, int a_row_major, const int32_t* b_outer,
                                             const int32_t* b_inner, const double* b_val, char b_uplo, int b_row_major,
                                             mispec_symshift** out)
{
    bool la, lb;
    if (!parse_uplo(a_uplo, la) || !parse_uplo(b_uplo, lb) || !b_outer)
    {
        set_last_error("mispec_symshift_create_pencil: bad argument (uplo must be 'L' or 'U', B must be given)");
        return MISPEC_EINVAL;
    }
    const TriangleInput B{b_outer, b_inner, b_val, lb, b_row_major != 0};
    return symshift_create_impl(ctx, n, TriangleInput{a_outer, a_inner, a_val, la, a_row_major != 0}, &B, out);
}

extern "C" int mispec_symshift_destroy(mispec_symshift* S)
{
    return guarded([&] {
        if (S)
        {
            S->ctx->make_current();
            delete S;
        }
    });
}

extern "C" int64_t mispec_symshift_rows(const mispec_symshift* S) { return S ? S->n : 0; }
extern "C" int mispec_symshift_bandwidth(const mispec_symshift* S, int64_t* as_given, int64_t* stored, int* reordered)
{
    return guarded([&] {
        MISPEC_REQUIRE(S, "mispec_symshift_bandwidth: NULL argument");
        if (as_given)
            *as_given = S->half_bandwidth_as_given;
        if (stored)
            *stored = S->half_bandwidth;
        if (reordered)
            *reordered = S->reordered() ? 1 : 0;
    });
}

// The levels the banded path plans for an n x n matrix of the given half-bandwidth (no device needed): rows, half-bandwidth,
// chunk length and chunk count per level, the last level (one chunk) being the dense one.  Returns the number of levels, 0 when
// the matrix takes the dense path instead (n <= 4096, band wider than 8), MISPEC_EINVAL when it is unsupported.
extern "C" int mispec_symshift_level_plan(int64_t n, int64_t half_bandwidth, int max_levels, int64_t* rows, int64_t* bandwidth,
                                          int64_t* chunk_rows, int64_t* chunks)
{
    int levels = 0;
    const int rc = guarded([&] {
        MISPEC_REQUIRE(n >= 1 && half_bandwidth >= 0 && max_levels >= 0, "mispec_symshift_level_plan: bad argument");
        if (!band_path(n, half_bandwidth))
        {
            MISPEC_REQUIRE(n <= kMaxDense, "SparseSymShiftSolve: only banded matrices (half-bandwidth <= 64) or n <= 4096 are supported on the GPU");
            return;
        }
        int64_t N = n;
        int b = int(std::max<int64_t>(1, std::min<int64_t>(half_bandwidth, n - 1)));
        struct WideTop
        {
            explicit WideTop(bool on) { g_wide_top = on; }
            ~WideTop() { g_wide_top = false; }
        } wide_top(b > kNarrowBandwidth);
        for (;;)
        {
            int64_t L, P;
            plan_level(N, b, L, P);
            MISPEC_REQUIRE(P == 1 || b <= kMaxBandwidth, "internal: band wider than the chunk kernel supports");
            if (levels < max_levels)
            {
                if (rows)
                    rows[levels] = N;
                if (bandwidth)
                    bandwidth[levels] = b;
                if (chunk_rows)
                    chunk_rows[levels] = L;
                if (chunks)
                    chunks[levels] = P;
            }
            levels++;
            if (P == 1)
                break;
            const int64_t nsep = (P - 1) * b;
            b = int(std::min<int64_t>(2 * int64_t(b) - 1, std::max<int64_t>(nsep - 1, 0)));
            N = nsep;
            MISPEC_REQUIRE(levels < 64, "internal: the level chain does not terminate");
        }
    });
    return rc == MISPEC_OK ? levels : rc;
}

extern "C" int mispec_symshift_set_shift(mispec_symshift* S, double sigma)
{
    return guarded([&] {
        MISPEC_REQUIRE(S, "mispec_symshift_set_shift: NULL argument");
        S->ctx->make_current();
        S->factored = false;
        S->sigma = sigma;
        const int64_t n = S->n, b = S->half_bandwidth;
        if (!S->general && band_path(n, b))
        {
            struct WideTop  // the planning mode of this factorisation, whichever way the scope is left
            {
                explicit WideTop(bool on) { g_wide_top = on; }
                ~WideTop() { g_wide_top = false; }
            } wide_top(S->band_b > kNarrowBandwidth);
            HostBand M;
            M.n = n;
            M.b = S->band_b;
            if (S->band0_dev.p && factored_on_device(n, M.b))
            {
                M.view = &S->band0;  // A - sigma I (or A - sigma B) is formed on the device from the resident band(s)
                M.view_dev = S->band0_dev.p;
                if (S->pencil)
                {
                    M.viewB = &S->bandB0;
                    M.viewB_dev = S->bandB0_dev.p;
                }
                M.shift = sigma;
            }
            else
            {
                // levels factored on the host: the shifted band is built per attempt below, from the resident unshifted copy,
                // by the host's cores (it used to be copied twice and shifted by one core: 0.2 s at n = 1e6, b = 32)
                M.view = &S->band0;
                if (S->pencil)
                    M.viewB = &S->bandB0;
                M.shift = sigma;
            }
            // Up to four attempts with the separators moved by one row each time: an attempt fails when the calibration
            // of the iterative refinement does not converge (singular chunk interiors of a nonsingular matrix)
            S->dense = false;
            for (int attempt = 0;; attempt++)
            {
                g_chunk_bias = attempt < 4 ? attempt : 0;
                g_single_chunk = attempt == 4;
                try
                {
                    HostBand Mt = M;  // factor_level consumes its argument
                    if (Mt.view && (!factored_on_device(n, Mt.b) || !Mt.view_dev))
                    {
                        // the attempt factors this level on the host (or there is no device copy of the band): it needs the shifted band itself
                        PhaseTimer pt("shifted band built (host)", n, int(Mt.b));
                        Mt.a.resize(S->band0.size());
                        const int64_t bw = Mt.b + 1;
                        const double* a0 = S->band0.data();
                        const double* b0 = S->pencil ? S->bandB0.data() : nullptr;
                        double* dst = Mt.a.data();
                        parallel_ranges(n, n * bw >= (int64_t(1) << 22) ? ingest_threads() : 1, [=](int, int64_t r0, int64_t r1) {
                            for (int64_t i = r0; i < r1; i++)
                                for (int64_t d = 0; d < bw; d++)
                                {
                                    const int64_t e = i * bw + d;
                                    dst[e] = a0[e] - sigma * (b0 ? b0[e] : (d == 0 ? 1.0 : 0.0));
                                }
                        });
                        Mt.view = nullptr;
                        Mt.view_dev = nullptr;
                        Mt.viewB = nullptr;
                        Mt.viewB_dev = nullptr;
                    }
                    S->top = std::make_unique<BandLevel>();
                    FactorStats fs;
                    fs.want_cholesky = S->want_cholesky;
                    {
                        PhaseTimer pt("factor_level (all levels)", n, int(Mt.b));
                        factor_level(S->ctx, Mt, *S->top, fs);
                    }
                    {
                        PhaseTimer pt("calibrate_refinement", n, int(S->band_b));
                        calibrate_refinement(*S, fs);
                    }
                    S->negative_pivots = fs.negative;
                    S->cholesky_ready = S->want_cholesky && fs.negative == 0 && fs.boosts == 0;
                    g_chunk_bias = 0;
                    g_single_chunk = false;
                    break;
                }
                catch (const Error& e)
                {
                    g_chunk_bias = 0;
                    g_single_chunk = false;
                    if (attempt == 4 || (attempt == 3 && n > kPivotedLimit))
                    {
                        throw Error(e.code, std::string(e.what()) + " [n = " + std::to_string(n) + ", " + std::to_string(attempt + 1) + " attempts]");
                    }
                }
            }
        }
        else if (n <= kMaxDense)
        {
            std::vector<double> A(size_t(n) * n, 0.0), inv;
            for (size_t e = 0; e < S->vals.size(); e++)
            {
                A[size_t(S->cols[e]) * n + S->rows[e]] += S->vals[e];
                if (!S->general && S->rows[e] != S->cols[e])  // symmetric operators keep one triangle: mirror it
                    A[size_t(S->rows[e]) * n + S->cols[e]] += S->vals[e];
            }
            if (S->pencil)
                for (size_t e = 0; e < S->valsB.size(); e++)
                {
                    A[size_t(S->colsB[e]) * n + S->rowsB[e]] -= sigma * S->valsB[e];
                    if (S->rowsB[e] != S->colsB[e])
                        A[size_t(S->rowsB[e]) * n + S->colsB[e]] -= sigma * S->valsB[e];
                }
            else
                for (int64_t i = 0; i < n; i++)
                    A[size_t(i) * n + i] -= sigma;
            dense_inverse<double>(int(n), A, inv);
            upload_row_major(inv, n, S->inverse);
            S->dense = true;
        }
        else
            throw Error(MISPEC_EINVAL,
                        "SparseSymShiftSolve: only banded matrices (half-bandwidth <= 64 as given or after a reverse Cuthill-McKee "
                        "ordering) or n <= 4096 are supported on the GPU (the reference uses a general sparse LU)");
        S->factored = true;
    });
}

// SparseGenComplexShiftSolve::set_shift(sigmar, sigmai) (MatOp/SparseGenComplexShiftSolve.h:74-99): the operator becomes
// y = Re((A - sigma I)^{-1} x) for real x.  Dense path only: complex LU with partial pivoting on the host, the REAL PART of
// the explicit inverse goes to HBM and is applied by the same GEMV kernel (Re(M x) = Re(M) x for real x).
extern "C" int mispec_symshift_set_shift_complex(mispec_symshift* S, double sigmar, double sigmai)
{
    if (S && sigmai == 0.0)
        return mispec_symshift_set_shift(S, sigmar);
    return guarded([&] {
        MISPEC_REQUIRE(S, "mispec_symshift_set_shift_complex: NULL argument");
        MISPEC_REQUIRE(S->general && !S->pencil, "mispec_symshift_set_shift_complex: complex shifts are for the general (non-symmetric) operator");
        S->ctx->make_current();
        S->factored = false;
        S->sigma = sigmar;
        const int64_t n = S->n;
        MISPEC_REQUIRE(n <= kMaxDense, "SparseGenComplexShiftSolve: only n <= 4096 is supported on the GPU");
        typedef std::complex<double> Cx;
        std::vector<Cx> A(size_t(n) * n, Cx(0.0, 0.0)), inv;
        for (size_t e = 0; e < S->vals.size(); e++)
            A[size_t(S->cols[e]) * n + S->rows[e]] += S->vals[e];
        for (int64_t i = 0; i < n; i++)
            A[size_t(i) * n + i] -= Cx(sigmar, sigmai);
        dense_inverse<Cx>(int(n), A, inv);
        std::vector<double> re(inv.size());
        for (size_t e = 0; e < inv.size(); e++)
            re[e] = inv[e].real();
        upload_row_major(re, n, S->inverse);
        S->dense = true;
        S->factored = true;
    });
}

extern "C" int mispec_symshift_refinement_info(const mispec_symshift* S, int* refine_steps, int64_t* boosted_pivots,
                                               double* min_pivot_ratio, double* probe_backward_error)
{
    return guarded([&] {
        MISPEC_REQUIRE(S, "mispec_symshift_refinement_info: NULL argument");
        if (refine_steps)
            *refine_steps = S->refine_steps;
        if (boosted_pivots)
            *boosted_pivots = S->boosted_pivots;
        if (min_pivot_ratio)
            *min_pivot_ratio = S->min_pivot_ratio;
        if (probe_backward_error)
            *probe_backward_error = S->probe_backward_error;
    });
}

extern "C" int mispec_symshift_solve(const mispec_symshift* S, const double* x_dev, double* y_dev)
{
    return guarded([&] {
        MISPEC_REQUIRE(S && x_dev && y_dev, "mispec_symshift_solve: NULL argument");
        S->ctx->make_current();
        launch_shiftsolve(*S, x_dev, y_dev);
    });
}

extern "C" int mispec_symshift_solve_host(const mispec_symshift* S, const double* x_host, double* y_host)
{
    return guarded([&] {
        MISPEC_REQUIRE(S && x_host && y_host, "mispec_symshift_solve_host: NULL argument");
        S->ctx->make_current();
        if (S->stage_x.n < size_t(S->n))
        {
            S->stage_x.alloc(size_t(S->n));
            S->stage_y.alloc(size_t(S->n));
        }
        hipStream_t s = S->ctx->stream;
        MISPEC_HIP(hipMemcpyAsync(S->stage_x.p, x_host, size_t(S->n) * sizeof(double), hipMemcpyHostToDevice, s));
        launch_shiftsolve(*S, S->stage_x.p, S->stage_y.p);
        MISPEC_HIP(hipMemcpyAsync(y_host, S->stage_y.p, size_t(S->n) * sizeof(double), hipMemcpyDeviceToHost, s));
        MISPEC_HIP(hipStreamSynchronize(s));
    });
}
