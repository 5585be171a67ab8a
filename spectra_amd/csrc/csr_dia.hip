// Diagonal storage of a banded / stencil matrix (SpMV format 2) — the automatic choice for BASELINE.json's matrices: the values
// of a 256-row block as one contiguous [nd][256] piece, y[r] = sum_k dia[k][r] * x[r + off_k] in ascending offset order (the CSR
// row sum's products in its order; absent entries add 0: bit-identical to it), no index, no gather.  k_spmv_dia_win2 — the
// headline kernel — handles two rows per thread with 16-byte loads and takes x from LDS windows; k_spmv_dia_win: one row per thread
// (unaligned operands); k_spmv_dia: direct x loads (offsets in more than 8 clusters).  Replaces SparseSymMatProd::perform_op /
// SparseGenMatProd::perform_op (MatOp/SparseSymMatProd.h:85-90) for such matrices.  Bound: HBM, 8 nd n + 16 n bytes per product.
#include "csr_kernels.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

using namespace mispec;

namespace {

// ---- diagonal storage -----------------------------------------------------------------------------------------------
constexpr int kMaxDia = 32;      // diagonals of the diagonal format
constexpr int kDiaGroup = 8;     // loads issued together per thread: 8 values + 8 x entries

// One thread per row: scatter the row's values into the diagonal-major array.  `pos_of_code` maps a dictionary code to the
// rank of its offset.  Within a row the ranks must increase strictly (columns sorted, no duplicates), else the diagonal
// sum would not be the CSR row sum bit for bit: such matrices raise *bad and keep the CSR kernels.
__global__ __launch_bounds__(256) void k_build_dia(const int32_t* __restrict__ rowptr, const uint8_t* __restrict__ codes,
                                                   const double* __restrict__ val, const int32_t* __restrict__ pos_of_code,
                                                   int64_t nloc, int64_t ld, double* __restrict__ dia, int* __restrict__ bad, int nd_blocked)
{
    const int64_t r = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (r >= nloc)
        return;
    int last = -1;
    for (int p = rowptr[r]; p < rowptr[r + 1]; p++)
    {
        const int pos = pos_of_code[codes[p]];
        if (pos <= last)
            *bad = 1;
        last = pos;
        // nd_blocked > 0: the values of a 256-row block are one contiguous [nd][256] piece (one stream per workgroup)
        if (nd_blocked)
            dia[(int64_t(blockIdx.x) * nd_blocked + pos) * 256 + threadIdx.x] = val[p];
        else
            dia[int64_t(pos) * ld + r] = val[p];
    }
}

// acc + a*b with the product rounded before the sum, as the CSR kernels do it (their products pass through LDS)
__device__ __forceinline__ double add_rounded_product(double acc, double a, double b)
{
#pragma clang fp contract(off)
    const double p = a * b;
    return acc + p;
}

struct DiaArgs
{
    const double* dia;
    const int32_t* off;
    int64_t ld;       // diagonal-major layout: dia[k * ld + r]; 0: block layout dia[(block * nd + k) * 256 + r % 256]
    int nd;
    int col_max;
    int64_t row_begin;
};
// start of thread t's column of values in row-block lb and the stride between consecutive diagonals
__device__ __forceinline__ const double* dia_row(const DiaArgs& da, int lb, int t, int64_t& stride)
{
    if (da.ld == 0)
    {
        stride = 256;
        return da.dia + int64_t(lb) * da.nd * 256 + t;
    }
    stride = da.ld;
    return da.dia + int64_t(lb) * 256 + t;
}

template <bool EPI>
__global__ __launch_bounds__(256) void k_spmv_dia(DiaArgs da, const double* __restrict__ x, double* __restrict__ y, int64_t nrows,
                                                  int nblocks, SpmvEpilogue epi)
{
    __shared__ int off_s[kMaxDia];
    __shared__ double red[4];
    // same XCD-aware row-block map and the same 256-row blocks as k_spmv_csr_stream: the alpha partials of the fused
    // epilogue are identical records
    const int per = (nblocks + 7) >> 3;
    const int lmap = (int(blockIdx.x) & 7) * per + (int(blockIdx.x) >> 3);
    if (lmap >= nblocks)
        return;
    const int lb = epi.first_block + lmap;  // a launch may cover a sub-range of the row-blocks (comm / compute overlap)
    if (EPI && epi.status && *epi.status != 0)
        return;
    const int tid = threadIdx.x;
    if (tid < da.nd)
        off_s[tid] = da.off[tid];
    __syncthreads();
    const int64_t row0 = int64_t(lb) * 256;
    const int nr = int(min(int64_t(256), nrows - row0));
    const int64_t r = row0 + min(tid, nr - 1);  // threads past the last row repeat it (their result is dropped)
    int64_t vstride;
    const double* vrow = dia_row(da, lb, min(tid, nr - 1), vstride);
    const int64_t grow = da.row_begin + r;
    double acc = 0.0;
    for (int g = 0; g < da.nd; g += kDiaGroup)
    {
        double v[kDiaGroup], xv[kDiaGroup];
#pragma unroll
        for (int u = 0; u < kDiaGroup; u++)
        {
            const int d = min(g + u, da.nd - 1);
            v[u] = __builtin_nontemporal_load(vrow + int64_t(d) * vstride);  // read once per SpMV
            const int64_t c = grow + off_s[d];
            xv[u] = x[min(max(c, int64_t(0)), int64_t(da.col_max))];  // out of range only where the value is a padding zero
        }
#pragma unroll
        for (int u = 0; u < kDiaGroup; u++)
            if (g + u < da.nd)
                acc = add_rounded_product(acc, v[u], xv[u]);  // no FMA: bit-identical to the CSR row sum
    }
    if (EPI)
    {
        double contrib = 0.0;
        if (tid < nr)
        {
            const int64_t row = row0 + tid;
            double yv = acc;
            if (epi.v_prev)
                yv -= (epi.h_prev_dev ? *epi.h_prev_dev : epi.h_prev) * epi.v_prev[row];  // Lanczos.h:139
            y[row] = yv;
            contrib = epi.v_rows[row] * yv;  // Lanczos.h:142 partial <v, w>
        }
        const double total = block_reduce_sum(contrib, red);
        if (tid == 0)
            epi.partials[lb] = total;
    }
    else if (tid < nr)
        y[row0 + tid] = acc;
}

// The same product with the x entries of a row-block staged through LDS: the offsets cluster, so a block of 256 rows reads
// a few contiguous windows of x (coalesced, once) instead of one 8-byte load per row and diagonal through the L1.
// NG = groups of eight diagonals whose values a thread keeps in registers.
// POST (one-sweep Lanczos steps only, fac.hip lanczos_step_lagged): the input is the UN-normalised residual f and the division
// by beta = |f| (read from the step state) is applied to the row sums and to the epilogue's v instead of to every window entry —
// w = (A f)/beta - beta v_prev, alpha partial = (f/beta) w — together with the step start that k_scale_step otherwise does
// (H(i,i-1) = beta, the beta < sqrt(eps) stop): no scaling pass and no scaled copy of f, two divisions per row.
template <bool EPI, int NG, int NCW = 8, bool POST = false>  // NCW: registers for window entries (>= number of windows)
__global__ __launch_bounds__(256) void k_spmv_dia_win(DiaArgs da, mispec_dia_windows w, const double* __restrict__ x,
                                                      double* __restrict__ y, int64_t nrows, int nblocks, SpmvEpilogue epi)
{
    extern __shared__ double xs[];
    __shared__ double red[4];
    const int per = (nblocks + 7) >> 3;
    const int lmap = (int(blockIdx.x) & 7) * per + (int(blockIdx.x) >> 3);
    if (lmap >= nblocks)
        return;
    const int lb = epi.first_block + lmap;  // a launch may cover a sub-range of the row-blocks (comm / compute overlap)
    if (EPI && epi.status && *epi.status != 0)
        return;
    const int tid = threadIdx.x;
    double beta = 1.0;
    if (POST)
    {
        // every block takes the same decision from the same beta; one thread records it (Lanczos.h:99-128 without the restart branch)
        StepState* st = static_cast<StepState*>(epi.post_scale_state);
        beta = st->beta;
        const bool first = (lmap == 0 && tid == 0);
        if (beta < epi.post_scale_eps_sqrt)
        {
            if (first)
            {
                st->status = kStepSmallBeta;
                st->stop_step = epi.post_scale_step;
                st->stop_count = 0;
            }
            return;
        }
        if (first)
            st->subd[epi.post_scale_step - 1] = beta;
    }
    const int64_t row0 = int64_t(lb) * 256;
    const int nr = int(min(int64_t(256), nrows - row0));
    int64_t vstride;
    const double* vrow = dia_row(da, lb, min(tid, nr - 1), vstride);
    double v[NG * kDiaGroup];
#pragma unroll
    for (int k = 0; k < NG * kDiaGroup; k++)
        v[k] = __builtin_nontemporal_load(vrow + int64_t(min(k, da.nd - 1)) * vstride);
    // The epilogue's operands travel with the matrix values: issued here, they are in flight during the window staging and
    // the barrier instead of costing the block a second round trip to HBM after its row sums (the kernel is bound by the
    // number of resident blocks, i.e. by latency per block: profiles/rounds_1_2/r02r_*, r03q_*).
    double vprev_early = 0.0, vrow_early = 0.0, hprev_early = 0.0;
    const bool early = EPI && tid < nr;
    if (early)
    {
        if (epi.v_prev)
        {
            vprev_early = epi.v_prev[row0 + tid];
            hprev_early = POST ? beta : (epi.h_prev_dev ? *epi.h_prev_dev : epi.h_prev);
        }
        vrow_early = epi.v_rows[row0 + tid];
        if (POST)
            vrow_early = vrow_early / beta;  // Lanczos.h:106
    }
    const int64_t g0 = da.row_begin + row0;
    // x windows -> LDS.  A window is 256 + span entries: two per thread, ALL loaded before the first LDS write.  (Written as
    // a loop over windows and pieces, each piece was a load, a wait and a write: ten dependent round trips per block on the
    // five clusters of M-band, the latency the occupancy experiments of profiles/rounds_1_2/r02r_* were measuring.)
    const auto xat = [&](int64_t col) { return x[min(max(col, int64_t(0)), int64_t(da.col_max))]; };
    // entry tid of every window in a register of its own; the entries past 256 (the spans: 10 in all for M-band) one per
    // thread, thread t taking the t-th of them — 64 VGPRs in total, i.e. eight workgroups per CU as before
    double xw[NCW], xtail = 0.0;
    int tail_pos = -1;  // LDS slot of this thread's tail entry
    const int tails = w.total - 256 * w.nc;
    {
        int before = 0;
#pragma unroll
        for (int c = 0; c < NCW; c++)
        {
            xw[c] = 0.0;
            if (c < w.nc)
            {
                xw[c] = xat(g0 + w.start[c] + tid);
                const int span = w.len[c] - 256;
                if (tid >= before && tid < before + span)
                {
                    tail_pos = w.base[c] + 256 + (tid - before);
                    xtail = xat(g0 + w.start[c] + 256 + (tid - before));
                }
                before += span;
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NCW; c++)
        if (c < w.nc)
            xs[w.base[c] + tid] = xw[c];
    if (tail_pos >= 0)
        xs[tail_pos] = xtail;
    if (tails > 256)  // more tail entries than threads (very wide clusters): the rest the slow way
    {
        int before = 0;
        for (int c = 0; c < w.nc; c++)
        {
            const int span = w.len[c] - 256;
            for (int t = tid + 256; t < before + span; t += 256)
                if (t >= before)
                {
                    const double xv = xat(g0 + w.start[c] + 256 + (t - before));
                    xs[w.base[c] + 256 + (t - before)] = xv;
                }
            before += span;
        }
    }
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < NG * kDiaGroup; k++)
        if (k < da.nd)
            acc = add_rounded_product(acc, v[k], xs[w.idx[k] + tid]);
    if (EPI)
    {
        double contrib = 0.0;
        if (tid < nr)
        {
            const int64_t row = row0 + tid;
            double yv = POST ? acc / beta : acc;
            if (epi.v_prev)
                yv -= (early ? hprev_early : (epi.h_prev_dev ? *epi.h_prev_dev : epi.h_prev)) *
                      (early ? vprev_early : epi.v_prev[row]);  // Lanczos.h:139
            y[row] = yv;
            contrib = (early ? vrow_early : epi.v_rows[row]) * yv;  // Lanczos.h:142 partial <v, w>
        }
        const double total = block_reduce_sum(contrib, red);
        if (tid == 0)
            epi.partials[lb] = total;
    }
    else if (tid < nr)
        y[row0 + tid] = acc;
}

// Two rows per thread (round 5): the values of a 256-row block are read with 16-byte loads by 128 threads — half the load
// instructions per byte (the one-row-per-thread kernel above issues 8-byte loads, which the memory pipeline serves at 0.54-0.70 of
// the 16-byte rate: it moved 1.38 GB at 5.5 TB/s where the 16-byte kernels of this library reach 5.8-6.3) — rows 2t and 2t + 1,
// y / v_prev / v as 16-byte accesses too.  Same products in the same order, and the alpha record of the block is formed by the
// same tree as everywhere else (per-row contributions through LDS, then the four 64-row shuffle trees and (w0 + w1) + (w2 + w3)):
// bit-identical results and records.  Needs the block layout of the values (dia_row: ld == 0) and 16-byte aligned y / v vectors.
template <bool EPI, int NG, int NCW = 8, bool POST = false>
__global__ __launch_bounds__(128) void k_spmv_dia_win2(DiaArgs da, mispec_dia_windows w, const double* __restrict__ x,
                                                       double* __restrict__ y, int64_t nrows, int nblocks, SpmvEpilogue epi)
{
    extern __shared__ double xs[];  // windows, then 256 per-row contributions of the epilogue
    __shared__ double red[4];
    const int per = (nblocks + 7) >> 3;
    const int lmap = (int(blockIdx.x) & 7) * per + (int(blockIdx.x) >> 3);
    if (lmap >= nblocks)
        return;
    const int lb = epi.first_block + lmap;
    if (EPI && epi.status && *epi.status != 0)
        return;
    const int tid = threadIdx.x;
    double beta = 1.0;
    if (POST)
    {
        StepState* st = static_cast<StepState*>(epi.post_scale_state);
        beta = st->beta;
        const bool first = (lmap == 0 && tid == 0);
        if (beta < epi.post_scale_eps_sqrt)
        {
            if (first)
            {
                st->status = kStepSmallBeta;
                st->stop_step = epi.post_scale_step;
                st->stop_count = 0;
            }
            return;
        }
        if (first)
            st->subd[epi.post_scale_step - 1] = beta;
    }
    const int64_t row0 = int64_t(lb) * 256;
    const int nr = int(min(int64_t(256), nrows - row0));
    const int r0 = 2 * tid;  // rows r0, r0 + 1 of the block (the value array is zero-padded to whole blocks)
    const double* vrow = da.dia + int64_t(lb) * da.nd * 256 + r0;
    double2 v[NG * kDiaGroup];
#pragma unroll
    for (int k = 0; k < NG * kDiaGroup; k++)
    {
        const v2d t2 = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(vrow + int64_t(min(k, da.nd - 1)) * 256));
        v[k] = make_double2(t2.x, t2.y);
    }
    double2 vprev_e = make_double2(0.0, 0.0), vrow_e = make_double2(0.0, 0.0);
    double hprev_e = 0.0;
    const bool have0 = r0 < nr, have1 = r0 + 1 < nr;
    if (EPI && have0)
    {
        if (epi.v_prev)
        {
            if (have1)
                vprev_e = *reinterpret_cast<const double2*>(epi.v_prev + row0 + r0);
            else
                vprev_e.x = epi.v_prev[row0 + r0];
            hprev_e = POST ? beta : (epi.h_prev_dev ? *epi.h_prev_dev : epi.h_prev);
        }
        if (have1)
            vrow_e = *reinterpret_cast<const double2*>(epi.v_rows + row0 + r0);
        else
            vrow_e.x = epi.v_rows[row0 + r0];
        if (POST)
        {
            vrow_e.x = vrow_e.x / beta;  // Lanczos.h:106
            vrow_e.y = vrow_e.y / beta;
        }
    }
    const int64_t g0 = da.row_begin + row0;
    const auto xat = [&](int64_t col) { return x[min(max(col, int64_t(0)), int64_t(da.col_max))]; };
    // windows -> LDS: entries tid and tid + 128 of every window, the entries past 256 (the spans) two per thread
    double xw[NCW][2], xtail[2] = {0.0, 0.0};
    int tail_pos[2] = {-1, -1};
    const int tails = w.total - 256 * w.nc;
    {
        int before = 0;
#pragma unroll
        for (int c = 0; c < NCW; c++)
        {
            xw[c][0] = xw[c][1] = 0.0;
            if (c < w.nc)
            {
                xw[c][0] = xat(g0 + w.start[c] + tid);
                xw[c][1] = xat(g0 + w.start[c] + tid + 128);
                const int span = w.len[c] - 256;
#pragma unroll
                for (int h = 0; h < 2; h++)
                {
                    const int t = tid + 128 * h;
                    if (t >= before && t < before + span)
                    {
                        tail_pos[h] = w.base[c] + 256 + (t - before);
                        xtail[h] = xat(g0 + w.start[c] + 256 + (t - before));
                    }
                }
                before += span;
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NCW; c++)
        if (c < w.nc)
        {
            xs[w.base[c] + tid] = xw[c][0];
            xs[w.base[c] + tid + 128] = xw[c][1];
        }
#pragma unroll
    for (int h = 0; h < 2; h++)
        if (tail_pos[h] >= 0)
            xs[tail_pos[h]] = xtail[h];
    if (tails > 256)
    {
        int before = 0;
        for (int c = 0; c < w.nc; c++)
        {
            const int span = w.len[c] - 256;
            for (int t = tid + 256; t < before + span; t += 128)
                if (t >= before)
                    xs[w.base[c] + 256 + (t - before)] = xat(g0 + w.start[c] + 256 + (t - before));
            before += span;
        }
    }
    __syncthreads();
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int k = 0; k < NG * kDiaGroup; k++)
        if (k < da.nd)
        {
            acc0 = add_rounded_product(acc0, v[k].x, xs[w.idx[k] + r0]);
            acc1 = add_rounded_product(acc1, v[k].y, xs[w.idx[k] + r0 + 1]);
        }
    if (EPI)
    {
        double* cbuf = xs + w.total;  // per-row contributions of the block
        double c0 = 0.0, c1 = 0.0;
        double2 yv;
        yv.x = POST ? acc0 / beta : acc0;
        yv.y = POST ? acc1 / beta : acc1;
        if (epi.v_prev)
        {
            yv.x -= hprev_e * vprev_e.x;  // Lanczos.h:139
            yv.y -= hprev_e * vprev_e.y;
        }
        if (have1)
            *reinterpret_cast<double2*>(y + row0 + r0) = yv;
        else if (have0)
            y[row0 + r0] = yv.x;
        if (have0)
            c0 = vrow_e.x * yv.x;  // Lanczos.h:142 partial <v, w>
        if (have1)
            c1 = vrow_e.y * yv.y;
        cbuf[r0] = c0;
        cbuf[r0 + 1] = c1;
        __syncthreads();
        // the record's tree: wave k of a 256-thread block sums rows 64 k .. 64 k + 63 by shuffles, then (w0 + w1) + (w2 + w3)
        const int wv = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int h = 0; h < 2; h++)
        {
            const int k = 2 * wv + h;
            const double s = wave_reduce_sum(cbuf[64 * k + lane]);
            if (lane == 0)
                red[k] = s;
        }
        __syncthreads();
        if (tid == 0)
            epi.partials[lb] = (red[0] + red[1]) + (red[2] + red[3]);
    }
    else
    {
        if (have1)
            *reinterpret_cast<double2*>(y + row0 + r0) = make_double2(acc0, acc1);
        else if (have0)
            y[row0 + r0] = acc0;
    }
}

}  // namespace

namespace mispec {

// Diagonal storage from the offset codes (device): only for small, well-filled dictionaries whose rows are sorted and free
// of duplicates; anything else keeps the CSR kernels (mispec_csr_set_spmv_format selects among the formats a matrix has).
void build_dia(mispec_csr& A, const std::vector<int32_t>& dict)
{
    const int64_t nloc = A.local_rows();
    const int nd = int(dict.size());
    if (nd == 0 || nd > kMaxDia || nloc == 0 || double(A.nnz) < 0.75 * double(nd) * double(nloc))
        return;
    std::vector<int32_t> order(static_cast<size_t>(nd)), pos(static_cast<size_t>(nd)), offs(static_cast<size_t>(nd));
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return dict[size_t(a)] < dict[size_t(b)]; });
    for (int k = 0; k < nd; k++)
    {
        pos[size_t(order[size_t(k)])] = k;
        offs[size_t(k)] = dict[size_t(order[size_t(k)])];
    }
    // the values of a 256-row block are stored as one contiguous [nd][256] piece, so a workgroup streams ONE 30 KB run instead
    // of nd runs of 2 KB that are 80 MB apart (the diagonal-major layout dia[k][row] of round 1 measured 1.5 % slower and is gone)
    constexpr bool blocked = true;
    const int64_t ld = round_up(nloc, 256);
    DevBuf<int32_t> d_pos;
    DevBuf<int> d_bad;
    d_pos.alloc(size_t(nd));
    d_bad.alloc(1);
    A.dia.alloc(size_t(ld) * size_t(nd));
    A.dia_off.alloc(size_t(nd));
    hipStream_t st = A.ctx->stream;
    MISPEC_HIP(hipMemsetAsync(A.dia.p, 0, A.dia.n * sizeof(double), st));
    MISPEC_HIP(hipMemsetAsync(d_bad.p, 0, sizeof(int), st));
    MISPEC_HIP(hipMemcpyAsync(d_pos.p, pos.data(), pos.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    MISPEC_HIP(hipMemcpyAsync(A.dia_off.p, offs.data(), offs.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_build_dia, dim3(unsigned((nloc + 255) / 256)), dim3(256), 0, st, A.rowptr.p, A.codes.p, A.val.p, d_pos.p, nloc, ld,
                       A.dia.p, d_bad.p, blocked ? nd : 0);
    MISPEC_HIP(hipGetLastError());
    int bad = 0;
    MISPEC_HIP(hipMemcpyAsync(&bad, d_bad.p, sizeof(int), hipMemcpyDeviceToHost, st));
    MISPEC_HIP(hipStreamSynchronize(st));
    if (bad)
    {
        A.dia.release();
        A.dia_off.release();
        return;
    }
    A.dia_ld = blocked ? 0 : ld;
    A.ndia = nd;
    // x windows: consecutive sorted offsets share a window while it stays within 256 + 256 entries
    mispec_dia_windows w;
    int first = 0;
    bool ok = true;
    for (int k = 0; k <= nd && ok; k++)
        if (k == nd || int64_t(offs[size_t(k)]) - int64_t(offs[size_t(first)]) > 256)
        {
            if (w.nc == 8)
            {
                ok = false;
                break;
            }
            const int c = w.nc++;
            w.start[c] = offs[size_t(first)];
            w.len[c] = 256 + (offs[size_t(k) - 1] - offs[size_t(first)]);
            w.base[c] = w.total;
            for (int d = first; d < k; d++)
                w.idx[d] = w.total + (offs[size_t(d)] - offs[size_t(first)]);
            w.total += w.len[c];
            first = k;
        }
    if (ok)
        A.dia_win = w;
}

void launch_spmv_dia(const mispec_csr& A, const SpmvLaunch& L)
{
    const dim3 grid = L.grid, block = L.block;
    const int64_t nloc = L.nloc;
    const int nblocks = L.nblocks;
    const SpmvEpilogue* epi = L.epi;
    const SpmvEpilogue e = L.e;
    const hipEvent_t ev_start = L.ev_start, ev_stop = L.ev_stop;
    const double* x_dev = L.x_dev;
    double* y_dev = L.y_dev;
        const DiaArgs da{A.dia.p, A.dia_off.p, A.dia_ld, A.ndia, int(A.n_cols - 1), A.row_begin};
        // x staged through LDS windows when the offsets form at most 8 clusters, else direct loads (k_spmv_dia)
        // two rows per thread with 16-byte loads (k_spmv_dia_win2) when the layout and the alignment allow; MISPEC_DIA2=0: the
        // one-row-per-thread kernel
        const bool dia2_off = option_int("dia2", 1) == 0;
        const bool dia2 = !dia2_off && A.dia_win.nc > 0 && A.dia_ld == 0 && A.ndia <= 2 * kDiaGroup &&
                          (reinterpret_cast<uintptr_t>(y_dev) & 15) == 0 &&
                          (!epi || ((reinterpret_cast<uintptr_t>(e.v_rows) & 15) == 0 && (reinterpret_cast<uintptr_t>(e.v_prev) & 15) == 0));
        if (dia2)
        {
            const size_t lds2 = size_t(A.dia_win.total + 256) * sizeof(double);
            const dim3 block2(128);
            const int ng = (A.ndia + kDiaGroup - 1) / kDiaGroup;
            const bool post = epi && e.post_scale_state;
#define MISPEC_DIA2_LAUNCH(E, G, W, P)                                                                                                  \
    do                                                                                                                                  \
    {                                                                                                                                   \
        if (ev_start && ev_stop)                                                                                                        \
            hipExtLaunchKernelGGL((k_spmv_dia_win2<E, G, W, P>), grid, block2, lds2, A.ctx->stream, ev_start, ev_stop, 0, da, A.dia_win, \
                                  x_dev, y_dev, nloc, nblocks, e);                                                                      \
        else                                                                                                                            \
            hipLaunchKernelGGL((k_spmv_dia_win2<E, G, W, P>), grid, block2, lds2, A.ctx->stream, da, A.dia_win, x_dev, y_dev, nloc,     \
                               nblocks, e);                                                                                             \
    } while (0)
#define MISPEC_DIA2_W(E, G, P)               \
    do                                       \
    {                                        \
        if (A.dia_win.nc <= 4)               \
            MISPEC_DIA2_LAUNCH(E, G, 4, P);  \
        else if (A.dia_win.nc <= 6)          \
            MISPEC_DIA2_LAUNCH(E, G, 6, P);  \
        else                                 \
            MISPEC_DIA2_LAUNCH(E, G, 8, P);  \
    } while (0)
#define MISPEC_DIA2_G(E, P)          \
    do                               \
    {                                \
        if (ng == 1)                 \
            MISPEC_DIA2_W(E, 1, P);  \
        else                         \
            MISPEC_DIA2_W(E, 2, P);  \
    } while (0)
            if (post)
                MISPEC_DIA2_G(true, true);
            else if (epi)
                MISPEC_DIA2_G(true, false);
            else
                MISPEC_DIA2_G(false, false);
#undef MISPEC_DIA2_G
#undef MISPEC_DIA2_W
#undef MISPEC_DIA2_LAUNCH
            MISPEC_HIP(hipGetLastError());
            return;
        }
        if (A.dia_win.nc > 0)
        {
            const size_t lds = size_t(A.dia_win.total) * sizeof(double);
            const int ng = (A.ndia + kDiaGroup - 1) / kDiaGroup;
#define MISPEC_DIA_WIN_W(E, G, W)                                                                                             \
    do                                                                                                                     \
    {                                                                                                                      \
        if (ev_start && ev_stop)                                                                                           \
            hipExtLaunchKernelGGL((k_spmv_dia_win<E, G, W>), grid, block, lds, A.ctx->stream, ev_start, ev_stop, 0, da, A.dia_win, x_dev, \
                                  y_dev, nloc, nblocks, e);                                                                \
        else                                                                                                               \
            hipLaunchKernelGGL((k_spmv_dia_win<E, G, W>), grid, block, lds, A.ctx->stream, da, A.dia_win, x_dev, y_dev, nloc, nblocks, e); \
    } while (0)
#define MISPEC_DIA_WIN(E, G)              \
    do                                    \
    {                                     \
        if (A.dia_win.nc <= 4)            \
            MISPEC_DIA_WIN_W(E, G, 4);    \
        else if (A.dia_win.nc <= 6)       \
            MISPEC_DIA_WIN_W(E, G, 6);    \
        else                              \
            MISPEC_DIA_WIN_W(E, G, 8);    \
    } while (0)
#define MISPEC_DIA_WIN_G(E)          \
    do                               \
    {                                \
        if (ng == 1)                 \
            MISPEC_DIA_WIN(E, 1);    \
        else if (ng == 2)            \
            MISPEC_DIA_WIN(E, 2);    \
        else if (ng == 3)            \
            MISPEC_DIA_WIN(E, 3);    \
        else                         \
            MISPEC_DIA_WIN(E, 4);    \
    } while (0)
            if (epi && e.post_scale_state)
            {
#define MISPEC_DIA_WIN_POST_W(G, W)                                                                                                    \
    do                                                                                                                                 \
    {                                                                                                                                  \
        if (ev_start && ev_stop)                                                                                                       \
            hipExtLaunchKernelGGL((k_spmv_dia_win<true, G, W, true>), grid, block, lds, A.ctx->stream, ev_start, ev_stop, 0, da, A.dia_win, \
                                  x_dev, y_dev, nloc, nblocks, e);                                                                     \
        else                                                                                                                           \
            hipLaunchKernelGGL((k_spmv_dia_win<true, G, W, true>), grid, block, lds, A.ctx->stream, da, A.dia_win, x_dev, y_dev, nloc, \
                               nblocks, e);                                                                                            \
    } while (0)
#define MISPEC_DIA_WIN_POST(G)               \
    do                                       \
    {                                        \
        if (A.dia_win.nc <= 4)               \
            MISPEC_DIA_WIN_POST_W(G, 4);     \
        else if (A.dia_win.nc <= 6)          \
            MISPEC_DIA_WIN_POST_W(G, 6);     \
        else                                 \
            MISPEC_DIA_WIN_POST_W(G, 8);     \
    } while (0)
                if (ng == 1)
                    MISPEC_DIA_WIN_POST(1);
                else if (ng == 2)
                    MISPEC_DIA_WIN_POST(2);
                else if (ng == 3)
                    MISPEC_DIA_WIN_POST(3);
                else
                    MISPEC_DIA_WIN_POST(4);
#undef MISPEC_DIA_WIN_POST
#undef MISPEC_DIA_WIN_POST_W
            }
            else if (epi)
                MISPEC_DIA_WIN_G(true);
            else
                MISPEC_DIA_WIN_G(false);
#undef MISPEC_DIA_WIN_G
#undef MISPEC_DIA_WIN
#undef MISPEC_DIA_WIN_W
            MISPEC_HIP(hipGetLastError());
            return;
        }
        if (ev_start && ev_stop)
        {
            if (epi)
                hipExtLaunchKernelGGL(k_spmv_dia<true>, grid, block, 0, A.ctx->stream, ev_start, ev_stop, 0, da, x_dev, y_dev, nloc, nblocks, e);
            else
                hipExtLaunchKernelGGL(k_spmv_dia<false>, grid, block, 0, A.ctx->stream, ev_start, ev_stop, 0, da, x_dev, y_dev, nloc, nblocks, e);
        }
        else if (epi)
            hipLaunchKernelGGL(k_spmv_dia<true>, grid, block, 0, A.ctx->stream, da, x_dev, y_dev, nloc, nblocks, e);
        else
            hipLaunchKernelGGL(k_spmv_dia<false>, grid, block, 0, A.ctx->stream, da, x_dev, y_dev, nloc, nblocks, e);
        MISPEC_HIP(hipGetLastError());
        return;
}

}  // namespace mispec
