// The int32 CSR SpMV with the x entries of a row-block staged through LDS windows (k_spmv_csr_win, SpMV format 0 when a window
// table was adopted at ingest) — the kernel BASELINE.json's north_star names: val / col_ind streamed with 16-byte loads, x from
// per-block LDS windows found at ingest (k_build_windows; the same selection code runs on the host: mispec_csr_windows_host),
// products through LDS, row sums in storage order — bit-identical to k_spmv_csr_stream and to the CPU row-dot.  Replaces
// SparseSymMatProd::perform_op / SparseGenMatProd::perform_op (MatOp/SparseSymMatProd.h:85-90).  Bound: HBM, 12 nnz + 20 n + 4.
#include "csr_kernels.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

using namespace mispec;

namespace {

// ---- int32 CSR with the x entries of a row-block staged through LDS windows ------------------------------------------
// k_spmv_csr_stream issues one 8-byte gather per stored entry AFTER its column index has arrived: a third dependent round trip
// per block (row pointers -> val / col_ind -> x), every gather a separate request to the vector L1.  For matrices whose
// columns are local — banded matrices, stencils, meshes after a bandwidth-reducing ordering, with or without a fixed offset
// structure — the columns a 256-row block references fall into a few contiguous ranges of x.  Those ranges are found ONCE at
// ingest (k_build_windows: at most kWinMax windows per block, 128-byte aligned, a 32-int record per block), and the kernel
//   * loads the windows with coalesced 16-byte loads, TOGETHER with the val / col_ind stream and the epilogue's operands
//     (one round trip after the row pointers),
//   * turns every column index into an LDS slot with a chain of compares against the window starts (block-uniform, SGPRs),
//   * reads x from LDS; entries outside every window ("far" entries of a block, flagged in its record) keep the global gather.
// The matrix arrays are the plain int32 CSR (12 bytes per entry, SURVEY.md §8d's bytes); products and the summation order are
// those of k_spmv_csr_stream, so the result is bit-identical to it and to the CPU row-dot.
constexpr int kWinMax = 8;            // windows per 256-row block
constexpr int kWinRec = 32;           // int32 per block: [0] nw | far << 8, [1] total doubles, [2] covered entries, [4..] start, [12..] adj, [20..] end
constexpr int kWinPad = 0x3fffffff;   // start of an unused window (no column reaches it)
constexpr int kWinCapMax = 6144;      // doubles of LDS a block's windows may take (48 KiB); the launch reserves the matrix's maximum
constexpr int kWinLines = int((2 * kFarWindow + 512) / 16);  // 128-byte lines of x a block's bitmap covers (row0 - 131072 ... row0 + 256 + 131072)
constexpr int kWinWords = (kWinLines + 31) / 32;
constexpr int kWinRuns = 256;          // raw runs of touched lines a block may have before neighbours further apart are joined

// ---- the window selection of one 256-row block: plain sequential code shared by the device builder (one thread of the block)
// and the host hook mispec_csr_windows_host, so that the CPU tests exercise the code the device runs -----------------------------
__host__ __device__ inline int64_t win_origin(int64_t row_begin, int64_t row0)
{
    const int64_t o = row_begin + row0 - kFarWindow;
    return (o > 0 ? o : 0) & ~int64_t(15);
}
__host__ __device__ inline int win_ctz(uint32_t w)
{
    int n = 0;
    while (!(w & 1u))
    {
        w >>= 1;
        n++;
    }
    return n;
}
// Runs of touched 128-byte lines of the bitmap; neighbours closer than `gap` lines are one run (the lines in between are loaded
// too).  The gap grows until the runs fit the table.  Returns the number of runs (0: they do not fit at any gap).
__host__ __device__ inline int win_find_runs(const uint32_t* bits, int* rs, int* re)
{
    int n = 0;
    bool fits = false;
    for (int gap = 2; gap <= 2048 && !fits; gap *= 4)
    {
        n = 0;
        fits = true;
        int cs = -1, ce = -1;
        for (int w = 0; w < kWinWords && fits; w++)
        {
            uint32_t word = bits[w];
            while (word)
            {
                const int line = w * 32 + win_ctz(word);
                word &= word - 1;
                if (cs < 0)
                {
                    cs = line;
                    ce = line + 1;
                }
                else if (line - ce <= gap)
                    ce = line + 1;
                else
                {
                    if (n == kWinRuns)
                    {
                        fits = false;
                        break;
                    }
                    rs[n] = cs;
                    re[n] = ce;
                    n++;
                    cs = line;
                    ce = line + 1;
                }
            }
        }
        if (fits && cs >= 0)
        {
            if (n == kWinRuns)
                fits = false;
            else
            {
                rs[n] = cs;
                re[n] = ce;
                n++;
            }
        }
    }
    return fits ? n : 0;
}
// index of the run that holds `line` (the last run with rs <= line)
__host__ __device__ inline int win_run_of(const int* rs, int nruns, int line)
{
    int lo = 0, hi = nruns - 1;
    while (lo < hi)
    {
        const int mid = (lo + hi + 1) >> 1;
        if (rs[mid] <= line)
            lo = mid;
        else
            hi = mid - 1;
    }
    return lo;
}
// From the runs and their entry counts to the block's record: thin runs are left to the gather, the closest runs are merged or the
// lightest dropped until at most kWinMax windows within `cap_doubles` of LDS remain.  entries: stored entries of the block; far:
// entries outside the bitmap's range.
__host__ __device__ inline void win_select(int nruns, int* rs, int* re, int* cnt, int64_t origin, int64_t n_cols, int cap_doubles, int entries,
                                           int far, int32_t* rec)
{
    int n = nruns;
    int dropped = 0;  // entries of runs that are not kept: gathered one by one like the far ones
    const auto remove = [&](int i) {
        for (int k = i; k + 1 < n; k++)
        {
            rs[k] = rs[k + 1];
            re[k] = re[k + 1];
            cnt[k] = cnt[k + 1];
        }
        n--;
    };
    // a window pays for itself when its entries outnumber its lines (a gather moves a 64-byte sector per entry): runs thinner than
    // one entry per two lines are dropped, thin AND short ones too
    for (int i = 0; i < n;)
        if (2 * cnt[i] < re[i] - rs[i] || cnt[i] < 8)
        {
            dropped += cnt[i];
            remove(i);
        }
        else
            i++;
    const int64_t col_end = (n_cols + 1) & ~int64_t(1);  // windows hold pairs of doubles
    const auto length = [&](int i) {
        const int64_t e = origin + int64_t(re[i]) * 16;
        return int((e < col_end ? e : col_end) - (origin + int64_t(rs[i]) * 16));
    };
    int total = 0;
    for (int i = 0; i < n; i++)
        total += length(i);
    // down to kWinMax windows: merge the closest pair when the lines that adds (128 bytes each) cost less than gathering the
    // lightest run's entries (a 64-byte sector each) and the LDS budget allows it, else drop the lightest run
    while (n > kWinMax)
    {
        int best = 0, bestgap = 0x7fffffff, light = 0;
        for (int i = 0; i < n; i++)
        {
            if (i + 1 < n && rs[i + 1] - re[i] < bestgap)
            {
                bestgap = rs[i + 1] - re[i];
                best = i;
            }
            if (cnt[i] < cnt[light])
                light = i;
        }
        if (2 * bestgap <= cnt[light] && total + 16 * bestgap <= cap_doubles)
        {
            total += 16 * bestgap;
            re[best] = re[best + 1];
            cnt[best] += cnt[best + 1];
            remove(best + 1);
        }
        else
        {
            total -= length(light);
            dropped += cnt[light];
            remove(light);
        }
    }
    total = 0;
    for (int i = 0; i < n; i++)
        total += length(i);
    while (n > 0 && total > cap_doubles)  // over the LDS budget: the window with the fewest entries per line goes
    {
        int worst = 0;
        for (int i = 1; i < n; i++)
            if (int64_t(cnt[i]) * (re[worst] - rs[worst]) < int64_t(cnt[worst]) * (re[i] - rs[i]))
                worst = i;
        total -= length(worst);
        dropped += cnt[worst];
        remove(worst);
    }
    const int outside = far + dropped + (nruns == 0 ? entries - far : 0);
    rec[0] = n | ((outside || n == 0 ? 1 : 0) << 8);
    rec[1] = total;
    rec[2] = entries - outside;
    rec[3] = 0;
    for (int i = 28; i < kWinRec; i++)
        rec[i] = 0;
    int base = 0;
    for (int i = 0; i < kWinMax; i++)
    {
        if (i < n)
        {
            const int64_t start = origin + int64_t(rs[i]) * 16;
            const int64_t e = origin + int64_t(re[i]) * 16;
            const int64_t end = e < col_end ? e : col_end;
            rec[4 + i] = int32_t(start);
            rec[4 + kWinMax + i] = int32_t(int64_t(base) - start);
            rec[4 + 2 * kWinMax + i] = int32_t(end);
            base += int(end - start);
        }
        else
        {
            rec[4 + i] = kWinPad;
            rec[4 + kWinMax + i] = 0;
            rec[4 + 2 * kWinMax + i] = kWinPad;
        }
    }
}

__global__ __launch_bounds__(256) void k_build_windows(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind, int64_t nrows,
                                                       int64_t row_begin, int64_t n_cols, int cap_doubles, int32_t* __restrict__ wtab)
{
    __shared__ uint32_t bits[kWinWords];
    __shared__ int s_far, s_n;
    __shared__ int rs[kWinRuns], re[kWinRuns], cnt[kWinRuns];
    const int lb = int(blockIdx.x), tid = int(threadIdx.x);
    const int64_t row0 = int64_t(lb) * 256;
    const int nr = int(min(int64_t(256), nrows - row0));
    const int bs = rowptr[row0], be = rowptr[row0 + nr];
    const int64_t origin = win_origin(row_begin, row0);
    for (int w = tid; w < kWinWords; w += 256)
        bits[w] = 0u;
    if (tid < kWinRuns)
        cnt[tid] = 0;
    if (tid == 0)
        s_far = 0;
    __syncthreads();
    int far = 0;
    for (int p = bs + tid; p < be; p += 256)
    {
        const int64_t rel = int64_t(colind[p]) - origin;
        if (rel < 0 || rel >= int64_t(kWinLines) * 16)
            far++;
        else
            atomicOr(&bits[rel >> 9], 1u << ((rel >> 4) & 31));
    }
    if (far)
        atomicAdd(&s_far, far);
    __syncthreads();
    if (tid == 0)
        s_n = win_find_runs(bits, rs, re);
    __syncthreads();
    const int nruns = s_n;  // entries per run
    if (nruns > 0)
        for (int p = bs + tid; p < be; p += 256)
        {
            const int64_t rel = int64_t(colind[p]) - origin;
            if (rel < 0 || rel >= int64_t(kWinLines) * 16)
                continue;
            atomicAdd(&cnt[win_run_of(rs, nruns, int(rel >> 4))], 1);
        }
    __syncthreads();
    if (tid == 0)
        win_select(nruns, rs, re, cnt, origin, n_cols, cap_doubles, be - bs, s_far, wtab + size_t(lb) * kWinRec);
}

// The same table from HOST arrays, one block after the other (mispec_csr_windows_host: the CPU tests run the selection code the
// device runs, and the GPU tests require the device's table to equal this one).
void build_windows_host(int64_t nrows, int64_t n_cols, int64_t row_begin, const int32_t* rowptr, const int32_t* colind, int32_t* wtab)
{
    const int64_t nblocks = (nrows + 255) / 256;
    std::vector<uint32_t> bits(static_cast<size_t>(kWinWords));
    std::vector<int> rs(static_cast<size_t>(kWinRuns)), re(static_cast<size_t>(kWinRuns)), cnt(static_cast<size_t>(kWinRuns));
    for (int64_t lb = 0; lb < nblocks; lb++)
    {
        const int64_t row0 = lb * 256;
        const int64_t r1 = std::min<int64_t>(nrows, row0 + 256);
        const int bs = rowptr[row0], be = rowptr[r1];
        const int64_t origin = win_origin(row_begin, row0);
        std::fill(bits.begin(), bits.end(), 0u);
        std::fill(cnt.begin(), cnt.end(), 0);
        int far = 0;
        for (int p = bs; p < be; p++)
        {
            const int64_t rel = int64_t(colind[p]) - origin;
            if (rel < 0 || rel >= int64_t(kWinLines) * 16)
                far++;
            else
                bits[size_t(rel >> 9)] |= 1u << ((rel >> 4) & 31);
        }
        const int nruns = win_find_runs(bits.data(), rs.data(), re.data());
        if (nruns > 0)
            for (int p = bs; p < be; p++)
            {
                const int64_t rel = int64_t(colind[p]) - origin;
                if (rel < 0 || rel >= int64_t(kWinLines) * 16)
                    continue;
                cnt[size_t(win_run_of(rs.data(), nruns, int(rel >> 4)))]++;
            }
        win_select(nruns, rs.data(), re.data(), cnt.data(), origin, n_cols, kWinCapMax, be - bs, far, wtab + size_t(lb) * kWinRec);
    }
}

// XI: 16-byte window loads per thread (windows of at most 512 * XI doubles); PF: the next chunk's val / col_ind loads are
// issued before the current chunk's products (two chunks of registers).
template <bool EPI, int ITERS, int XI, bool PF, bool NT = false>
__global__ __launch_bounds__(256) void k_spmv_csr_win(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                                                      const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y,
                                                      int64_t nrows, int nblocks, SpmvEpilogue epi, const int32_t* __restrict__ wtab,
                                                      int col_max)
{
    constexpr int kThreads = 256;
    constexpr int kCap = chunk_cap(kThreads) - (4 - ITERS) * kThreads * 4;
    extern __shared__ __attribute__((aligned(16))) double smem_win[];
    double* const prod = smem_win;             // kCap + 4 products
    double* const xs = smem_win + kCap + 4;    // the block's windows of x
    __shared__ double red[4];

    const int per = (nblocks + 7) >> 3;
    const int lmap = (int(blockIdx.x) & 7) * per + (int(blockIdx.x) >> 3);
    if (lmap >= nblocks)
        return;
    const int lb = epi.first_block + lmap;
    if (EPI && epi.status && *epi.status != 0)
        return;

    const int tid = threadIdx.x;
    const int64_t row0 = int64_t(lb) * kThreads;
    const int nr = int(min(int64_t(kThreads), nrows - row0));
    const int32_t* __restrict__ rec = wtab + size_t(lb) * kWinRec;
    const bool far = (rec[0] >> 8) != 0;
    const int total = rec[1];
    int st[kWinMax], ad[kWinMax], en[kWinMax];
#pragma unroll
    for (int w = 0; w < kWinMax; w++)
    {
        st[w] = rec[4 + w];
        ad[w] = rec[4 + kWinMax + w];
        en[w] = rec[4 + 2 * kWinMax + w];
    }
    const int bs = rowptr[row0];
    const int be = rowptr[row0 + nr];
    int rs = 0, re = 0;
    if (tid < nr)
    {
        rs = rowptr[row0 + tid];
        re = rowptr[row0 + tid + 1];
    }
    // the epilogue's operands travel with the matrix stream (as in k_spmv_dia_win)
    double vprev_early = 0.0, vrow_early = 0.0, hprev_early = 0.0;
    const bool early = EPI && tid < nr;
    if (early)
    {
        if (epi.v_prev)
        {
            vprev_early = epi.v_prev[row0 + tid];
            hprev_early = epi.h_prev_dev ? *epi.h_prev_dev : epi.h_prev;
        }
        vrow_early = epi.v_rows[row0 + tid];
    }

    struct Chunk
    {
        double2 va[ITERS][2];
        int4 ci[ITERS];
    };
    const auto load_chunk = [&](Chunk& C, int cs) {
        const int a0 = cs & ~3;
        const int ce = min(be, a0 + kCap);
        const int last = (ce - 1) & ~3;
#pragma unroll
        for (int it = 0; it < ITERS; it++)
        {
            const int base = min(a0 + tid * 4 + it * (kThreads * 4), last);
            if (NT)  // the matrix stream is read once: keep it from evicting the x windows the XCD's blocks share in L2
            {
                const v2d a01 = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(val + base));
                const v2d a23 = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(val + base + 2));
                const v4i c4 = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(colind + base));
                C.va[it][0] = make_double2(a01.x, a01.y);
                C.va[it][1] = make_double2(a23.x, a23.y);
                C.ci[it] = make_int4(c4.x, c4.y, c4.z, c4.w);
            }
            else
            {
                C.va[it][0] = *reinterpret_cast<const double2*>(val + base);
                C.va[it][1] = *reinterpret_cast<const double2*>(val + base + 2);
                C.ci[it] = *reinterpret_cast<const int4*>(colind + base);
            }
        }
    };
    Chunk cur, nxt;
    if (bs < be)
        load_chunk(cur, bs);

    // windows of x -> LDS: position p of the concatenated windows is column p - adj of its window
    {
        double2 xv[XI];
#pragma unroll
        for (int k = 0; k < XI; k++)
        {
            const int p = 2 * (tid + k * kThreads);
            xv[k] = make_double2(0.0, 0.0);
            if (p < total)
            {
                int a = ad[0];
#pragma unroll
                for (int w = 1; w < kWinMax; w++)
                    a = p >= st[w] + ad[w] ? ad[w] : a;
                const int c = p - a;
                if (c + 1 <= col_max)
                    xv[k] = *reinterpret_cast<const double2*>(x + c);
                else
                    xv[k].x = x[min(c, col_max)];
            }
        }
#pragma unroll
        for (int k = 0; k < XI; k++)
        {
            const int p = 2 * (tid + k * kThreads);
            if (p < total)
                *reinterpret_cast<double2*>(&xs[p]) = xv[k];
        }
    }
    __syncthreads();

    const unsigned slot_max = unsigned(max(total, 1) - 1);
    double acc = 0.0;
    for (int cs = bs; cs < be;)
    {
        const int a0 = cs & ~3;
        const int ce = min(be, a0 + kCap);
        if (PF && ce < be)
            load_chunk(nxt, ce);
        // x of every entry: LDS slot through the window chain; far entries from global memory
        double xg[ITERS][4];
        if (!far)
        {
#pragma unroll
            for (int it = 0; it < ITERS; it++)
            {
                const int c4[4] = {cur.ci[it].x, cur.ci[it].y, cur.ci[it].z, cur.ci[it].w};
#pragma unroll
                for (int j = 0; j < 4; j++)
                {
                    int a = ad[0];
#pragma unroll
                    for (int w = 1; w < kWinMax; w++)
                        a = c4[j] >= st[w] ? ad[w] : a;
                    // entries of the alignment lead-in / the padding belong to other blocks: clamp, their products are never summed
                    xg[it][j] = xs[min(unsigned(c4[j] + a), slot_max)];
                }
            }
        }
        else
        {
#pragma unroll
            for (int it = 0; it < ITERS; it++)
            {
                const int c4[4] = {cur.ci[it].x, cur.ci[it].y, cur.ci[it].z, cur.ci[it].w};
#pragma unroll
                for (int j = 0; j < 4; j++)
                {
                    int a = ad[0], e = en[0];
#pragma unroll
                    for (int w = 1; w < kWinMax; w++)
                    {
                        const bool ge = c4[j] >= st[w];
                        a = ge ? ad[w] : a;
                        e = ge ? en[w] : e;
                    }
                    if (c4[j] >= st[0] && c4[j] < e)
                        xg[it][j] = xs[c4[j] + a];
                    else
                        xg[it][j] = x[c4[j]];
                }
            }
        }
#pragma unroll
        for (int it = 0; it < ITERS; it++)
        {
            const int base = a0 + tid * 4 + it * (kThreads * 4);
            if (base < ce)
            {
                double2 p0, p1;
                p0.x = cur.va[it][0].x * xg[it][0];
                p0.y = cur.va[it][0].y * xg[it][1];
                p1.x = cur.va[it][1].x * xg[it][2];
                p1.y = cur.va[it][1].y * xg[it][3];
                *reinterpret_cast<double2*>(&prod[base - a0]) = p0;
                *reinterpret_cast<double2*>(&prod[base - a0 + 2]) = p1;
            }
        }
        __syncthreads();
        const int lo = max(rs, cs), hi = min(re, ce);
        {
            int k = lo;
            for (; k + 4 <= hi; k += 4)
            {
                const double p0 = prod[k - a0], p1 = prod[k - a0 + 1], p2 = prod[k - a0 + 2], p3 = prod[k - a0 + 3];
                acc += p0;
                acc += p1;
                acc += p2;
                acc += p3;
            }
            for (; k < hi; k++)
                acc += prod[k - a0];
        }
        cs = ce;
        if (cs < be)
        {
            __syncthreads();
            if (PF)
                cur = nxt;
            else
                load_chunk(cur, cs);
        }
    }

    if (EPI)
    {
        double contrib = 0.0;
        if (tid < nr)
        {
            const int64_t row = row0 + tid;
            double yv = acc;
            if (epi.v_prev)
                yv -= hprev_early * vprev_early;  // Lanczos.h:139
            y[row] = yv;
            contrib = vrow_early * yv;  // Lanczos.h:142 partial <v, w>
        }
        const double total_c = block_reduce_sum(contrib, red);
        if (tid == 0)
            epi.partials[lb] = total_c;
    }
    else if (tid < nr)
        y[row0 + tid] = acc;
}

}  // namespace

namespace mispec {

// x windows for the int32 CSR kernel (k_spmv_csr_win), from the device copy of the index arrays.  Adopted when at least 75 % of
// the entries get their x from a window; MISPEC_CSR_WIN=0 keeps k_spmv_csr_stream for every matrix.
void build_windows(mispec_csr& A)
{
    const bool off = option_is("csr_win", "0");
    const int64_t nloc = A.local_rows();
    // (columns at or beyond the start sentinel of unused windows would select a padding window: no table for such a matrix — ADVICE r05)
    if (off || nloc == 0 || A.nnz == 0 || spmv_rows_per_block() != 256 || A.n_cols > int64_t(kWinPad))
        return;
    const int nblocks = spmv_num_blocks(nloc);
    hipStream_t st = A.ctx->stream;
    A.wtab.alloc(size_t(nblocks) * kWinRec);
    hipLaunchKernelGGL(k_build_windows, dim3(unsigned(nblocks)), dim3(256), 0, st, A.rowptr.p, A.colind.p, nloc, A.row_begin, A.n_cols,
                       kWinCapMax, A.wtab.p);
    MISPEC_HIP(hipGetLastError());
    std::vector<int32_t> h(size_t(nblocks) * kWinRec);
    MISPEC_HIP(hipMemcpyAsync(h.data(), A.wtab.p, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    MISPEC_HIP(hipStreamSynchronize(st));
    int64_t covered = 0, blocks = 0;
    int lds = 0;
    for (int b = 0; b < nblocks; b++)
    {
        const int32_t* rec = h.data() + size_t(b) * kWinRec;
        if ((rec[0] & 255) == 0)
            continue;
        blocks++;
        covered += rec[2];
        lds = std::max(lds, int(rec[1]));
    }
    A.win_covered = covered;
    A.win_blocks = blocks;
    A.win_lds_doubles = lds;
    if (double(covered) < 0.75 * double(A.nnz))
    {
        A.wtab.release();
        A.win_lds_doubles = 0;
    }
}

void launch_spmv_csr_win(const mispec_csr& A, const SpmvLaunch& L)
{
    const dim3 grid = L.grid, block = L.block;
    const int64_t nloc = L.nloc;
    const int nblocks = L.nblocks;
    const SpmvEpilogue* epi = L.epi;
    const SpmvEpilogue e = L.e;
    const hipEvent_t ev_start = L.ev_start, ev_stop = L.ev_stop;
    const double* x_dev = L.x_dev;
    double* y_dev = L.y_dev;
        // MISPEC_CSR_WIN_ITERS = 2 | 4 (chunk of 2032 / 4080 products), MISPEC_CSR_WIN_PF = 0 | 1 (next chunk's loads ahead)
        // (read once; MISPEC_KERNEL_PROBE=1 — tools/probe_csr_win.py — re-reads them at every launch so that one process can compare)
        struct Knobs
        {
            int iters, pf;
            bool nt;
        };
        const auto read_knobs = [] {
            const char* e_iters = option("csr_win_iters");
            const char* e_pf = option("csr_win_pf");
            const char* e_nt = option("csr_win_nt");
            return Knobs{e_iters ? atoi(e_iters) : 0, e_pf ? atoi(e_pf) : -1, e_nt && atoi(e_nt) != 0};
        };
        const bool probe = option("kernel_probe") != nullptr;
        static const Knobs cached = read_knobs();
        const Knobs knobs = probe ? read_knobs() : cached;
        const int env_iters = knobs.iters, env_pf = knobs.pf;
        const bool nt = knobs.nt;
        // measured in the solver loop (profiles/r09a, r09b): chunks of 1008 products with the next chunk's loads ahead — the
        // smallest LDS footprint, most resident blocks — for up to 16 entries per row (M-band 0.372 -> 0.369 ms, jittered band
        // 0.412 -> 0.368 ms against the gather kernel on the same box); longer rows take larger chunks (fewer barrier rounds)
        const int auto_iters = double(A.nnz) <= 16.0 * double(nloc) ? 1 : (double(A.nnz) <= 32.0 * double(nloc) ? 2 : 4);
        int iters = env_iters == 1 || env_iters == 2 || env_iters == 4 ? env_iters : auto_iters;
        // (64 bytes of margin: the kernel's static `red[4]` shares the 64 KiB with the dynamic allocation — ADVICE r05)
        while (iters > 1 && size_t(chunk_cap(256) - (4 - iters) * 1024 + 4 + A.win_lds_doubles) * sizeof(double) > 65536 - 64)
            iters >>= 1;  // products + windows within the 64 KiB a launch gets without an attribute
        const bool pf = env_pf >= 0 ? env_pf != 0 : true;
        const int xi = (A.win_lds_doubles + 511) / 512;
        const int cap = chunk_cap(256) - (4 - iters) * 1024;
        const size_t lds = size_t(cap + 4 + A.win_lds_doubles) * sizeof(double);
        const int col_max = int(A.n_cols - 1);
#define MISPEC_WIN_LAUNCH_NT(E, I, X, P, N)                                                                                           \
    do                                                                                                                             \
    {                                                                                                                              \
        if (ev_start && ev_stop)                                                                                                   \
            hipExtLaunchKernelGGL((k_spmv_csr_win<E, I, X, P, N>), grid, block, lds, A.ctx->stream, ev_start, ev_stop, 0, A.rowptr.p, \
                                  A.colind.p, A.val.p, x_dev, y_dev, nloc, nblocks, e, A.wtab.p, col_max);                          \
        else                                                                                                                       \
            hipLaunchKernelGGL((k_spmv_csr_win<E, I, X, P, N>), grid, block, lds, A.ctx->stream, A.rowptr.p, A.colind.p, A.val.p,  \
                               x_dev, y_dev, nloc, nblocks, e, A.wtab.p, col_max);                                                 \
    } while (0)
#define MISPEC_WIN_LAUNCH(E, I, X, P)               \
    do                                              \
    {                                               \
        if (nt)                                     \
            MISPEC_WIN_LAUNCH_NT(E, I, X, P, true); \
        else                                        \
            MISPEC_WIN_LAUNCH_NT(E, I, X, P, false); \
    } while (0)
#define MISPEC_WIN_X(E, I, P)              \
    do                                     \
    {                                      \
        if (xi <= 3)                       \
            MISPEC_WIN_LAUNCH(E, I, 3, P); \
        else if (xi <= 5)                  \
            MISPEC_WIN_LAUNCH(E, I, 5, P); \
        else if (xi <= 8)                  \
            MISPEC_WIN_LAUNCH(E, I, 8, P); \
        else                               \
            MISPEC_WIN_LAUNCH(E, I, 12, P); \
    } while (0)
#define MISPEC_WIN(E)                     \
    do                                    \
    {                                     \
        if (iters == 2 && pf)             \
            MISPEC_WIN_X(E, 2, true);     \
        else if (iters == 2)              \
            MISPEC_WIN_X(E, 2, false);    \
        else if (iters == 1 && pf)        \
            MISPEC_WIN_X(E, 1, true);     \
        else if (iters == 1)              \
            MISPEC_WIN_X(E, 1, false);    \
        else                              \
            MISPEC_WIN_X(E, 4, false);    \
    } while (0)
        if (epi)
            MISPEC_WIN(true);
        else
            MISPEC_WIN(false);
#undef MISPEC_WIN
#undef MISPEC_WIN_X
#undef MISPEC_WIN_LAUNCH
#undef MISPEC_WIN_LAUNCH_NT
        MISPEC_HIP(hipGetLastError());
        return;
}

}  // namespace mispec

extern "C" int mispec_csr_windows_host(int64_t n_rows, int64_t n_cols, int64_t row_begin, const int32_t* rowptr, const int32_t* colind,
                                       int32_t* records_out)
{
    return guarded([&] {
        MISPEC_REQUIRE(n_rows >= 0 && n_cols >= 0 && rowptr && records_out && (colind || rowptr[n_rows] == rowptr[0]),
                       "mispec_csr_windows_host: bad argument");
        build_windows_host(n_rows, n_cols, row_begin, rowptr, colind, records_out);
    });
}
