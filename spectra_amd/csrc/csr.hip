// CSR ingest, synthetic matrix generation and the fp64 CSR SpMV kernel for gfx950.
//
// SpMV design ("CSR-stream", row-block per workgroup, LDS partial products):
//   * a workgroup of 256 threads owns 256 consecutive rows, i.e. ONE contiguous run of
//     val/col_ind (~3.8k entries at 15 nnz/row).  All 256 threads stream that run with
//     16-byte loads (2 x double2 + 1 x int4 per thread per step), gather x[col] and park the
//     products in LDS — the HBM side is perfectly coalesced whatever the row lengths are;
//   * after one barrier, thread r adds up row r's products from LDS in storage order, which
//     is exactly the order of the CPU row-dot (oracle SparseCsr / Eigen's row-major product),
//     so the result is bit-identical to it (up to FMA contraction in the fused epilogue);
//   * rows longer than the LDS chunk are handled by looping over chunks, each thread
//     accumulating the part of its row inside the chunk;
//   * blockIdx -> row-block map is XCD-aware: hardware sends block b to XCD b%8, so XCD k is
//     given the k-th contiguous eighth of the rows and its private 4 MiB L2 sees one sliding
//     window of x instead of eight interleaved ones.
//   * offset-coded variant (CODES): when every stored entry lies on one of <= 256 distinct diagonals
//     (banded / stencil matrices, all of BASELINE.json's configs) the matrix also keeps one byte per
//     entry, col = global_row + dict[code], and the kernel streams 9 instead of 12 bytes per entry.
//     The row of an entry is found through a byte table that the row-owning threads write into LDS
//     while the streaming loads are in flight (it aliases the product buffer, which is not live yet);
//     products and summation order are unchanged, so the result is bit-identical to the plain kernel.
//   * diagonal variant (k_spmv_dia): when the dictionary has at most 32 diagonals and they are at least 3/4 full,
//     the values are also kept diagonal-major (dia[k][row], zero where the matrix has no entry) and
//     y[r] = sum_k dia[k][r] * x[r + off_k] in ascending offset order — no index, no gather, no LDS, every load
//     coalesced, 8 bytes per stored slot.  Same products in the same order as the CSR row sum (absent entries add 0), so
//     again bit-identical.
//   * x windows (k_spmv_csr_win, round 5): for matrices whose columns are local but not on fixed offsets the x entries of a
//     row-block come from LDS windows found at ingest instead of one gather per entry; the arrays stay the plain int32 CSR
//     (see the comment above k_build_windows).  The default for format 0 when the table is adopted and rows hold >= 9 entries.
//   * k_spmv_dia_win / k_spmv_dia_win2: diagonal storage with the x windows of a block in LDS; win2 (round 5, the default) handles
//     two rows per thread with 16-byte loads of the values.
// Bound: HBM.  Algorithmic bytes per launch: 12*nnz + 4*(rows+1) + 8*cols + 8*rows (CSR with int32
// indices); the offset-coded variant's compulsory traffic is 9*nnz + ... .
#include "csr.hpp"
#include "krylov.hpp"
#include "reorder.hpp"

#include <hip/hip_ext.h>

#include <chrono>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>

using namespace mispec;

namespace {

typedef double v2d __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

// THREADS * 4 entries * 4 load steps = 16 * THREADS >= cap + 3 (k_spmv_csr_stream's ITERS = 4)
// products per LDS chunk for a THREADS-row workgroup: 256 -> (4080+4)*8 B + 32 B <= 32 KiB -> 5 workgroups / CU
constexpr int chunk_cap(int threads) { return threads * 16 - 16; }
// offset-coded variant: 1 KiB of the 32 KiB goes to the dictionary -> (3952+4)*8 + 1024 + 32 B, still 5 workgroups / CU
constexpr int chunk_cap_codes(int threads) { return threads * 16 - 144; }
constexpr int kMaxDict = 256;

struct SpmvCodes
{
    const uint8_t* codes;
    const int32_t* dict;
    int ndict;
    int col_max;       // n_cols - 1
    int64_t row_begin; // global index of local row 0
};

__device__ __forceinline__ double wave_reduce_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, 64);
    return v;
}

// Deterministic 256-thread sum; every thread returns the total.
__device__ __forceinline__ double block_reduce_sum(double v, double* red)
{
    v = wave_reduce_sum(v);
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// ITERS: 16-byte load groups per thread and chunk (4: chunks of ~4080 entries, 32 KiB of LDS, 5 workgroups per CU; 2: chunks of
// ~2032 entries, 16 KiB, 8 workgroups per CU — for matrices with few entries per row, whose 256-row blocks would leave half of
// the large chunk unused while the LDS it reserves caps the occupancy)
template <bool EPI, bool NT, int kThreads, bool CODES, int ITERS = 4>
__global__ __launch_bounds__(kThreads) void k_spmv_csr_stream(const int32_t* __restrict__ rowptr,
                                                               const int32_t* __restrict__ colind,
                                                               const double* __restrict__ val,
                                                               const double* __restrict__ x, double* __restrict__ y,
                                                               int64_t nrows, int nblocks, SpmvEpilogue epi, SpmvCodes cd)
{
    constexpr int kLoadIters = ITERS;
    constexpr int kCap = (CODES ? chunk_cap_codes(kThreads) : chunk_cap(kThreads)) - (4 - ITERS) * kThreads * 4;
    __shared__ __attribute__((aligned(16))) double prod[kCap + 4];
    __shared__ double red[4];
    __shared__ int dict_s[CODES ? kMaxDict : 1];

    // XCD-aware map: gridDim.x == 8 * per; block b runs on XCD b % 8 and takes the (b/8)-th
    // row-block of that XCD's contiguous range.
    const int per = (nblocks + 7) >> 3;
    const int lmap = (int(blockIdx.x) & 7) * per + (int(blockIdx.x) >> 3);
    if (lmap >= nblocks)
        return;
    const int lb = epi.first_block + lmap;  // a launch may cover a sub-range of the row-blocks (comm / compute overlap)
    if (EPI && epi.status && *epi.status != 0)
        return;

    const int tid = threadIdx.x;
    const int64_t row0 = int64_t(lb) * kThreads;
    const int nr = int(min(int64_t(kThreads), nrows - row0));
    const int bs = rowptr[row0];
    const int be = rowptr[row0 + nr];
    int rs = 0, re = 0;
    if (tid < nr)
    {
        rs = rowptr[row0 + tid];
        re = rowptr[row0 + tid + 1];
    }

    if (CODES && tid < cd.ndict)
        dict_s[tid] = cd.dict[tid];  // visible after the first barrier of the chunk loop
    const uint32_t grow0 = uint32_t(cd.row_begin + row0);

    double acc = 0.0;
    for (int cs = bs; cs < be;)
    {
        const int a0 = cs & ~3;  // 32-byte aligned start for the vector loads
        const int ce = min(be, a0 + kCap);

        // phase 1: issue every streaming load of this chunk.  No branches: lanes past the end of
        // the chunk re-read its last aligned group (one broadcast line), so the loads of all
        // steps are in flight together and waits are counted, not drained.
        const int last = (ce - 1) & ~3;
        double2 va[kLoadIters][2];
        int4 ci[kLoadIters];
#pragma unroll
        for (int it = 0; it < kLoadIters; it++)
        {
            const int base = min(a0 + tid * 4 + it * (kThreads * 4), last);
            // (NT instantiations are not launched: non-temporal val / col_ind streams measured slower stand-alone in round 1 and in
            // the solver loop in round 3 — 0.424 -> 0.459 ms int32, 0.363 -> 0.395 ms coded, profiles/r05k_*)
            if (NT)
            {
                const v2d a01 = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(val + base));
                const v2d a23 = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(val + base + 2));
                va[it][0] = make_double2(a01.x, a01.y);
                va[it][1] = make_double2(a23.x, a23.y);
                if (CODES)
                    ci[it].x = int(__builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(cd.codes + base)));
                else
                {
                    const v4i c4 = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(colind + base));
                    ci[it] = make_int4(c4.x, c4.y, c4.z, c4.w);
                }
            }
            else
            {
                va[it][0] = *reinterpret_cast<const double2*>(val + base);
                va[it][1] = *reinterpret_cast<const double2*>(val + base + 2);
                if (CODES)
                    ci[it].x = int(*reinterpret_cast<const uint32_t*>(cd.codes + base));
                else
                    ci[it] = *reinterpret_cast<const int4*>(colind + base);
            }
        }
        if (CODES)
        {
            // Row table: thread r stamps r on the entries of row r inside this chunk.  The table lives in the
            // product buffer, which nobody reads or writes until the second barrier below.
            uint8_t* rowid = reinterpret_cast<uint8_t*>(prod);
            const int flo = max(rs, cs), fhi = min(re, ce);
            for (int k = flo; k < fhi; k++)
                rowid[k - a0] = uint8_t(tid);
            __syncthreads();
            uint32_t rid[kLoadIters];
#pragma unroll
            for (int it = 0; it < kLoadIters; it++)
            {
                const int base = min(a0 + tid * 4 + it * (kThreads * 4), last);
                rid[it] = *reinterpret_cast<const uint32_t*>(rowid + (base - a0));
            }
            __syncthreads();
            // col = global row + dict[code]; entries outside [cs, ce) (alignment lead-in, padding) carry
            // stale row ids: clamp so the gather stays inside x — their products are never summed.
#pragma unroll
            for (int it = 0; it < kLoadIters; it++)
            {
                const uint32_t c4 = uint32_t(ci[it].x);
                const uint32_t r4 = rid[it];
                const int c0 = int(grow0 + (r4 & 255u) + uint32_t(dict_s[c4 & 255u]));
                const int c1 = int(grow0 + ((r4 >> 8) & 255u) + uint32_t(dict_s[(c4 >> 8) & 255u]));
                const int c2 = int(grow0 + ((r4 >> 16) & 255u) + uint32_t(dict_s[(c4 >> 16) & 255u]));
                const int c3 = int(grow0 + (r4 >> 24) + uint32_t(dict_s[c4 >> 24]));
                ci[it].x = min(max(c0, 0), cd.col_max);
                ci[it].y = min(max(c1, 0), cd.col_max);
                ci[it].z = min(max(c2, 0), cd.col_max);
                ci[it].w = min(max(c3, 0), cd.col_max);
            }
        }
        // phase 2: gather x
        double xg[kLoadIters][4];
#pragma unroll
        for (int it = 0; it < kLoadIters; it++)
        {
            xg[it][0] = x[ci[it].x];
            xg[it][1] = x[ci[it].y];
            xg[it][2] = x[ci[it].z];
            xg[it][3] = x[ci[it].w];
        }
        // phase 3: products -> LDS
#pragma unroll
        for (int it = 0; it < kLoadIters; it++)
        {
            const int base = a0 + tid * 4 + it * (kThreads * 4);
            if (base < ce)
            {
                double2 p0, p1;
                p0.x = va[it][0].x * xg[it][0];
                p0.y = va[it][0].y * xg[it][1];
                p1.x = va[it][1].x * xg[it][2];
                p1.y = va[it][1].y * xg[it][3];
                *reinterpret_cast<double2*>(&prod[base - a0]) = p0;
                *reinterpret_cast<double2*>(&prod[base - a0 + 2]) = p1;
            }
        }
        __syncthreads();
        // phase 4: thread r sums the part of row r that lies in [cs, ce), in storage order
        const int lo = max(rs, cs), hi = min(re, ce);
        {
            // four LDS reads in flight, added in storage order: the serial read-add chain of a 15-entry row is
            // otherwise 15 LDS round trips on the critical path of the workgroup
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
            __syncthreads();
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
        if (kThreads < 256 && tid < 4)
            red[tid] = 0.0;
        if (kThreads < 256)
            __syncthreads();
        const double total = block_reduce_sum(contrib, red);
        if (tid == 0)
            epi.partials[lb] = total;
    }
    else if (tid < nr)
        y[row0 + tid] = acc;
}

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

// ---- reordered matrices: vector permutations and the un-fused epilogue ----------------------------------------------
__global__ __launch_bounds__(256) void k_perm_gather(int64_t n, const int32_t* __restrict__ perm, const double* __restrict__ src,
                                                     double* __restrict__ dst)
{
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n)
        dst[i] = src[perm[i]];
}
__global__ __launch_bounds__(256) void k_perm_scatter(int64_t n, const int32_t* __restrict__ perm, const double* __restrict__ src,
                                                      double* __restrict__ dst)
{
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n)
        dst[perm[i]] = src[i];
}
// The Lanczos epilogue of the SpMV kernels as its own pass, on the same 256-row blocks (so the alpha partials are the same
// records): y -= h_prev * v_prev, partials[block] = sum v * y  (Lanczos.h:139,142)
__global__ __launch_bounds__(256) void k_epilogue_blocks(double* __restrict__ y, int64_t nrows, SpmvEpilogue epi)
{
    __shared__ double red[4];
    if (epi.status && *epi.status != 0)
        return;
    if (blockDim.x < 256)  // fewer than four waves: the unused wave slots stay zero
    {
        if (threadIdx.x < 4)
            red[threadIdx.x] = 0.0;
        __syncthreads();
    }
    const int64_t row = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    double contrib = 0.0;
    if (row < nrows)
    {
        double yv = y[row];
        if (epi.v_prev)
            yv -= (epi.h_prev_dev ? *epi.h_prev_dev : epi.h_prev) * epi.v_prev[row];
        y[row] = yv;
        contrib = epi.v_rows[row] * yv;
    }
    const double total = block_reduce_sum(contrib, red);
    if (threadIdx.x == 0)
        epi.partials[blockIdx.x] = total;
}

// ---- synthetic band matrix, bit-identical to oracle/synth_matrix.h -------------
__host__ __device__ inline uint64_t synth_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__host__ __device__ inline double synth_value(uint64_t seed, uint64_t a, uint64_t b)
{
    const uint64_t k = synth_mix64(synth_mix64(seed ^ a) ^ (b * 0xD6E8FEB86659FD93ULL));
    return double(k >> 11) * (1.0 / 9007199254740992.0) - 0.5;
}

constexpr int kMaxOffsets = 72;
struct BandSpec
{
    int64_t off[kMaxOffsets];  // sorted, unique, signed (contains 0)
    int count;
};

// number of stored entries in rows [0, i)
__host__ __device__ inline int64_t band_prefix(const BandSpec& s, int64_t n, int64_t i)
{
    int64_t total = 0;
    for (int k = 0; k < s.count; k++)
    {
        const int64_t o = s.off[k];
        const int64_t lo = o < 0 ? -o : 0;            // first row with a valid column
        const int64_t hi = o > 0 ? n - o : n;         // one past the last such row
        const int64_t top = i < hi ? i : hi;
        if (top > lo)
            total += top - lo;
    }
    return total;
}

__global__ void k_synth_band(BandSpec spec, int64_t n, int64_t row_begin, int64_t nloc, int64_t base_nnz, uint64_t seed,
                             int symmetric, int32_t* __restrict__ rowptr, int32_t* __restrict__ colind,
                             double* __restrict__ val, uint8_t* __restrict__ codes)
{
    const int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r > nloc)
        return;
    const int64_t i = row_begin + r;
    int64_t p = band_prefix(spec, n, i) - base_nnz;
    rowptr[r] = int32_t(p);
    if (r == nloc)
        return;
    for (int k = 0; k < spec.count; k++)
    {
        const int64_t j = i + spec.off[k];
        if (j < 0 || j >= n)
            continue;
        const uint64_t a = symmetric ? uint64_t(i < j ? i : j) : uint64_t(i);
        const uint64_t b = symmetric ? uint64_t(i < j ? j : i) : uint64_t(j);
        colind[p] = int32_t(j);
        val[p] = synth_value(seed, a, b);
        if (codes)
            codes[p] = uint8_t(k);  // dictionary = the offset list itself
        p++;
    }
}

void alloc_entries(mispec_csr& A, int64_t nnz)
{
    MISPEC_REQUIRE(nnz < (int64_t(1) << 31) - 16, "matrix shard has too many non-zeros for int32 row pointers");
    const size_t cap = size_t(round_up(nnz, 4) + 8);
    A.colind.alloc(cap);
    A.val.alloc(cap);
    MISPEC_HIP(hipMemsetAsync(A.colind.p, 0, cap * sizeof(int32_t), A.ctx->stream));
    MISPEC_HIP(hipMemsetAsync(A.val.p, 0, cap * sizeof(double), A.ctx->stream));
    A.nnz = nnz;
}

void alloc_codes(mispec_csr& A)
{
    const size_t cap = size_t(round_up(A.nnz, 4) + 8);
    A.codes.alloc(cap);
    MISPEC_HIP(hipMemsetAsync(A.codes.p, 0, cap, A.ctx->stream));
}

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

// Host side of the offset-coded format: one byte per entry of rows [b, e) if the shard's entries lie on at most
// kMaxDict distinct diagonals (col - global row), nothing otherwise.
bool build_offset_codes(int64_t b, int64_t e, const int32_t* rowptr, const int32_t* colind, std::vector<int32_t>& dict,
                        std::vector<uint8_t>& codes)
{
    constexpr int kSlots = 1024;  // open addressing, <= 25 % full
    struct Table
    {
        int64_t key[kSlots];
        int code_of[kSlots];
        std::vector<int64_t> order;  // distinct diagonals in the order of their first appearance
        Table() { std::fill(key, key + kSlots, INT64_MIN); }
        static unsigned slot(int64_t d) { return unsigned(uint64_t(d) * 0x9E3779B97F4A7C15ULL >> 54) & (kSlots - 1); }
        int find(int64_t d) const
        {
            unsigned h = slot(d);
            while (key[h] != INT64_MIN && key[h] != d)
                h = (h + 1) & (kSlots - 1);
            return key[h] == INT64_MIN ? -1 : code_of[h];
        }
        bool insert(int64_t d)  // false: the dictionary is full
        {
            unsigned h = slot(d);
            while (key[h] != INT64_MIN && key[h] != d)
                h = (h + 1) & (kSlots - 1);
            if (key[h] == d)
                return true;
            if (int(order.size()) == kMaxDict)
                return false;
            key[h] = d;
            code_of[h] = int(order.size());
            order.push_back(d);
            return true;
        }
    };
    dict.clear();
    const int64_t p0 = rowptr[b], nnz = rowptr[e] - rowptr[b];
    // pass 1 (host threads over contiguous row ranges): the distinct diagonals of every range in order of first appearance;
    // merged in range order they give the dictionary a single scan would build
    const int nt = int(std::max<int64_t>(1, std::min<int64_t>(ingest_threads(), nnz / 1048576)));
    std::vector<Table> local(static_cast<size_t>(nt));
    std::vector<char> full(static_cast<size_t>(nt), 0);
    parallel_ranges(e - b, nt, [&](int t, int64_t rb, int64_t re) {
        Table& T = local[size_t(t)];
        int64_t last_d = INT64_MIN;
        for (int64_t i = b + rb; i < b + re && !full[size_t(t)]; i++)
            for (int64_t p = rowptr[i]; p < rowptr[i + 1]; p++)
            {
                const int64_t d = int64_t(colind[p]) - i;
                if (d == last_d)
                    continue;
                last_d = d;
                if (!T.insert(d))
                {
                    full[size_t(t)] = 1;
                    break;
                }
            }
    });
    Table G;
    for (int t = 0; t < nt; t++)
    {
        if (full[size_t(t)])
            return false;
        for (int64_t d : local[size_t(t)].order)
            if (!G.insert(d))
                return false;
    }
    if (G.order.empty())
        return false;
    for (int64_t d : G.order)
        dict.push_back(int32_t(d));
    // pass 2: encode
    codes.resize(size_t(nnz));
    parallel_ranges(e - b, nt, [&](int, int64_t rb, int64_t re) {
        int64_t last_d = INT64_MIN;
        int last_code = 0;
        for (int64_t i = b + rb; i < b + re; i++)
            for (int64_t p = rowptr[i]; p < rowptr[i + 1]; p++)
            {
                const int64_t d = int64_t(colind[p]) - i;
                if (d != last_d)
                {
                    last_d = d;
                    last_code = G.find(d);
                }
                codes[size_t(p - p0)] = uint8_t(last_code);
            }
    });
    return true;
}

// Wall-clock seconds of the host stages of the last ingest on this thread (mispec_last_ingest_info): [0] the whole library call,
// [1] triangle -> full matrix, [2] validation + local row pointers, [3] index formats (offset codes, diagonal storage) incl. the
// H2D copies of the CSR arrays, [4] far-gather statistics + reordering, [5] tile image on the host, [6] its upload and split.
thread_local double g_ingest[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
struct IngestTimer
{
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    int slot;
    explicit IngestTimer(int s) : slot(s) {}
    ~IngestTimer() { g_ingest[slot] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

// Upload host CSR rows [begin,end) of a global matrix.
bool reorder_matrix(mispec_csr& A, const int32_t* rowptr, const int32_t* colind, const double* val, bool forced);

mispec_csr* upload_rows(mispec_ctx* ctx, int64_t n_rows, int64_t n_cols, const int32_t* rowptr, const int32_t* colind,
                        const double* val, bool allow_reorder = true, bool structurally_symmetric = false)
{
    MISPEC_REQUIRE(ctx && rowptr && n_rows >= 0 && n_cols >= 0, "csr upload: bad argument");
    MISPEC_REQUIRE(n_cols < (int64_t(1) << 31), "csr upload: column count exceeds int32");
    ctx->make_current();
    auto* A = new mispec_csr();
    try
    {
        A->ctx = ctx;
        A->n_rows = n_rows;
        A->n_cols = n_cols;
        int64_t b, e;
        if (mispec_shard_range(n_rows, ctx->world(), ctx->rank(), &b, &e) != MISPEC_OK)
            throw Error(MISPEC_EINVAL, mispec_last_error());
        A->row_begin = b;
        A->row_end = e;
        const int64_t nloc = e - b;
        const int64_t p0 = rowptr[b], p1 = rowptr[e];
        MISPEC_REQUIRE(p1 >= p0, "csr upload: row pointers must be non-decreasing");
        alloc_entries(*A, p1 - p0);
        std::vector<int32_t> rp(size_t(nloc) + 1);
        {
            IngestTimer timer(2);
            const int nt = int(std::max<int64_t>(1, std::min<int64_t>(ingest_threads(), (p1 - p0) / 1048576)));
            std::vector<char> bad_rp(static_cast<size_t>(nt), 0), bad_col(static_cast<size_t>(nt), 0);
            parallel_ranges(nloc + 1, nt, [&](int t, int64_t ib, int64_t ie) {
                for (int64_t i = ib; i < ie; i++)
                {
                    if (rowptr[b + i] < rowptr[b + (i ? i - 1 : 0)])
                        bad_rp[size_t(t)] = 1;
                    rp[size_t(i)] = int32_t(rowptr[b + i] - p0);
                }
            });
            parallel_ranges(p1 - p0, nt, [&](int t, int64_t pb, int64_t pe) {
                char bad = 0;
                for (int64_t p = p0 + pb; p < p0 + pe; p++)
                    bad |= char(colind[p] < 0 || colind[p] >= n_cols);
                bad_col[size_t(t)] = bad;
            });
            for (int t = 0; t < nt; t++)
            {
                MISPEC_REQUIRE(!bad_rp[size_t(t)], "csr upload: row pointers must be non-decreasing");
                MISPEC_REQUIRE(!bad_col[size_t(t)], "csr upload: column index out of range");
            }
        }
        std::unique_ptr<IngestTimer> formats(new IngestTimer(3));
        A->rowptr.alloc(size_t(nloc) + 1);
        MISPEC_HIP(hipMemcpyAsync(A->rowptr.p, rp.data(), rp.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        if (p1 > p0)
        {
            MISPEC_HIP(hipMemcpyAsync(A->colind.p, colind + p0, size_t(p1 - p0) * sizeof(int32_t), hipMemcpyHostToDevice,
                                      ctx->stream));
            MISPEC_HIP(hipMemcpyAsync(A->val.p, val + p0, size_t(p1 - p0) * sizeof(double), hipMemcpyHostToDevice,
                                      ctx->stream));
        }
        std::vector<int32_t> dict;
        std::vector<uint8_t> codes;
        if (p1 > p0 && build_offset_codes(b, e, rowptr, colind, dict, codes))
        {
            alloc_codes(*A);
            A->dict.alloc(dict.size());
            MISPEC_HIP(hipMemcpyAsync(A->codes.p, codes.data(), codes.size(), hipMemcpyHostToDevice, ctx->stream));
            MISPEC_HIP(hipMemcpyAsync(A->dict.p, dict.data(), dict.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
            A->ndict = int(dict.size());
            build_dia(*A, dict);
        }
        MISPEC_HIP(hipStreamSynchronize(ctx->stream));
        if (p1 > p0)
            build_windows(*A);
        formats.reset();
        A->structurally_symmetric = structurally_symmetric;
        // MISPEC_REORDER = auto (default) | rcm | none: an unsharded square matrix whose gathers are scattered (more than a
        // quarter of the entries further than kFarWindow from the diagonal, and x larger than an L2 slice) is reordered
        // at ingest when reverse Cuthill-McKee localises them (reorder.hip)
        // MISPEC_SPMV_TILES = auto (default) | 0 | 1: the column-blocked tile format for scattered patterns that stay
        // scattered (decided below, after the reordering attempt)
        const char* tmode = option("spmv_tiles");
        const bool tiles_off = tmode && std::strcmp(tmode, "0") == 0, tiles_force = tmode && std::strcmp(tmode, "1") == 0;
        const char* mode = option("reorder");
        const bool off = mode && std::strcmp(mode, "none") == 0;
        const bool force = mode && std::strcmp(mode, "rcm") == 0;
        if (allow_reorder && !off && ctx->world() == 1 && ctx->comm.allgather == nullptr && n_rows == n_cols && p1 > p0)
        {
            IngestTimer timer(4);
            A->far_before = far_fraction(n_rows, rowptr, colind, nullptr, kFarWindow);
            if (force || (n_rows >= 2 * kFarWindow && A->far_before > 0.25))
                reorder_matrix(*A, rowptr, colind, val, force);
        }
        // the two formats for scattered patterns are built from the rows this rank keeps (a row shard: all columns, x is the
        // gathered vector), so they serve sharded runs too; only the reordering above needs the whole matrix on one rank
        const bool whole = ctx->world() == 1 && ctx->comm.allgather == nullptr;
        if (!A->reordered() && p1 > p0 && spmv_rows_per_block() == 256)
        {
            const double far = (whole && allow_reorder && !off && n_rows == n_cols) ? A->far_before
                                                                                     : far_fraction(nloc, rowptr, colind, nullptr, kFarWindow, b);
            A->far_before = far;
            // MISPEC_SPMV_STAGED = auto (default) | 0 | 1: the two-phase format with x and y in LDS (staged.hip).  Since round 4 it is
            // what scattered patterns get (M-rand n = 1e7 in the solver loop: 1.02 ms against 1.45 ms from the tiles); the tiles
            // are then built only on request (MISPEC_SPMV_TILES=1) or when the staged format declines the matrix.
            const char* smode = option("spmv_staged");
            const bool st_off = smode && std::strcmp(smode, "0") == 0, st_force = smode && std::strcmp(smode, "1") == 0;
            const bool scattered = n_cols >= 2 * kFarWindow && far > 0.25;
            bool staged_built = false;
            if (!st_off && (st_force || scattered))
            {
                HostStaged H;
                {
                    IngestTimer timer(8);
                    staged_built = build_staged(nloc, n_cols, rp.data(), colind + p0, val + p0, H, 2 * ctx->num_cu);
                    // a few heavy rows among scattered ones: the image is correct but its batches are nearly empty (serial barrier
                    // rounds); only MISPEC_SPMV_STAGED=1 keeps it, the automatic choice falls back to the tiles / CSR kernels
                    if (staged_built && !H.well_filled && !st_force)
                        staged_built = false;
                }
                if (staged_built)
                {
                    IngestTimer timer(9);
                    upload_staged(H, ctx->stream, A->staged);
                }
            }
            if (!tiles_off && (tiles_force || (scattered && !staged_built)))
            {
                HostTiles H;
                bool built;
                {
                    IngestTimer timer(5);
                    built = build_tiles(nloc, n_cols, rp.data(), colind + p0, val + p0, H);
                }
                if (built)
                {
                    IngestTimer timer(6);
                    upload_tiles(H, ctx->stream, A->tiles);
                }
            }
        }
    }
    catch (...)
    {
        delete A;
        throw;
    }
    return A;
}

// Replace the stored matrix by P A P' (reverse Cuthill-McKee).  Host arrays of the FULL matrix in the caller's order.
// Automatic mode adopts the ordering only when it at least halves the fraction of far gathers; `forced` always does.
bool reorder_matrix(mispec_csr& A, const int32_t* rowptr, const int32_t* colind, const double* val, bool forced)
{
    const int64_t n = A.n_rows;
    MISPEC_REQUIRE(A.n_rows == A.n_cols && A.ctx->world() == 1 && A.ctx->comm.allgather == nullptr,
                   "reordering needs an unsharded square matrix");
    MISPEC_REQUIRE(!A.reordered(), "the matrix is already reordered");
    std::vector<int32_t> perm;
    ReorderStats st;
    if (!rcm_order(n, rowptr, colind, A.structurally_symmetric, forced ? 0.0 : 0.125, perm, &st))
        return false;
    std::vector<int32_t> inv(static_cast<size_t>(n));
    for (int64_t i = 0; i < n; i++)
        inv[size_t(perm[size_t(i)])] = int32_t(i);
    const double before = far_fraction(n, rowptr, colind, nullptr, kFarWindow);
    const double after = far_fraction(n, rowptr, colind, inv.data(), kFarWindow);
    if (!forced && !(after <= 0.5 * before))
        return false;
    std::vector<int32_t> rp, ci;
    std::vector<double> v;
    permute_csr(n, rowptr, colind, val, perm, rp, ci, v);
    std::unique_ptr<mispec_csr> B(upload_rows(A.ctx, n, n, rp.data(), ci.data(), v.data(), false, A.structurally_symmetric));
    A.rowptr.swap(B->rowptr);
    A.colind.swap(B->colind);
    A.val.swap(B->val);
    A.codes.swap(B->codes);
    A.dict.swap(B->dict);
    A.dia.swap(B->dia);
    A.dia_off.swap(B->dia_off);
    std::swap(A.ndict, B->ndict);
    std::swap(A.dia_ld, B->dia_ld);
    std::swap(A.ndia, B->ndia);
    std::swap(A.dia_win, B->dia_win);
    A.tiles.swap(B->tiles);
    A.staged.swap(B->staged);
    A.wtab.swap(B->wtab);
    std::swap(A.win_lds_doubles, B->win_lds_doubles);
    std::swap(A.win_covered, B->win_covered);
    std::swap(A.win_blocks, B->win_blocks);
    A.nnz = B->nnz;
    A.perm.alloc(size_t(n));
    MISPEC_HIP(hipMemcpy(A.perm.p, perm.data(), size_t(n) * sizeof(int32_t), hipMemcpyHostToDevice));
    A.perm_host.swap(perm);
    A.inv_host.swap(inv);
    A.reorder_method = 1;
    A.far_before = before;
    A.far_after = after;
    return true;
}

// Smallest / largest column referenced inside every rank's row block (block = rows per rank): which part of
// the other ranks' slices of x this shard's SpMV reads.  lo starts at INT64_MAX, hi at -1.
constexpr int kMaxPeers = 64;
__global__ __launch_bounds__(256) void k_col_ranges(const int32_t* __restrict__ colind, int64_t nnz, int64_t block, int world,
                                                     long long* __restrict__ lo, long long* __restrict__ hi)
{
    __shared__ int s_lo[kMaxPeers], s_hi[kMaxPeers];
    for (int p = threadIdx.x; p < world; p += blockDim.x)
    {
        s_lo[p] = 0x7fffffff;
        s_hi[p] = -1;
    }
    __syncthreads();
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nnz; i += int64_t(gridDim.x) * blockDim.x)
    {
        const int c = colind[i];
        const int p = int(int64_t(c) / block);
        atomicMin(&s_lo[p], c);
        atomicMax(&s_hi[p], c);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < world; p += blockDim.x)
        if (s_hi[p] >= 0)
        {
            atomicMin(&lo[p], (long long) s_lo[p]);
            atomicMax(&hi[p], (long long) s_hi[p]);
        }
}

// flag[b] = 1 when every entry of 256-row block b has its column in [lo, hi)
__global__ __launch_bounds__(256) void k_block_local(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind, int64_t nrows,
                                                      int rows_per_block, int lo, int hi, int* __restrict__ flag)
{
    __shared__ int outside;
    if (threadIdx.x == 0)
        outside = 0;
    __syncthreads();
    const int64_t r0 = int64_t(blockIdx.x) * rows_per_block;
    const int64_t r1 = min(r0 + rows_per_block, nrows);
    int bad = 0;
    for (int p = rowptr[r0] + int(threadIdx.x); p < rowptr[r1]; p += 256)
    {
        const int c = colind[p];
        bad |= (c < lo || c >= hi);
    }
    if (bad)
        outside = 1;
    __syncthreads();
    if (threadIdx.x == 0)
        flag[blockIdx.x] = outside ? 0 : 1;
}

}  // namespace

namespace mispec {

void interior_blocks(const mispec_csr& A, int64_t col_lo, int64_t col_hi, int& first, int& count)
{
    first = 0;
    count = 0;
    const int64_t nloc = A.local_rows();
    if (nloc == 0 || A.nnz == 0)
        return;
    const int nblocks = spmv_num_blocks(nloc);
    DevBuf<int> d;
    d.alloc(size_t(nblocks));
    hipLaunchKernelGGL(k_block_local, dim3(unsigned(nblocks)), dim3(256), 0, A.ctx->stream, A.rowptr.p, A.colind.p, nloc,
                       spmv_rows_per_block(), int(col_lo), int(col_hi), d.p);
    MISPEC_HIP(hipGetLastError());
    std::vector<int> h(static_cast<size_t>(nblocks));
    MISPEC_HIP(hipMemcpyAsync(h.data(), d.p, h.size() * sizeof(int), hipMemcpyDeviceToHost, A.ctx->stream));
    MISPEC_HIP(hipStreamSynchronize(A.ctx->stream));
    int best_first = 0, best = 0, run_first = 0, run = 0;
    for (int b = 0; b < nblocks; b++)
    {
        if (h[size_t(b)])
        {
            if (run == 0)
                run_first = b;
            if (++run > best)
            {
                best = run;
                best_first = run_first;
            }
        }
        else
            run = 0;
    }
    first = best_first;
    count = best;
}

int spmv_rows_per_block() { return 256; }

bool spmv_can_post_scale(const mispec_csr& A) { return A.spmv_format() == 2 && A.dia_win.nc > 0; }

void launch_to_stored_order(const mispec_csr& A, const double* src, double* dst)
{
    const int64_t n = A.local_rows();
    hipLaunchKernelGGL(k_perm_gather, dim3(unsigned((n + 255) / 256)), dim3(256), 0, A.ctx->stream, n, A.perm.p, src, dst);
    MISPEC_HIP(hipGetLastError());
}
void launch_from_stored_order(const mispec_csr& A, const double* src, double* dst)
{
    const int64_t n = A.local_rows();
    hipLaunchKernelGGL(k_perm_scatter, dim3(unsigned((n + 255) / 256)), dim3(256), 0, A.ctx->stream, n, A.perm.p, src, dst);
    MISPEC_HIP(hipGetLastError());
}

void launch_spmv(const mispec_csr& A, const double* x_dev, double* y_dev, const SpmvEpilogue* epi, hipEvent_t ev_start,
                 hipEvent_t ev_stop)
{
    if (!A.reordered())
    {
        launch_spmv_raw(A, x_dev, y_dev, epi, ev_start, ev_stop);
        return;
    }
    // the caller's index order is kept: x -> stored order, product with P A P', y back, then the epilogue as its own pass
    // over the same 256-row blocks (identical partial records)
    const int64_t n = A.local_rows();
    if (n == 0)
        return;
    if (A.perm_x.n < size_t(n) + 2)
    {
        A.perm_x.alloc(size_t(n) + 2);
        A.perm_y.alloc(size_t(n) + 2);
    }
    launch_to_stored_order(A, x_dev, A.perm_x.p);
    launch_spmv_raw(A, A.perm_x.p, A.perm_y.p, nullptr, ev_start, ev_stop);
    launch_from_stored_order(A, A.perm_y.p, y_dev);
    if (epi)
    {
        const int nblocks = spmv_num_blocks(n);
        hipLaunchKernelGGL(k_epilogue_blocks, dim3(unsigned(nblocks)), dim3(unsigned(spmv_rows_per_block())), 0, A.ctx->stream, y_dev, n,
                           *epi);
        MISPEC_HIP(hipGetLastError());
    }
}

void launch_spmv_raw(const mispec_csr& A, const double* x_dev, double* y_dev, const SpmvEpilogue* epi, hipEvent_t ev_start,
                     hipEvent_t ev_stop, int block_first, int block_count)
{
    const int64_t nloc = A.local_rows();
    if (nloc == 0)
        return;
    const int all_blocks = spmv_num_blocks(nloc);
    if (block_count < 0)
    {
        block_first = 0;
        block_count = all_blocks;
    }
    if (block_count == 0)
        return;
    MISPEC_REQUIRE(block_first >= 0 && block_first + block_count <= all_blocks, "SpMV: row-block range out of bounds");
    const int nblocks = block_count;  // the kernels map blockIdx onto [first_block, first_block + nblocks)
    MISPEC_REQUIRE(!(epi && epi->post_scale_state) || spmv_can_post_scale(A),
                   "SpMV: a post-scaled step start was requested for a matrix that cannot take it");
    const int per = (nblocks + 7) >> 3;
    const int threads = spmv_rows_per_block();
    const dim3 grid(unsigned(per * 8)), block(static_cast<unsigned>(threads));
    SpmvEpilogue e = epi ? *epi : SpmvEpilogue{};
    e.first_block = block_first;
    const int format = A.spmv_format();
    const bool coded = format == 1;
    const SpmvCodes cd{A.codes.p, A.dict.p, A.ndict, int(A.n_cols - 1), A.row_begin};
    if (format == 4)
    {
        MISPEC_REQUIRE(block_count == all_blocks, "SpMV: the staged format does not take row-block sub-ranges");
        launch_spmv_staged(A.staged, A.ctx->stream, x_dev, y_dev, nloc, A.n_cols, all_blocks, epi, ev_start, ev_stop);
        return;
    }
    if (format == 3)
    {
        MISPEC_REQUIRE(block_count == all_blocks, "SpMV: the tile format does not take row-block sub-ranges");
        launch_spmv_tiles(A.tiles, A.ctx->stream, x_dev, y_dev, nloc, all_blocks, epi, ev_start, ev_stop);
        return;
    }
    if (format == 2)
    {
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
    // With an event pair the launch is timed through the dispatch's own completion signal (start/stop of the
    // kernel itself, as a profiler sees it) instead of marker packets around it.
#define MISPEC_SPMV_LAUNCH(K)                                                                                          \
    do                                                                                                                 \
    {                                                                                                                  \
        if (ev_start && ev_stop)                                                                                       \
            hipExtLaunchKernelGGL((K), grid, block, 0, A.ctx->stream, ev_start, ev_stop, 0, A.rowptr.p, A.colind.p,   \
                                  A.val.p, x_dev, y_dev, nloc, nblocks, e, cd);                                       \
        else                                                                                                           \
            hipLaunchKernelGGL((K), grid, block, 0, A.ctx->stream, A.rowptr.p, A.colind.p, A.val.p, x_dev, y_dev,    \
                               nloc, nblocks, e, cd);                                                                  \
    } while (0)
    // up to 16 entries per row on average: the 16 KiB-chunk instantiation.  Measured in the solver loop: 7 per row (reordered
    // stencil) 0.258 -> 0.215 ms, 15 per row (M-band) 0.413 -> 0.394 ms, stand-alone equal
    // (round 4, profiles/r07a: on M-band the two chunk sizes measure alike in the loop, 0.417-0.422 ms, in both step flows)
    const bool small_chunk = !coded && double(A.nnz) <= 16.0 * double(nloc);
    if (!coded && A.windows_active() && (reinterpret_cast<uintptr_t>(x_dev) & 15) == 0)
    {
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
#define MISPEC_SPMV(E)                                                    \
    do                                                                    \
    {                                                                     \
        if (small_chunk)                                                  \
            MISPEC_SPMV_LAUNCH((k_spmv_csr_stream<E, false, 256, false, 2>)); \
        else if (coded)                                                   \
            MISPEC_SPMV_LAUNCH((k_spmv_csr_stream<E, false, 256, true>)); \
        else                                                              \
            MISPEC_SPMV_LAUNCH((k_spmv_csr_stream<E, false, 256, false>)); \
    } while (0)
    if (epi)
        MISPEC_SPMV(true);
    else
        MISPEC_SPMV(false);
#undef MISPEC_SPMV
#undef MISPEC_SPMV_LAUNCH
    MISPEC_HIP(hipGetLastError());
}

bool column_ranges(const mispec_csr& A, int64_t block, int world, std::vector<int64_t>& lo, std::vector<int64_t>& hi)
{
    lo.assign(size_t(world), INT64_MAX);
    hi.assign(size_t(world), -1);
    if (world > kMaxPeers)
        return false;
    if (A.nnz == 0)
        return true;
    DevBuf<long long> d;
    d.alloc(2 * size_t(world));
    std::vector<long long> h(2 * size_t(world));
    for (int p = 0; p < world; p++)
    {
        h[size_t(p)] = INT64_MAX;
        h[size_t(world + p)] = -1;
    }
    MISPEC_HIP(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(long long), hipMemcpyHostToDevice, A.ctx->stream));
    const int grid = int(std::min<int64_t>((A.nnz + 255) / 256, int64_t(A.ctx->num_cu) * 8));
    hipLaunchKernelGGL(k_col_ranges, dim3(unsigned(grid)), dim3(256), 0, A.ctx->stream, A.colind.p, A.nnz, block, world, d.p,
                       d.p + world);
    MISPEC_HIP(hipGetLastError());
    MISPEC_HIP(hipMemcpyAsync(h.data(), d.p, h.size() * sizeof(long long), hipMemcpyDeviceToHost, A.ctx->stream));
    MISPEC_HIP(hipStreamSynchronize(A.ctx->stream));
    for (int p = 0; p < world; p++)
    {
        lo[size_t(p)] = h[size_t(p)];
        hi[size_t(p)] = h[size_t(world + p)];
    }
    return true;
}

}  // namespace mispec

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" int mispec_csr_upload(mispec_ctx* ctx, int64_t n_rows, int64_t n_cols, const int32_t* rowptr_host,
                                 const int32_t* colind_host, const double* val_host, mispec_csr** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && out && rowptr_host, "mispec_csr_upload: NULL argument");
        MISPEC_REQUIRE((colind_host && val_host) || rowptr_host[n_rows] == 0, "mispec_csr_upload: NULL arrays");
        std::fill(g_ingest, g_ingest + 10, 0.0);
        IngestTimer total(0);
        *out = upload_rows(ctx, n_rows, n_cols, rowptr_host, colind_host, val_host);
    });
}

extern "C" int mispec_ingest_threads(void) { return mispec::ingest_threads(); }

extern "C" int mispec_last_ingest_info(double* seconds_out, int count)
{
    return guarded([&] {
        MISPEC_REQUIRE(seconds_out && count >= 1 && count <= 10, "mispec_last_ingest_info: bad argument");
        for (int i = 0; i < count; i++)
            seconds_out[i] = g_ingest[i];
    });
}

extern "C" int mispec_csr_from_csc(mispec_ctx* ctx, int64_t n_rows, int64_t n_cols, const int32_t* colptr,
                                   const int32_t* rowind, const double* val, mispec_csr** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && out && colptr, "mispec_csr_from_csc: NULL argument");
        const int64_t nnz = colptr[n_cols];
        std::vector<int32_t> rp(size_t(n_rows) + 1, 0), ci(static_cast<size_t>(nnz));
        std::vector<double> v(static_cast<size_t>(nnz));
        for (int64_t p = 0; p < nnz; p++)
        {
            MISPEC_REQUIRE(rowind[p] >= 0 && rowind[p] < n_rows, "mispec_csr_from_csc: row index out of range");
            rp[size_t(rowind[p]) + 1]++;
        }
        for (int64_t i = 0; i < n_rows; i++)
            rp[size_t(i) + 1] += rp[size_t(i)];
        std::vector<int32_t> fill(rp.begin(), rp.end() - 1);
        for (int64_t j = 0; j < n_cols; j++)  // column order => each row ends up sorted by column
            for (int32_t p = colptr[j]; p < colptr[j + 1]; p++)
            {
                const int32_t q = fill[size_t(rowind[p])]++;
                ci[size_t(q)] = int32_t(j);
                v[size_t(q)] = val[p];
            }
        *out = upload_rows(ctx, n_rows, n_cols, rp.data(), ci.data(), v.data());
    });
}

namespace {
// The full symmetric matrix (CSR, rows sorted by column) that one stored triangle defines: what mispec_csr_from_triangle uploads.
// ci / v are left uninitialised by the allocation (a std::vector would zero-fill 12 bytes per entry on one thread, and fault
// every page in there): the threads that own the rows touch them first.
struct MirroredCsr
{
    std::vector<int32_t> rp;
    RawVec<int32_t> ci;
    RawVec<double> v;
    int64_t nnz = 0;
};
void mirror_triangle(int64_t n, const int32_t* outer, const int32_t* inner, const double* val, bool lower, bool row_major, MirroredCsr& M)
{
    // (r, c) is the matrix position of an entry whatever the storage order; an entry is kept iff it lies in the requested
    // triangle (selfadjointView<Uplo> ignores the rest).  Row r of the full matrix receives column c, row c column r.
    //
    // A bucketed transpose on the host threads.  The rows are cut into buckets of 2^shift rows whose output (12 bytes per
    // entry) stays cache-resident; pass 1: every thread walks its piece of the INPUT once, sequentially, and appends
    // (row, column, value) records to staging areas per (bucket, thread) — sized by a counting walk, laid out bucket-major and
    // thread-minor, so a bucket's records are contiguous and in input order; pass 2: every bucket is turned into its rows by
    // one thread (count, prefix, place).  Every row is therefore filled in input order — for a sorted triangle that IS
    // ascending column order — with no atomics, and the bytes do not depend on the number of threads.  All traffic is
    // sequential except the placement inside a bucket.
    std::vector<int32_t>& rp = M.rp;
    const int64_t nnz_in = n > 0 ? int64_t(outer[n]) - int64_t(outer[0]) : 0;
    rp.assign(size_t(n) + 1, 0);
    int shift = 15;  // 32768 rows per bucket: ~6 MB of output at 15 entries per row
    while ((n >> shift) > 4096)
        shift++;
    const int64_t nb = (n >> shift) + 1;
    const int nt = int(std::max<int64_t>(1, std::min<int64_t>(ingest_threads(), nnz_in / 131072)));
    struct Rec
    {
        int32_t row, col;
        double val;
    };
    // input pieces: contiguous runs of outers with about the same number of entries
    std::vector<int64_t> piece(static_cast<size_t>(nt) + 1, n);
    piece[0] = 0;
    for (int t = 1; t < nt; t++)
    {
        const int64_t want = outer[0] + nnz_in * t / nt;
        piece[size_t(t)] = std::lower_bound(outer, outer + n, want, [](int32_t a, int64_t w) { return int64_t(a) < w; }) - outer;
    }
    std::vector<char> bad(static_cast<size_t>(nt), 0);
    auto walk = [&](int t, auto&& fn) {
        for (int64_t o = piece[size_t(t)]; o < piece[size_t(t) + 1]; o++)
            for (int32_t p = outer[o]; p < outer[o + 1]; p++)
            {
                const int64_t in = inner[p];
                if (in < 0 || in >= n)
                {
                    bad[size_t(t)] = 1;
                    continue;
                }
                const int64_t r = row_major ? o : in, c = row_major ? in : o;
                if (!(lower ? (r >= c) : (r <= c)))
                    continue;
                fn(r, c, p);
                if (r != c)
                    fn(c, r, p);
            }
    };
    // counting walk: records per (thread, bucket)
    std::vector<int64_t> cnt(size_t(nt) * size_t(nb), 0);
    parallel_ranges(nt, nt, [&](int, int64_t t0, int64_t t1) {
        for (int64_t t = t0; t < t1; t++)
        {
            int64_t* c = cnt.data() + size_t(t) * size_t(nb);
            walk(int(t), [&](int64_t row, int64_t, int32_t) { c[row >> shift]++; });
        }
    });
    for (char f : bad)
        MISPEC_REQUIRE(!f, "mispec_csr_from_triangle: index out of range");
    // staging offsets, bucket-major / thread-minor
    std::vector<int64_t> off(size_t(nt) * size_t(nb) + 1, 0), bucket_begin(size_t(nb) + 1, 0);
    int64_t total = 0;
    for (int64_t b = 0; b < nb; b++)
    {
        bucket_begin[size_t(b)] = total;
        for (int t = 0; t < nt; t++)
        {
            off[size_t(t) * size_t(nb) + size_t(b)] = total;
            total += cnt[size_t(t) * size_t(nb) + size_t(b)];
        }
    }
    bucket_begin[size_t(nb)] = total;
    MISPEC_REQUIRE(total <= INT32_MAX, "mispec_csr_from_triangle: more than 2^31 - 1 entries");
    M.nnz = total;
    RawVec<Rec> stage;
    stage.resize_uninitialized(size_t(std::max<int64_t>(total, 1)));
    M.ci.resize_uninitialized(size_t(std::max<int64_t>(total, 1)));
    M.v.resize_uninitialized(size_t(std::max<int64_t>(total, 1)));
    int32_t* const ci = M.ci.data();
    double* const v = M.v.data();
    parallel_ranges(nt, nt, [&](int, int64_t t0, int64_t t1) {
        for (int64_t t = t0; t < t1; t++)
        {
            std::vector<int64_t> cur(off.begin() + t * nb, off.begin() + (t + 1) * nb);
            walk(int(t), [&](int64_t row, int64_t col, int32_t p) {
                Rec& r = stage[size_t(cur[size_t(row >> shift)]++)];
                r.row = int32_t(row);
                r.col = int32_t(col);
                r.val = val[p];
            });
        }
    });
    // pass 2: a bucket at a time — row counts, then placement; the row pointers get their global base from the bucket's offset
    parallel_ranges(nb, std::min<int64_t>(ingest_threads(), nb), [&](int, int64_t b0, int64_t b1) {
        std::vector<int32_t> cursor;
        for (int64_t b = b0; b < b1; b++)
        {
            const int64_t r0 = b << shift, r1 = std::min<int64_t>(n, (b + 1) << shift);
            if (r0 >= r1)
                continue;
            const Rec* rec = stage.data() + bucket_begin[size_t(b)];
            const int64_t m = bucket_begin[size_t(b) + 1] - bucket_begin[size_t(b)];
            cursor.assign(size_t(r1 - r0) + 1, 0);
            for (int64_t k = 0; k < m; k++)
                cursor[size_t(rec[k].row - r0) + 1]++;
            int64_t run = bucket_begin[size_t(b)];
            for (int64_t r = r0; r < r1; r++)
            {
                const int64_t c = cursor[size_t(r - r0) + 1];
                rp[size_t(r)] = int32_t(run);  // rp[n] is set below
                cursor[size_t(r - r0)] = int32_t(run - bucket_begin[size_t(b)]);
                run += c;
            }
            for (int64_t k = 0; k < m; k++)
            {
                const int64_t q = bucket_begin[size_t(b)] + cursor[size_t(rec[k].row - r0)]++;
                ci[size_t(q)] = rec[k].col;
                v[size_t(q)] = rec[k].val;
            }
        }
    });
    rp[size_t(n)] = int32_t(total);
    stage.resize_uninitialized(0);
    // sort every row by column (the SpMV sums in storage order); rows of a sorted triangle are sorted already
    parallel_ranges(n, ingest_threads(), [&](int, int64_t b, int64_t e) {
        std::vector<int32_t> perm, tc;
        std::vector<double> tv;
        for (int64_t i = b; i < e; i++)
        {
            const int32_t s0 = rp[size_t(i)], e0 = rp[size_t(i) + 1];
            if (e0 - s0 < 2 || std::is_sorted(ci + s0, ci + e0))
                continue;
            perm.resize(size_t(e0 - s0));
            std::iota(perm.begin(), perm.end(), 0);
            std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t c) { return ci[size_t(s0 + a)] < ci[size_t(s0 + c)]; });
            tc.assign(ci + s0, ci + e0);
            tv.assign(v + s0, v + e0);
            for (int32_t k = 0; k < e0 - s0; k++)
            {
                ci[size_t(s0 + k)] = tc[size_t(perm[size_t(k)])];
                v[size_t(s0 + k)] = tv[size_t(perm[size_t(k)])];
            }
        }
    });
}
}  // namespace

// Host-only test hook (no device): the mirrored matrix of a triangle.  rowptr_out: n + 1 entries; colind_out / val_out: capacity
// entries (2 nnz of the input is always enough); *nnz_out = entries written.
extern "C" int mispec_mirror_triangle_host(int64_t n, const int32_t* outer, const int32_t* inner, const double* val, char uplo, int row_major,
                                           int32_t* rowptr_out, int32_t* colind_out, double* val_out, int64_t capacity, int64_t* nnz_out)
{
    return guarded([&] {
        MISPEC_REQUIRE(outer && rowptr_out && nnz_out, "mispec_mirror_triangle_host: NULL argument");
        MISPEC_REQUIRE(uplo == 'L' || uplo == 'U' || uplo == 'l' || uplo == 'u', "mispec_mirror_triangle_host: uplo must be 'L' or 'U'");
        MirroredCsr M;
        mirror_triangle(n, outer, inner, val, uplo == 'L' || uplo == 'l', row_major != 0, M);
        MISPEC_REQUIRE(M.nnz <= capacity, "mispec_mirror_triangle_host: output capacity too small");
        std::copy(M.rp.begin(), M.rp.end(), rowptr_out);
        std::copy(M.ci.data(), M.ci.data() + M.nnz, colind_out);
        std::copy(M.v.data(), M.v.data() + M.nnz, val_out);
        *nnz_out = M.nnz;
    });
}

extern "C" int mispec_csr_from_triangle(mispec_ctx* ctx, int64_t n, const int32_t* outer, const int32_t* inner,
                                        const double* val, char uplo, int row_major, mispec_csr** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && out && outer, "mispec_csr_from_triangle: NULL argument");
        MISPEC_REQUIRE(uplo == 'L' || uplo == 'U' || uplo == 'l' || uplo == 'u', "mispec_csr_from_triangle: uplo must be 'L' or 'U'");
        const bool lower = (uplo == 'L' || uplo == 'l');
        std::fill(g_ingest, g_ingest + 10, 0.0);
        IngestTimer total(0);
        MirroredCsr M;
        {
            IngestTimer timer(1);
            mirror_triangle(n, outer, inner, val, lower, row_major != 0, M);
        }
        *out = upload_rows(ctx, n, n, M.rp.data(), M.ci.data(), M.v.data(), true, true);
    });
}

extern "C" int mispec_csr_synth_band(mispec_ctx* ctx, int64_t n, uint64_t seed, const int64_t* offsets, int noff,
                                     int symmetric, mispec_csr** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && out && n > 0 && noff >= 0 && (noff == 0 || offsets), "mispec_csr_synth_band: bad argument");
        MISPEC_REQUIRE(n < (int64_t(1) << 31), "mispec_csr_synth_band: n exceeds int32 column indices");
        std::vector<int64_t> offs{0};
        for (int k = 0; k < noff; k++)
        {
            offs.push_back(offsets[k]);
            offs.push_back(-offsets[k]);
        }
        std::sort(offs.begin(), offs.end());
        offs.erase(std::unique(offs.begin(), offs.end()), offs.end());
        MISPEC_REQUIRE(int(offs.size()) <= kMaxOffsets, "mispec_csr_synth_band: too many offsets");
        BandSpec spec;
        spec.count = int(offs.size());
        for (int k = 0; k < spec.count; k++)
            spec.off[k] = offs[size_t(k)];

        ctx->make_current();
        auto* A = new mispec_csr();
        try
        {
            A->ctx = ctx;
            A->n_rows = A->n_cols = n;
            int64_t b, e;
            if (mispec_shard_range(n, ctx->world(), ctx->rank(), &b, &e) != MISPEC_OK)
                throw Error(MISPEC_EINVAL, mispec_last_error());
            A->row_begin = b;
            A->row_end = e;
            const int64_t nloc = e - b;
            const int64_t base = band_prefix(spec, n, b);
            alloc_entries(*A, band_prefix(spec, n, e) - base);
            A->rowptr.alloc(size_t(nloc) + 1);
            std::vector<int32_t> dict;
            if (A->nnz > 0)
            {
                alloc_codes(*A);
                dict.assign(offs.begin(), offs.end());
                A->dict.alloc(dict.size());
                MISPEC_HIP(hipMemcpyAsync(A->dict.p, dict.data(), dict.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
                MISPEC_HIP(hipStreamSynchronize(ctx->stream));  // dict is a local
                A->ndict = int(dict.size());
            }
            const int threads = 256;
            const unsigned blocks = unsigned((nloc + 1 + threads - 1) / threads);
            hipLaunchKernelGGL(k_synth_band, dim3(blocks), dim3(threads), 0, ctx->stream, spec, n, b, nloc, base, seed,
                               symmetric, A->rowptr.p, A->colind.p, A->val.p, A->codes.p);
            MISPEC_HIP(hipGetLastError());
            MISPEC_HIP(hipStreamSynchronize(ctx->stream));
            if (A->ndict > 0)
                build_dia(*A, dict);
            build_windows(*A);
        }
        catch (...)
        {
            delete A;
            throw;
        }
        *out = A;
    });
}

extern "C" int mispec_csr_destroy(mispec_csr* A)
{
    return guarded([&] {
        if (A)
        {
            A->ctx->make_current();
            delete A;
        }
    });
}
extern "C" int64_t mispec_csr_rows(const mispec_csr* A) { return A ? A->n_rows : 0; }
extern "C" int64_t mispec_csr_cols(const mispec_csr* A) { return A ? A->n_cols : 0; }
extern "C" int64_t mispec_csr_local_rows(const mispec_csr* A) { return A ? A->local_rows() : 0; }
extern "C" int64_t mispec_csr_local_nnz(const mispec_csr* A) { return A ? A->nnz : 0; }
int mispec_csr::spmv_format() const
{
    const bool blocks256 = spmv_rows_per_block() == 256;
    const bool can_codes = ndict > 0 && blocks256;
    const bool can_dia = ndia > 0 && blocks256;
    const bool can_tiles = tiles.present() && blocks256;
    const bool can_staged = staged.present() && blocks256;
    if (forced_format == 4)
        return can_staged ? 4 : (can_tiles ? 3 : 0);
    if (forced_format == 3)
        return can_tiles ? 3 : 0;
    if (forced_format == 0)
        return 0;
    if (forced_format == -1 && can_staged)
        return 4;  // built only when the pattern asked for it; measured faster than the tiles
    if (forced_format == -1 && can_tiles)
        return 3;
    if (!use_codes)
        return 0;
    if (forced_format == 1)
        return can_codes ? 1 : 0;
    if (forced_format == 2)
        return can_dia ? 2 : (can_codes ? 1 : 0);
    return can_dia ? 2 : (can_codes ? 1 : 0);
}

extern "C" int mispec_csr_offset_codes(const mispec_csr* A)
{
    return A && A->use_codes && A->forced_format != 0 ? A->ndict : 0;
}
extern "C" int mispec_csr_spmv_format(const mispec_csr* A) { return A ? A->spmv_format() : 0; }
extern "C" int mispec_csr_set_spmv_format(mispec_csr* A, int format)
{
    return guarded([&] {
        MISPEC_REQUIRE(A && format >= -1 && format <= 4, "mispec_csr_set_spmv_format: format must be -1 (automatic), 0, 1, 2, 3 or 4");
        A->forced_format = format;
    });
}
extern "C" int mispec_csr_use_windows(mispec_csr* A, int enable)
{
    return guarded([&] {
        MISPEC_REQUIRE(A && enable >= -1 && enable <= 1, "mispec_csr_use_windows: enable must be -1 (automatic), 0 or 1");
        A->use_windows = enable;
    });
}
extern "C" int mispec_csr_windows_info(const mispec_csr* A, int64_t* blocks, int64_t* covered_entries, int64_t* lds_doubles)
{
    return guarded([&] {
        MISPEC_REQUIRE(A, "mispec_csr_windows_info: NULL argument");
        if (blocks)
            *blocks = A->wtab.p ? A->win_blocks : 0;
        if (covered_entries)
            *covered_entries = A->wtab.p ? A->win_covered : 0;
        if (lds_doubles)
            *lds_doubles = A->wtab.p ? A->win_lds_doubles : 0;
    });
}
extern "C" int mispec_csr_windows_host(int64_t n_rows, int64_t n_cols, int64_t row_begin, const int32_t* rowptr, const int32_t* colind,
                                       int32_t* records_out)
{
    return guarded([&] {
        MISPEC_REQUIRE(n_rows >= 0 && n_cols >= 0 && rowptr && records_out && (colind || rowptr[n_rows] == rowptr[0]),
                       "mispec_csr_windows_host: bad argument");
        build_windows_host(n_rows, n_cols, row_begin, rowptr, colind, records_out);
    });
}
extern "C" int mispec_csr_windows_in_use(const mispec_csr* A) { return A && A->windows_active() ? 1 : 0; }
extern "C" int mispec_csr_windows_table(const mispec_csr* A, int32_t* records_out, int64_t capacity)
{
    return guarded([&] {
        MISPEC_REQUIRE(A && records_out, "mispec_csr_windows_table: NULL argument");
        MISPEC_REQUIRE(A->wtab.p && int64_t(A->wtab.n) <= capacity, "mispec_csr_windows_table: no table, or capacity below 32 ints per 256-row block");
        A->ctx->make_current();
        MISPEC_HIP(hipStreamSynchronize(A->ctx->stream));
        MISPEC_HIP(hipMemcpy(records_out, A->wtab.p, A->wtab.n * sizeof(int32_t), hipMemcpyDeviceToHost));
    });
}
extern "C" int mispec_csr_use_offset_codes(mispec_csr* A, int enable)
{
    return guarded([&] {
        MISPEC_REQUIRE(A, "mispec_csr_use_offset_codes: NULL argument");
        A->use_codes = enable != 0;
    });
}
extern "C" double mispec_csr_spmv_bytes(const mispec_csr* A, int stored)
{
    if (!A)
        return 0.0;
    return stored ? A->stored_bytes() : A->algorithmic_bytes();
}

extern "C" int mispec_csr_staged_info(const mispec_csr* A, int64_t* bins, int64_t* slots, int64_t* batches, int64_t* chunks)
{
    return guarded([&] {
        MISPEC_REQUIRE(A, "mispec_csr_staged_info: NULL argument");
        if (bins)
            *bins = A->staged.nbins;
        if (slots)
            *slots = A->staged.slots;
        if (batches)
            *batches = A->staged.nbatches;
        if (chunks)
            *chunks = A->staged.nchunks;
    });
}

extern "C" int mispec_csr_tiles_info(const mispec_csr* A, int64_t* segments, int64_t* entries, int64_t* padding, int64_t* chunks)
{
    return guarded([&] {
        MISPEC_REQUIRE(A, "mispec_csr_tiles_info: NULL argument");
        if (segments)
            *segments = A->tiles.nseg;
        if (entries)
            *entries = A->tiles.entries;
        if (padding)
            *padding = A->tiles.padding;
        if (chunks)
            *chunks = A->tiles.nchunks;
    });
}

extern "C" int mispec_csr_reorder(mispec_csr* A, int method, int* applied)
{
    return guarded([&] {
        MISPEC_REQUIRE(A && (method == 1 || method == -1), "mispec_csr_reorder: method must be 1 (reverse Cuthill-McKee) or -1 (automatic)");
        if (applied)
            *applied = 0;
        if (A->reordered())
        {
            if (applied)
                *applied = 1;
            return;
        }
        MISPEC_REQUIRE(A->n_rows == A->n_cols && A->ctx->world() == 1 && A->ctx->comm.allgather == nullptr,
                       "mispec_csr_reorder: needs an unsharded square matrix");
        A->ctx->make_current();
        MISPEC_HIP(hipStreamSynchronize(A->ctx->stream));
        const int64_t n = A->n_rows;
        std::vector<int32_t> rp(size_t(n) + 1), ci(size_t(A->nnz));
        std::vector<double> v(size_t(A->nnz));
        MISPEC_HIP(hipMemcpy(rp.data(), A->rowptr.p, rp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (A->nnz)
        {
            MISPEC_HIP(hipMemcpy(ci.data(), A->colind.p, ci.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
            MISPEC_HIP(hipMemcpy(v.data(), A->val.p, v.size() * sizeof(double), hipMemcpyDeviceToHost));
        }
        if (method == -1)
        {
            A->far_before = far_fraction(n, rp.data(), ci.data(), nullptr, kFarWindow);
            if (!(n >= 2 * kFarWindow && A->far_before > 0.25))
                return;
        }
        const bool done = reorder_matrix(*A, rp.data(), ci.data(), v.data(), method == 1);
        if (applied)
            *applied = done ? 1 : 0;
    });
}

extern "C" int mispec_csr_reordering(const mispec_csr* A, double* far_before, double* far_after)
{
    if (!A)
        return 0;
    if (far_before)
        *far_before = A->far_before;
    if (far_after)
        *far_after = A->reordered() ? A->far_after : A->far_before;
    return A->reorder_method;
}

extern "C" int mispec_csr_permutation(const mispec_csr* A, int32_t* perm_out)
{
    return guarded([&] {
        MISPEC_REQUIRE(A && perm_out, "mispec_csr_permutation: NULL argument");
        for (int64_t i = 0; i < A->n_rows; i++)
            perm_out[i] = A->reordered() ? A->perm_host[size_t(i)] : int32_t(i);
    });
}

extern "C" int mispec_csr_coeff(const mispec_csr* A, int64_t i, int64_t j, double* out)
{
    return guarded([&] {
        MISPEC_REQUIRE(A && out, "mispec_csr_coeff: NULL argument");
        MISPEC_REQUIRE(i >= A->row_begin && i < A->row_end && j >= 0 && j < A->n_cols, "mispec_csr_coeff: index outside this shard");
        A->ctx->make_current();
        if (A->reordered())  // stored: B(inv[i], inv[j]) = A(i, j)
        {
            i = A->inv_host[size_t(i)];
            j = A->inv_host[size_t(j)];
        }
        int32_t rp[2];
        MISPEC_HIP(hipMemcpy(rp, A->rowptr.p + (i - A->row_begin), sizeof(rp), hipMemcpyDeviceToHost));
        const int len = rp[1] - rp[0];
        *out = 0.0;
        if (len <= 0)
            return;
        std::vector<int32_t> ci(static_cast<size_t>(len));
        std::vector<double> v(static_cast<size_t>(len));
        MISPEC_HIP(hipMemcpy(ci.data(), A->colind.p + rp[0], size_t(len) * sizeof(int32_t), hipMemcpyDeviceToHost));
        MISPEC_HIP(hipMemcpy(v.data(), A->val.p + rp[0], size_t(len) * sizeof(double), hipMemcpyDeviceToHost));
        for (int k = 0; k < len; k++)
            if (ci[size_t(k)] == j)
                *out += v[size_t(k)];
    });
}

extern "C" int mispec_csr_download(const mispec_csr* A, int32_t* rowptr_host, int32_t* colind_host, double* val_host)
{
    return guarded([&] {
        MISPEC_REQUIRE(A, "mispec_csr_download: NULL argument");
        A->ctx->make_current();
        MISPEC_HIP(hipStreamSynchronize(A->ctx->stream));
        if (A->reordered())
        {
            // hand back the matrix in the caller's order: A = P' B P
            const int64_t n = A->n_rows;
            std::vector<int32_t> rp(size_t(n) + 1), ci(size_t(A->nnz));
            std::vector<double> v(size_t(A->nnz));
            MISPEC_HIP(hipMemcpy(rp.data(), A->rowptr.p, rp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
            if (A->nnz)
            {
                MISPEC_HIP(hipMemcpy(ci.data(), A->colind.p, ci.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
                MISPEC_HIP(hipMemcpy(v.data(), A->val.p, v.size() * sizeof(double), hipMemcpyDeviceToHost));
            }
            std::vector<int32_t> orp, oci;
            std::vector<double> ov;
            permute_csr(n, rp.data(), ci.data(), v.data(), A->inv_host, orp, oci, ov);  // the inverse permutation undoes it
            if (rowptr_host)
                std::memcpy(rowptr_host, orp.data(), orp.size() * sizeof(int32_t));
            if (colind_host && A->nnz)
                std::memcpy(colind_host, oci.data(), oci.size() * sizeof(int32_t));
            if (val_host && A->nnz)
                std::memcpy(val_host, ov.data(), ov.size() * sizeof(double));
            return;
        }
        if (rowptr_host)
            MISPEC_HIP(hipMemcpy(rowptr_host, A->rowptr.p, size_t(A->local_rows() + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (colind_host && A->nnz)
            MISPEC_HIP(hipMemcpy(colind_host, A->colind.p, size_t(A->nnz) * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (val_host && A->nnz)
            MISPEC_HIP(hipMemcpy(val_host, A->val.p, size_t(A->nnz) * sizeof(double), hipMemcpyDeviceToHost));
    });
}

extern "C" int mispec_spmv(const mispec_csr* A, const double* x_dev, double* y_dev)
{
    return guarded([&] {
        MISPEC_REQUIRE(A && x_dev && y_dev, "mispec_spmv: NULL argument");
        A->ctx->make_current();
        launch_spmv(*A, x_dev, y_dev, nullptr);
    });
}

extern "C" int mispec_spmv_host(const mispec_csr* A, const double* x_host, double* y_host)
{
    return mispec_spmm_host(A, x_host, A ? A->n_cols : 0, 1, y_host, A ? A->n_rows : 0);
}

extern "C" int mispec_spmm_host(const mispec_csr* A, const double* X_host, int64_t ldx, int k, double* Y_host, int64_t ldy)
{
    return guarded([&] {
        MISPEC_REQUIRE(A && X_host && Y_host && k >= 0, "mispec_spmm_host: bad argument");
        MISPEC_REQUIRE(A->ctx->world() == 1, "mispec_spmm_host: host-pointer products need an unsharded matrix");  // world 1 + communicator is fine
        MISPEC_REQUIRE(ldx >= A->n_cols && ldy >= A->n_rows, "mispec_spmm_host: leading dimension too small");
        A->ctx->make_current();
        hipStream_t s = A->ctx->stream;
        if (A->stage_x.n < size_t(A->n_cols))
            A->stage_x.alloc(size_t(A->n_cols));
        if (A->stage_y.n < size_t(A->n_rows))
            A->stage_y.alloc(size_t(A->n_rows));
        for (int c = 0; c < k; c++)
        {
            MISPEC_HIP(hipMemcpyAsync(A->stage_x.p, X_host + int64_t(c) * ldx, size_t(A->n_cols) * sizeof(double),
                                      hipMemcpyHostToDevice, s));
            launch_spmv(*A, A->stage_x.p, A->stage_y.p, nullptr);
            MISPEC_HIP(hipMemcpyAsync(Y_host + int64_t(c) * ldy, A->stage_y.p, size_t(A->n_rows) * sizeof(double),
                                      hipMemcpyDeviceToHost, s));
            MISPEC_HIP(hipStreamSynchronize(s));
        }
    });
}

extern "C" int mispec_spmv_time(const mispec_csr* A, const double* x_dev, double* y_dev, int reps, float* ms_per_launch)
{
    return guarded([&] {
        MISPEC_REQUIRE(A && x_dev && y_dev && reps > 0 && ms_per_launch, "mispec_spmv_time: bad argument");
        A->ctx->make_current();
        hipEvent_t e0, e1;
        MISPEC_HIP(hipEventCreate(&e0));
        MISPEC_HIP(hipEventCreate(&e1));
        MISPEC_HIP(hipEventRecord(e0, A->ctx->stream));
        for (int i = 0; i < reps; i++)
            launch_spmv_raw(*A, x_dev, y_dev, nullptr);  // the kernel itself (the stored, possibly reordered matrix)
        MISPEC_HIP(hipEventRecord(e1, A->ctx->stream));
        MISPEC_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        MISPEC_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void) hipEventDestroy(e0);
        (void) hipEventDestroy(e1);
        *ms_per_launch = ms / float(reps);
    });
}
