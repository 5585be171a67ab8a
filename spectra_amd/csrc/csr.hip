// CSR ingest, synthetic matrix generation, the format dispatch and the fp64 CSR-stream SpMV kernel for gfx950.
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
//   (the two variants below live in files of their own since round 6: csr_dia.hip and csr_win.hip; launch_spmv_raw here picks the
//   format and hands an SpmvLaunch to launch_spmv_dia / launch_spmv_csr_win; shared device helpers: csr_kernels.hpp)
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
#include "csr_kernels.hpp"
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
    SpmvLaunch L{grid, block, nloc, nblocks, epi, e, ev_start, ev_stop, x_dev, y_dev};
    if (format == 2)
    {
        launch_spmv_dia(A, L);  // csr_dia.hip
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
        launch_spmv_csr_win(A, L);  // csr_win.hip
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
