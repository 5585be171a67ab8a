// Column-blocked SpMV for scattered sparsity patterns (SURVEY.md 8d "M-rand"; north_star's CSR SpMV must hold up on any
// pattern the reference's operators accept: MatOp/SparseSymMatProd.h:83-88, SparseGenMatProd.h:82-87).
//
// Why.  A CSR row sweep over a matrix with uniformly scattered columns issues one 8-byte gather per entry into an 80 MB
// vector; every gather misses the 4 MiB per-XCD L2 and drags a 128-byte line through the fabric: 16x read amplification,
// 0.10 of the HBM roofline (profiles/r01g_spmv_patterns.jsonl).  No symmetric reordering helps an expander.
//
// Layout.  The rows are cut into SEGMENTS of 8192 rows, the columns into BLOCKS of 65536 columns (512 KiB of x).  A TILE is
// the part of a segment inside one column block; a segment stores its tiles one after the other (ascending block), every
// entry as (fp64 value, 32-bit index) = 12 bytes like CSR with int32 indices, the index packing the row inside the segment
// (13 bits), the column inside the block (16 bits) and the length of the row's run inside the tile (3 bits).  (The geometry
// is a set of compile-time constants, tiles.hpp; the sweep that chose it is recorded there.)
//
// Kernel.  One workgroup (512 threads) per segment, the segment's 8192 partial sums in LDS.  It walks its tiles in block
// order, so that at any moment all the workgroups resident on an XCD gather from the same one or two 1 MiB pieces of x,
// which stay in that XCD's L2 (fewer, larger workgroups drift apart less: 64 per XCD): x is read from HBM / Infinity Cache once per XCD and generation of workgroups instead of
// once per entry.  Inside a tile the entries are sorted by row, every row's entries (ascending column) are consecutive and
// never straddle a wavefront; the lane holding the first entry of a run collects the products of the run from its
// neighbours (shuffles) and adds them to the row's LDS accumulator one after the other.  A row's products are therefore
// added in ascending column order — the CSR storage order — and, rounded individually, the result is BIT-IDENTICAL to the
// CSR kernels and to the oracle's row-dot.  One barrier per <= 1024 entries; the next chunk's entries are in flight while
// the current one is gathered and accumulated.
// Bound: HBM for the 12 nnz bytes of the stream + L2 for the gathers.
#include "tiles.hpp"
#include "csr.hpp"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace mispec {

bool build_tiles(int64_t nrows, int64_t ncols, const int32_t* rowptr, const int32_t* colind, const double* val, HostTiles& T)
{
    T = HostTiles{};
    const int64_t ncb = (ncols + kTileCols - 1) / kTileCols;
    if (ncb > 65535 || nrows <= 0)
        return false;
    const int64_t nseg = (nrows + kTileRows - 1) / kTileRows;
    T.ncb = ncb;
    T.seg_entry.assign(size_t(nseg) + 1, 0);
    T.seg_chunk.assign(size_t(nseg) + 1, 0);
    const int64_t nnz = rowptr[nrows] - rowptr[0];
    T.val.reserve(size_t(nnz + nnz / 32));
    T.idx.reserve(size_t(nnz + nnz / 32));
    std::vector<int64_t> count(static_cast<size_t>(ncb)), start(size_t(ncb) + 1);
    std::vector<uint32_t> tidx;  // entries of the segment, bucketed by tile (row-major inside a tile)
    std::vector<double> tval;
    std::vector<std::pair<int64_t, int64_t>> groups;  // (first entry, length) of every row's run inside the current tile
    for (int64_t s = 0; s < nseg; s++)
    {
        const int64_t r0 = s * kTileRows, r1 = std::min<int64_t>(r0 + kTileRows, nrows);
        std::fill(count.begin(), count.end(), 0);
        for (int64_t r = r0; r < r1; r++)
            for (int32_t p = rowptr[r]; p < rowptr[r + 1]; p++)
            {
                if (p > rowptr[r] && colind[p] <= colind[p - 1])
                    return false;  // unsorted row or duplicate entry: the run order would not be the CSR order
                count[size_t(colind[p] >> kTileColBits)]++;
            }
        start[0] = 0;
        for (int64_t c = 0; c < ncb; c++)
            start[size_t(c) + 1] = start[size_t(c)] + count[size_t(c)];
        const int64_t seg_nnz = start[size_t(ncb)];
        tidx.resize(size_t(seg_nnz));
        tval.resize(size_t(seg_nnz));
        std::vector<int64_t> fill(start.begin(), start.end() - 1);
        for (int64_t r = r0; r < r1; r++)
        {
            int32_t p = rowptr[r];
            while (p < rowptr[r + 1])
            {
                const int64_t c = colind[p] >> kTileColBits;
                int32_t q = p + 1;
                while (q < rowptr[r + 1] && (colind[q] >> kTileColBits) == c)
                    q++;
                const int run = q - p;
                for (int k = 0; k < run; k++)
                {
                    const int64_t dst = fill[size_t(c)]++;
                    tidx[size_t(dst)] = (uint32_t(r - r0) << (kTileColBits + kTileRunBits)) |
                                        (uint32_t(colind[p + k] & (kTileCols - 1)) << kTileRunBits);  // run bits set at emission
                    tval[size_t(dst)] = val[p + k];
                }
                p = q;
            }
        }
        // emit: chunks of at most kTileChunk entries of one tile; a run never straddles a 64-entry group of its chunk
        const int64_t seg_first = int64_t(T.val.size());
        T.seg_entry[size_t(s)] = seg_first;
        T.seg_chunk[size_t(s)] = int32_t(T.chunks.size());
        for (int64_t c = 0; c < ncb; c++)
        {
            // A row's entries inside the tile are consecutive.  Pass 0 emits the first (up to) 7 of every row, pass 1 the next
            // 7 of the rows that have more, ... — every pass in chunks of its own, so that the barrier between chunks keeps a
            // row's additions in column order and no two run heads of one row ever share a chunk.
            groups.clear();
            for (int64_t e = start[size_t(c)]; e < start[size_t(c) + 1];)
            {
                int64_t q = e + 1;
                const uint32_t row = tidx[size_t(e)] >> (kTileColBits + kTileRunBits);
                while (q < start[size_t(c) + 1] && (tidx[size_t(q)] >> (kTileColBits + kTileRunBits)) == row)
                    q++;
                groups.emplace_back(e, q - e);
                e = q;
            }
            for (int pass = 0; !groups.empty(); pass++)
            {
                size_t gi = 0;
                while (gi < groups.size())
                {
                    TileChunk ch;
                    ch.offset = int32_t(int64_t(T.val.size()) - seg_first);
                    ch.colblock = uint16_t(c);
                    int cnt = 0;
                    while (gi < groups.size())
                    {
                        const int64_t e = groups[gi].first + int64_t(pass) * kTileMaxRun;
                        const int run = int(std::min<int64_t>(groups[gi].second - int64_t(pass) * kTileMaxRun, kTileMaxRun));
                        const int room = 64 - (cnt & 63);
                        int pad = 0;
                        if (run > room)
                        {
                            if (cnt + room >= kTileChunk)
                                break;  // the chunk ends here; the run opens the next one
                            pad = room;
                        }
                        if (cnt + pad + run > kTileChunk)
                            break;
                        for (int k = 0; k < pad; k++)
                        {
                            T.val.push_back(0.0);
                            T.idx.push_back(kTileSkip);
                        }
                        T.padding += pad;
                        for (int k = 0; k < run; k++)
                        {
                            T.val.push_back(tval[size_t(e + k)]);
                            T.idx.push_back(tidx[size_t(e + k)] | uint32_t(k == 0 ? run : 0));
                        }
                        cnt += pad + run;
                        gi++;
                    }
                    ch.count = uint16_t(cnt);
                    T.chunks.push_back(ch);
                }
                // rows with more entries than the passes so far have emitted stay for the next pass
                size_t keep = 0;
                for (size_t g = 0; g < groups.size(); g++)
                    if (groups[g].second > int64_t(pass + 1) * kTileMaxRun)
                        groups[keep++] = groups[g];
                groups.resize(keep);
            }
        }
    }
    T.seg_entry[size_t(nseg)] = int64_t(T.val.size());
    T.seg_chunk[size_t(nseg)] = int32_t(T.chunks.size());
    // slack so that the kernel's unconditional loads (two chunks ahead) never leave the arrays
    for (int k = 0; k < kTileSlack; k++)
    {
        T.val.push_back(0.0);
        T.idx.push_back(kTileSkip);
    }
    return true;
}

void tiles_spmv_host(const HostTiles& T, int64_t nrows, const double* x, double* y)
{
    const int64_t nseg = int64_t(T.seg_entry.size()) - 1;
    std::vector<double> acc(kTileRows);
    for (int64_t s = 0; s < nseg; s++)
    {
        std::fill(acc.begin(), acc.end(), 0.0);
        for (int32_t ci = T.seg_chunk[size_t(s)]; ci < T.seg_chunk[size_t(s) + 1]; ci++)
        {
            const TileChunk& ch = T.chunks[size_t(ci)];
            const int64_t base = T.seg_entry[size_t(s)] + ch.offset;
            for (int k = 0; k < ch.count; k++)
            {
                const uint32_t id = T.idx[size_t(base + k)];
                if (id == kTileSkip)
                    continue;
                const int64_t col = int64_t(ch.colblock) * kTileCols + ((id >> kTileRunBits) & uint32_t(kTileCols - 1));
                // continuation entries follow their head: adding in storage order IS the run order
                const volatile double p = T.val[size_t(base + k)] * x[col];
                acc[size_t(id >> (kTileColBits + kTileRunBits))] += p;
            }
        }
        const int64_t r0 = s * kTileRows;
        for (int64_t r = r0; r < std::min<int64_t>(r0 + kTileRows, nrows); r++)
            y[r] = acc[size_t(r - r0)];
    }
}

namespace {
__device__ __forceinline__ double tile_wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, 64);
    return v;
}
// product rounded on its own (no FMA with the accumulation): the sums are those of the CSR kernels, bit for bit
__device__ __forceinline__ double rounded_product(double a, double b)
{
#pragma clang fp contract(off)
    const double p = a * b;
    return p;
}
__device__ __forceinline__ double rounded_add(double a, double b)
{
#pragma clang fp contract(off)
    const double s = a + b;
    return s;
}

// Loose barrier among the workgroups of one group (the ~160 workgroups an XCD holds: blockIdx % 8): keeps the sweeps over
// the column blocks in step, so that the group's gathers stay inside the 1-3 MiB of x its L2 holds.  Thread 0 arrives on a
// monotonic counter and polls it; a wait that outlasts `spin_cap` polls gives up for the rest of the launch (a workgroup
// that is not resident can therefore never hang the others — only the locality is lost).
struct TileSync
{
    unsigned int* counter;  // 8 counters (one per XCD, 128 bytes apart), zero at launch
    int period;             // barrier every `period` column blocks; 0: no synchronisation (one segment per workgroup)
    int group_size;         // workgroups per group
    int sweeps;             // segments per workgroup (idle sweeps only pass the barriers)
    int ncb;                // column blocks
    int spin_cap;
    unsigned int zero;      // 0 (a value the compiler cannot fold)
    int xload;              // how x is gathered: 0 plain loads, 1 non-temporal, 2 system scope (L2 bypass) — MISPEC_TILES_XLOAD
};

// The counter lives in the L2 of the group's own XCD: read-modify-writes without the agent-scope cache-bypass bits are
// executed there (atomics never run in the per-CU L1), so arriving and polling cost an L2 round trip instead of a trip to
// memory; the XCDs' L2s are not coherent with each other, which is why every XCD has its own counter and only its own
// workgroups (hardware register XCC_ID) touch it.
__device__ __forceinline__ void tile_group_barrier(const TileSync& ts, int xcd, unsigned int target, bool& give_up)
{
    if (threadIdx.x == 0 && !give_up)
    {
        unsigned int* c = ts.counter + xcd * 32;  // one 128-byte line per counter
        __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        int spins = 0;
        // polled with a read-modify-write of a run-time zero: a plain (even atomic) load could be served by this CU's L1 forever
        while (__hip_atomic_fetch_add(c, ts.zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target)
        {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > ts.spin_cap)
            {
                give_up = true;
                break;
            }
        }
    }
}

template <int XL>
__device__ __forceinline__ double tile_load_x(const double* p)
{
    if (XL == 1)
        return __builtin_nontemporal_load(p);
    if (XL == 2)
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return *p;
}

template <bool EPI, int XL>
__global__ __launch_bounds__(kTileThreads) void k_spmv_tiles(const int64_t* __restrict__ seg_entry, const int32_t* __restrict__ seg_chunk,
                                                    const TileChunk* __restrict__ chunks, const double* __restrict__ val,
                                                    const uint32_t* __restrict__ idx, const double* __restrict__ x, double* __restrict__ y,
                                                    int64_t nrows, int nblocks256, int nseg, SpmvEpilogue epi, TileSync ts)
{
    __shared__ double acc[kTileRows];  // 64 KiB with the default geometry: two workgroups per CU
    if (EPI && epi.status && *epi.status != 0)
        return;
    const int tid = threadIdx.x;
    constexpr int kPer = kTileChunk / kTileThreads;  // entries per thread and chunk
    bool give_up = false;         // thread 0 only
    unsigned int arrivals = 0;    // barriers passed so far (all sweeps)
    // XCC_ID: the XCD this workgroup runs on (hwreg 20, bits 3:0); workgroups are dealt round-robin, gridDim.x / 8 per XCD
    const int xcd = ts.period > 0 ? int(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) : 0;
    const int nsweep = ts.period > 0 ? ts.sweeps : 1;
    for (int sweep = 0; sweep < nsweep; sweep++)
    {
        const int seg = int(blockIdx.x) + sweep * int(gridDim.x);
        int next_sync = ts.period;  // first column block that lies behind the next barrier
        if (seg < nseg)
        {
            for (int r = tid; r < kTileRows; r += kTileThreads)
                acc[r] = 0.0;
            const int c0 = seg_chunk[seg], c1 = seg_chunk[seg + 1];
            const int64_t base = seg_entry[seg];
            double v[kPer], nv[kPer];
            uint32_t id[kPer], nid[kPer];
            int off = (c0 < c1) ? chunks[c0].offset : 0;
#pragma unroll
            for (int k = 0; k < kPer; k++)
            {
                v[k] = __builtin_nontemporal_load(val + base + off + k * kTileThreads + tid);  // streamed once: keep x in the L2
                id[k] = __builtin_nontemporal_load(idx + base + off + k * kTileThreads + tid);
            }
            __syncthreads();
            for (int ci = c0; ci < c1; ci++)
            {
                const TileChunk ch = chunks[ci];
                const int count = ch.count;
                // the next chunk's entries follow this chunk's: issue their loads now (the arrays end with slack).  Also issuing
                // the NEXT chunk's x gathers here (three chunks in flight per thread) measured 1.46 against 1.33 ms on M-rand
                // (profiles/r03t_*): the gathers are bound by the fabric, more of them in flight only evict each other
                const int noff = off + count;
#pragma unroll
                for (int k = 0; k < kPer; k++)
                {
                    nv[k] = __builtin_nontemporal_load(val + base + noff + k * kTileThreads + tid);
                    nid[k] = __builtin_nontemporal_load(idx + base + noff + k * kTileThreads + tid);
                }
                if (ts.period > 0 && next_sync <= int(ch.colblock))
                {
                    while (next_sync <= int(ch.colblock))  // do not run ahead of the group into the next piece of x
                    {
                        arrivals++;
                        tile_group_barrier(ts, xcd, arrivals * unsigned(ts.group_size), give_up);
                        next_sync += ts.period;
                    }
                    __syncthreads();  // the whole workgroup waits for thread 0's wait
                }
                const int64_t col0 = int64_t(ch.colblock) << kTileColBits;
                double p[kPer];
                int run[kPer];
#pragma unroll
                for (int k = 0; k < kPer; k++)
                {
                    const bool live = (k * kTileThreads + tid < count) && id[k] != kTileSkip;
                    run[k] = live ? int(id[k] & uint32_t(kTileMaxRun)) : 0;
                    const int64_t col = live ? col0 + int64_t((id[k] >> kTileRunBits) & uint32_t(kTileCols - 1)) : col0;
                    p[k] = live ? rounded_product(v[k], tile_load_x<XL>(x + col)) : 0.0;
                }
#pragma unroll
                for (int k = 0; k < kPer; k++)
                {
                    // head lanes add the products of their run in order: own, then the next lanes' (a run never leaves its wavefront)
                    const int row = int(id[k] >> (kTileColBits + kTileRunBits)) & (kTileRows - 1);
                    double a = (run[k] > 0) ? acc[row] : 0.0;
                    a = rounded_add(a, p[k]);
                    for (int j = 1; j < kTileMaxRun; j++)
                    {
                        if (__ballot(run[k] > j) == 0ull)
                            break;
                        const double pj = __shfl_down(p[k], j, 64);
                        if (run[k] > j)
                            a = rounded_add(a, pj);
                    }
                    if (run[k] > 0)
                        acc[row] = a;
                }
                __syncthreads();  // the next chunk may address the same rows from other lanes
                off = noff;
#pragma unroll
                for (int k = 0; k < kPer; k++)
                {
                    v[k] = nv[k];
                    id[k] = nid[k];
                }
            }
            // rows of the segment -> y, in the 256-row blocks of the CSR kernels (identical alpha partial records)
            const int64_t row0 = int64_t(seg) * kTileRows;
#pragma unroll 1
            for (int j = 0; j < kTileRows / 256; j++)
            {
                const int64_t row = row0 + j * 256 + tid;
                const int64_t blk = row0 / 256 + j;
                if (blk >= nblocks256)
                    break;
                double contrib = 0.0;
                if (tid < 256 && row < nrows)
                {
                    double yv = acc[j * 256 + tid];
                    if (EPI)
                    {
                        if (epi.v_prev)
                            yv -= (epi.h_prev_dev ? *epi.h_prev_dev : epi.h_prev) * epi.v_prev[row];  // Lanczos.h:139
                        contrib = epi.v_rows[row] * yv;                                                 // Lanczos.h:142
                    }
                    y[row] = yv;
                }
                if (EPI)
                {
                    // cross-wave sum through the accumulator slots of this block, which every thread has read by now
                    const double t = tile_wave_sum(contrib);
                    __syncthreads();
                    double* red = acc + j * 256;
                    if ((tid & 63) == 0 && tid < 256)
                        red[tid >> 6] = t;
                    __syncthreads();
                    if (tid == 0)
                        epi.partials[blk] = (red[0] + red[1]) + (red[2] + red[3]);
                }
            }
            __syncthreads();  // acc is reused by the next sweep
        }
        // the barriers of this sweep that the segment's chunks did not reach (empty trailing tiles, idle sweep)
        if (ts.period > 0)
            while (next_sync < ts.ncb + ts.period)
            {
                arrivals++;
                tile_group_barrier(ts, xcd, arrivals * unsigned(ts.group_size), give_up);
                next_sync += ts.period;
            }
    }
}
}  // namespace

void upload_tiles(const HostTiles& H, hipStream_t stream, DevTiles& D)
{
    auto up = [&](auto& dst, const auto& src) {
        dst.alloc(src.size());
        MISPEC_HIP(hipMemcpyAsync(dst.p, src.data(), src.size() * sizeof(src[0]), hipMemcpyHostToDevice, stream));
    };
    up(D.seg_entry, H.seg_entry);
    up(D.seg_chunk, H.seg_chunk);
    up(D.chunks, H.chunks);
    up(D.val, H.val);
    up(D.idx, H.idx);
    MISPEC_HIP(hipStreamSynchronize(stream));
    D.nseg = int64_t(H.seg_entry.size()) - 1;
    D.entries = int64_t(H.val.size()) - kTileSlack;
    D.nchunks = int64_t(H.chunks.size());
    D.padding = H.padding;
    D.ncb = H.ncb;
    D.sync_counters.alloc(8 * 32);
    MISPEC_HIP(hipMemset(D.sync_counters.p, 0, 8 * 32 * sizeof(unsigned int)));
}

void launch_spmv_tiles(const DevTiles& T, hipStream_t stream, const double* x, double* y, int64_t nrows, int nblocks256,
                       const SpmvEpilogue* epi, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    const SpmvEpilogue e = epi ? *epi : SpmvEpilogue{};
    // sync_period = k > 0: persistent workgroups (as many as are resident at once), each sweeping several segments, with a
    // loose barrier per XCD group every k column blocks so that the sweeps stay in step and the gathers stay in the L2;
    // 0: one workgroup per segment, free-running.
    // sync_period: per matrix, chosen by calibrate_tiles() at ingest (or forced by MISPEC_TILES_SYNC=k)
    const int period = T.sync_period;
    static const int xload = getenv("MISPEC_TILES_XLOAD") ? atoi(getenv("MISPEC_TILES_XLOAD")) : 0;
    TileSync ts{nullptr, 0, 0, 1, int(T.ncb), 0, 0u, xload};
    dim3 grid(static_cast<unsigned>(T.nseg)), block(kTileThreads);
    if (period > 0 && T.sync_counters.p)
    {
        static int resident = 0;
        if (!resident)
        {
            int per_cu = 0, dev = 0;
            hipDeviceProp_t prop;
            MISPEC_HIP(hipGetDevice(&dev));
            MISPEC_HIP(hipGetDeviceProperties(&prop, dev));
            int per_cu2 = 0;
            MISPEC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&k_spmv_tiles<true, 0>), kTileThreads, 0));
            MISPEC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu2, reinterpret_cast<const void*>(&k_spmv_tiles<false, 0>), kTileThreads, 0));
            // one below what the occupancy calculator allows (5 x 32 KiB is the whole LDS of a CU: measured, the fifth
            // workgroup is not resident), overridable for experiments
            per_cu = std::max(1, std::min(per_cu, per_cu2) - 1);
            if (getenv("MISPEC_TILES_WG_PER_CU"))
                per_cu = std::max(1, atoi(getenv("MISPEC_TILES_WG_PER_CU")));
            resident = std::max(8, per_cu * prop.multiProcessorCount / 8 * 8);
        }
        const int g = int(std::min<int64_t>(resident, (T.nseg + 7) / 8 * 8));
        grid = dim3(static_cast<unsigned>(g));
        ts.counter = T.sync_counters.p;
        ts.period = period;
        ts.group_size = g / 8;
        ts.sweeps = int((T.nseg + g - 1) / g);
        ts.spin_cap = 20000;  // ~ a few ms of polling: far beyond any healthy wait
        MISPEC_HIP(hipMemsetAsync(T.sync_counters.p, 0, 8 * 32 * sizeof(unsigned int), stream));
    }
#define MISPEC_TILES(E, X)                                                                                                  \
    do                                                                                                                      \
    {                                                                                                                       \
        if (ev_start && ev_stop)                                                                                            \
            hipExtLaunchKernelGGL((k_spmv_tiles<E, X>), grid, block, 0, stream, ev_start, ev_stop, 0, T.seg_entry.p, T.seg_chunk.p, \
                                  T.chunks.p, T.val.p, T.idx.p, x, y, nrows, nblocks256, int(T.nseg), e, ts);              \
        else                                                                                                                \
            hipLaunchKernelGGL((k_spmv_tiles<E, X>), grid, block, 0, stream, T.seg_entry.p, T.seg_chunk.p, T.chunks.p, T.val.p, \
                               T.idx.p, x, y, nrows, nblocks256, int(T.nseg), e, ts);                                      \
    } while (0)
#define MISPEC_TILES_X(E)        \
    do                           \
    {                            \
        if (xload == 1)          \
            MISPEC_TILES(E, 1);  \
        else if (xload == 2)     \
            MISPEC_TILES(E, 2);  \
        else                     \
            MISPEC_TILES(E, 0);  \
    } while (0)
    if (epi)
        MISPEC_TILES_X(true);
    else
        MISPEC_TILES_X(false);
#undef MISPEC_TILES_X
#undef MISPEC_TILES
    MISPEC_HIP(hipGetLastError());
}

// Pick the launch variant of this matrix by measurement: free-running workgroups against persistent ones that meet every
// quarter of the column sweep (measured at n = 1e7: 1.87 vs 1.73-1.76 ms; frequent barriers lose, profiles/r02_mrand_variants.jsonl).
// The persistent variant relies on every workgroup of its grid being resident; if that ever fails its barriers time out and
// the timing here says so — the free-running kernel is then kept.  MISPEC_TILES_SYNC=k forces a period (0: free-running).
void calibrate_tiles(DevTiles& T, hipStream_t stream, int64_t nrows, int64_t ncols, int nblocks256)
{
    T.sync_period = 0;
    if (const char* e = getenv("MISPEC_TILES_SYNC"))
    {
        T.sync_period = std::max(0, atoi(e));
        return;
    }
    if (T.nseg < 2048 || T.ncb < 16)  // fewer segments than two generations of resident workgroups: nothing to keep in step
        return;
    DevBuf<double> x, y;
    x.alloc(size_t(ncols) + 2);
    y.alloc(size_t(nrows) + 2);
    MISPEC_HIP(hipMemsetAsync(x.p, 0, x.n * sizeof(double), stream));
    hipEvent_t e0, e1;
    MISPEC_HIP(hipEventCreate(&e0));
    MISPEC_HIP(hipEventCreate(&e1));
    const int candidate = int(std::max<int64_t>(1, T.ncb / 4));
    float best_ms = 0.f;
    int best = 0;
    for (int variant = 0; variant < 2; variant++)
    {
        T.sync_period = variant ? candidate : 0;
        launch_spmv_tiles(T, stream, x.p, y.p, nrows, nblocks256, nullptr, nullptr, nullptr);  // warm-up
        MISPEC_HIP(hipEventRecord(e0, stream));
        for (int r = 0; r < 3; r++)
            launch_spmv_tiles(T, stream, x.p, y.p, nrows, nblocks256, nullptr, nullptr, nullptr);
        MISPEC_HIP(hipEventRecord(e1, stream));
        MISPEC_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        MISPEC_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (variant == 0 || ms < 0.97f * best_ms)
        {
            best_ms = ms;
            best = T.sync_period;
        }
    }
    T.sync_period = best;
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
}

}  // namespace mispec

// Host-only test hook (no device): builds the tiles of a CSR matrix and multiplies with them on the host in the kernel's
// summation order.  *built = 0 when the format does not apply to this matrix.  stats: [entries incl. padding, padding, chunks]
extern "C" int mispec_tiles_spmv_host(int64_t nrows, int64_t ncols, const int32_t* rowptr, const int32_t* colind, const double* val,
                                      const double* x, double* y, int* built, int64_t* stats)
{
    return mispec::guarded([&] {
        MISPEC_REQUIRE(rowptr && x && y && built, "mispec_tiles_spmv_host: NULL argument");
        mispec::HostTiles T;
        *built = mispec::build_tiles(nrows, ncols, rowptr, colind, val, T) ? 1 : 0;
        if (!*built)
            return;
        mispec::tiles_spmv_host(T, nrows, x, y);
        if (stats)
        {
            stats[0] = int64_t(T.val.size()) - mispec::kTileSlack;
            stats[1] = T.padding;
            stats[2] = int64_t(T.chunks.size());
        }
    });
}
