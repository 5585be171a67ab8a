// Column-blocked SpMV for scattered sparsity patterns (SURVEY.md 8d "M-rand"; north_star's CSR SpMV must hold up on any
// pattern the reference's operators accept: MatOp/SparseSymMatProd.h:83-88, SparseGenMatProd.h:82-87).
//
// Why.  A CSR row sweep over a matrix with uniformly scattered columns issues one 8-byte gather per entry into an 80 MB
// vector; every gather misses the 4 MiB per-XCD L2 and drags a 128-byte line through the fabric: 16x read amplification,
// 0.10 of the HBM roofline (profiles/rounds_1_2/r01g_spmv_patterns.jsonl).  No symmetric reordering helps an expander.
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
// Bound: the gather rate.  Round 3 measured what a device-wide stream of random 8-byte gathers reaches when EVERY gather hits
// the L2 (a two-phase "propagation blocking" variant whose first phase did nothing but stream values in column-block-major order,
// gather x from one or two 512 KiB blocks and stream the products out: 150 M gathers in 1.59 ms, profiles/r05d_*): ~95 G gathers/s,
// i.e. 2.7 clocks per gather and CU — each one moves a 128-byte line from the L2 to the L1.  This kernel's 1.31-1.36 ms for the
// same 150 M gathers (~110 G/s) sits on that same limit whatever its L2 hit rate; only x staged in LDS (<= 16 Ki columns per
// block, hence tiles of ~25-200 entries and a different second phase) would lift it.  The two-phase variants (2.38 / 2.11 ms
// in total) were removed again.
#include "tiles.hpp"
#include "csr.hpp"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace mispec {

namespace {
// The tiles of segments [s0, s1) appended to `out` (entry offsets of the chunks are relative to their segment's first entry, so
// pieces built by different threads concatenate without patching).  seg_len / seg_nchunk: entries and chunks of every segment.
struct TilePiece  // what one thread builds: growable, concatenated into the HostTiles afterwards
{
    std::vector<TileChunk> chunks;
    std::vector<double> val;
    std::vector<uint32_t> idx;
    int64_t padding = 0;
};
bool build_tiles_range(int64_t s0, int64_t s1, int64_t nrows, int64_t ncb, const int32_t* rowptr, const int32_t* colind, const double* val,
                       TilePiece& out, std::vector<int64_t>& seg_len, std::vector<int32_t>& seg_nchunk)
{
    std::vector<int64_t> count(static_cast<size_t>(ncb)), start(size_t(ncb) + 1), fill(static_cast<size_t>(ncb));
    std::vector<uint32_t> tidx;  // entries of the segment, bucketed by tile (row-major inside a tile)
    std::vector<double> tval;
    std::vector<std::pair<int64_t, int64_t>> groups;  // (first entry, length) of every row's run inside the current tile
    for (int64_t s = s0; s < s1; s++)
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
        std::copy(start.begin(), start.end() - 1, fill.begin());
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
        const int64_t seg_first = int64_t(out.val.size());
        const size_t chunk_first = out.chunks.size();
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
                    ch.offset = int32_t(int64_t(out.val.size()) - seg_first);
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
                            out.val.push_back(0.0);
                            out.idx.push_back(kTileSkip);
                        }
                        out.padding += pad;
                        for (int k = 0; k < run; k++)
                        {
                            out.val.push_back(tval[size_t(e + k)]);
                            out.idx.push_back(tidx[size_t(e + k)] | uint32_t(k == 0 ? run : 0));
                        }
                        cnt += pad + run;
                        gi++;
                    }
                    ch.count = uint16_t(cnt);
                    out.chunks.push_back(ch);
                }
                // rows with more entries than the passes so far have emitted stay for the next pass
                size_t keep = 0;
                for (size_t g = 0; g < groups.size(); g++)
                    if (groups[g].second > int64_t(pass + 1) * kTileMaxRun)
                        groups[keep++] = groups[g];
                groups.resize(keep);
            }
        }
        seg_len[size_t(s)] = int64_t(out.val.size()) - seg_first;
        seg_nchunk[size_t(s)] = int32_t(out.chunks.size() - chunk_first);
    }
    return true;
}
}  // namespace

// Segments are independent: the host threads (ingest_threads) build contiguous ranges of them into pieces of their own, which are
// then concatenated in segment order — the image is the one a single thread builds, byte for byte.
bool build_tiles(int64_t nrows, int64_t ncols, const int32_t* rowptr, const int32_t* colind, const double* val, HostTiles& T)
{
    T = HostTiles{};
    const int64_t ncb = (ncols + kTileCols - 1) / kTileCols;
    if (ncb > 65535 || nrows <= 0)
        return false;
    const int64_t nseg = (nrows + kTileRows - 1) / kTileRows;
    T.ncb = ncb;
    std::vector<int64_t> seg_len(static_cast<size_t>(nseg), 0);
    std::vector<int32_t> seg_nchunk(static_cast<size_t>(nseg), 0);
    const int nt = int(std::max<int64_t>(1, std::min<int64_t>(ingest_threads(), nseg / 4)));
    std::vector<TilePiece> piece(static_cast<size_t>(nt));
    std::vector<char> ok(static_cast<size_t>(nt), 1);
    parallel_ranges(nseg, nt, [&](int t, int64_t s0, int64_t s1) {
        const int64_t nnz = rowptr[std::min<int64_t>(s1 * kTileRows, nrows)] - rowptr[s0 * kTileRows];
        piece[size_t(t)].val.reserve(size_t(nnz + nnz / 32 + 64));
        piece[size_t(t)].idx.reserve(size_t(nnz + nnz / 32 + 64));
        ok[size_t(t)] = build_tiles_range(s0, s1, nrows, ncb, rowptr, colind, val, piece[size_t(t)], seg_len, seg_nchunk) ? 1 : 0;
    });
    for (char o : ok)
        if (!o)
        {
            T = HostTiles{};
            return false;
        }
    T.seg_entry.assign(size_t(nseg) + 1, 0);
    T.seg_chunk.assign(size_t(nseg) + 1, 0);
    for (int64_t s = 0; s < nseg; s++)
    {
        T.seg_entry[size_t(s) + 1] = T.seg_entry[size_t(s)] + seg_len[size_t(s)];
        T.seg_chunk[size_t(s) + 1] = T.seg_chunk[size_t(s)] + seg_nchunk[size_t(s)];
    }
    const int64_t total = T.seg_entry[size_t(nseg)];
    T.val.resize_uninitialized(size_t(total) + kTileSlack);  // filled by the threads below: no zero-fill pass
    T.idx.resize_uninitialized(size_t(total) + kTileSlack);
    T.chunks.resize_uninitialized(size_t(T.seg_chunk[size_t(nseg)]));
    std::vector<int64_t> at(static_cast<size_t>(nt) + 1, 0), cat(static_cast<size_t>(nt) + 1, 0);
    for (int t = 0; t < nt; t++)
    {
        at[size_t(t) + 1] = at[size_t(t)] + int64_t(piece[size_t(t)].val.size());
        cat[size_t(t) + 1] = cat[size_t(t)] + int64_t(piece[size_t(t)].chunks.size());
        T.padding += piece[size_t(t)].padding;
    }
    parallel_ranges(nt, nt, [&](int, int64_t t0, int64_t t1) {
        for (int64_t t = t0; t < t1; t++)
        {
            const TilePiece& P = piece[size_t(t)];
            std::copy(P.val.begin(), P.val.end(), T.val.begin() + at[size_t(t)]);
            std::copy(P.idx.begin(), P.idx.end(), T.idx.begin() + at[size_t(t)]);
            std::copy(P.chunks.begin(), P.chunks.end(), T.chunks.begin() + cat[size_t(t)]);
        }
    });
    // slack so that the kernel's unconditional loads (two chunks ahead) never leave the arrays
    for (int k = 0; k < kTileSlack; k++)
    {
        T.val[size_t(total + k)] = 0.0;
        T.idx[size_t(total + k)] = kTileSkip;
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

template <bool EPI>
__global__ __launch_bounds__(kTileThreads) void k_spmv_tiles(const int64_t* __restrict__ seg_entry, const int32_t* __restrict__ seg_chunk,
                                                    const TileChunk* __restrict__ chunks, const double* __restrict__ val,
                                                    const uint32_t* __restrict__ idx, const double* __restrict__ x,
                                                    double* __restrict__ y, int64_t nrows, int nblocks256, int nseg, SpmvEpilogue epi)
{
    __shared__ double acc[kTileRows];  // 64 KiB with the default geometry: two workgroups per CU
    if (EPI && epi.status && *epi.status != 0)
        return;
    const int tid = threadIdx.x;
    constexpr int kPer = kTileChunk / kTileThreads;  // entries per thread and chunk
    const int seg = int(blockIdx.x);
    if (seg >= nseg)
        return;
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
        // (profiles/rounds_1_2/r03t_*): the gathers are bound by the fabric, more of them in flight only evict each other
        const int noff = off + count;
#pragma unroll
        for (int k = 0; k < kPer; k++)
        {
            nv[k] = __builtin_nontemporal_load(val + base + noff + k * kTileThreads + tid);
            nid[k] = __builtin_nontemporal_load(idx + base + noff + k * kTileThreads + tid);
        }
        const int64_t col0 = int64_t(ch.colblock) << kTileColBits;
        double p[kPer];
        int run[kPer], rowk[kPer];
#pragma unroll
        for (int k = 0; k < kPer; k++)
        {
            const bool live = (k * kTileThreads + tid < count) && id[k] != kTileSkip;
            run[k] = live ? int(id[k] & uint32_t(kTileMaxRun)) : 0;
            rowk[k] = int(id[k] >> (kTileColBits + kTileRunBits)) & (kTileRows - 1);
            const int64_t col = live ? col0 + int64_t((id[k] >> kTileRunBits) & uint32_t(kTileCols - 1)) : col0;
            p[k] = live ? rounded_product(v[k], x[col]) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < kPer; k++)
        {
            // head lanes add the products of their run in order: own, then the next lanes' (a run never leaves its wavefront)
            const int row = rowk[k];
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
}

}  // namespace

void upload_tiles(const HostTiles& H, hipStream_t stream, DevTiles& D)
{
    auto up = [&](auto& dst, const auto& src) {
        dst.alloc(src.size());
        if (src.size())
            MISPEC_HIP(hipMemcpyAsync(dst.p, src.data(), src.size() * sizeof(src[0]), hipMemcpyHostToDevice, stream));
    };
    up(D.seg_entry, H.seg_entry);
    up(D.seg_chunk, H.seg_chunk);
    up(D.chunks, H.chunks);
    up(D.val, H.val);
    up(D.idx, H.idx);
    D.nseg = int64_t(H.seg_entry.size()) - 1;
    D.entries = int64_t(H.val.size()) - kTileSlack;
    D.nchunks = int64_t(H.chunks.size());
    D.padding = H.padding;
    D.ncb = H.ncb;
    MISPEC_HIP(hipStreamSynchronize(stream));
}

void launch_spmv_tiles(const DevTiles& T, hipStream_t stream, const double* x, double* y, int64_t nrows, int nblocks256,
                       const SpmvEpilogue* epi, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    const SpmvEpilogue e = epi ? *epi : SpmvEpilogue{};
    const dim3 grid(static_cast<unsigned>(T.nseg)), block(kTileThreads);
#define MISPEC_TILES(E)                                                                                                     \
    do                                                                                                                      \
    {                                                                                                                       \
        if (ev_start && ev_stop)                                                                                            \
            hipExtLaunchKernelGGL((k_spmv_tiles<E>), grid, block, 0, stream, ev_start, ev_stop, 0, T.seg_entry.p, T.seg_chunk.p, \
                                  T.chunks.p, T.val.p, T.idx.p, x, y, nrows, nblocks256, int(T.nseg), e);                  \
        else                                                                                                                \
            hipLaunchKernelGGL((k_spmv_tiles<E>), grid, block, 0, stream, T.seg_entry.p, T.seg_chunk.p, T.chunks.p, T.val.p, \
                               T.idx.p, x, y, nrows, nblocks256, int(T.nseg), e);                                          \
    } while (0)
    if (epi)
        MISPEC_TILES(true);
    else
        MISPEC_TILES(false);
#undef MISPEC_TILES
    MISPEC_HIP(hipGetLastError());
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
