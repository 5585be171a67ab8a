// The "staged" SpMV format for scattered patterns (staged.hpp): y = A x in two streaming kernels with x and y in LDS.
// Replaces, for such matrices, the product of SparseSymMatProd / SparseGenMatProd::perform_op (MatOp/SparseSymMatProd.h:83-88,
// SparseGenMatProd.h:82-87); same results as every other format of this library and as the oracle's CSR row sum, bit for bit.
//
// Why two phases.  A one-phase kernel has to gather x (or scatter into y) through the memory hierarchy once per entry, and for
// uniformly scattered columns that costs an L1 miss each: the device sustains about 110 G such gathers per second, 1.3-1.4 ms for
// 1.5e8 entries, whatever the rest of the kernel does.  LDS serves random 8-byte reads an order of magnitude
// faster, but only from 64 KiB windows: so phase 1 walks the entries column block by column block (x window in LDS, products
// written back as a stream) and phase 2 walks them row bin by row bin (y window in LDS, products read as short contiguous runs).
// The price is the product array's round trip: 28 bytes per entry instead of 12.
#include "staged.hpp"

#include "csr.hpp"

#include <algorithm>
#include <cstring>

namespace mispec {

namespace {

typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double st_wave_sum(double v)  // the reduction tree of the CSR kernels' records
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ void st_store(v2d p, double* at)  // streamed: read back by phase 2 only after 1.2 GB more (plain stores
{                                                             // measured the same, profiles/r07g)
    __builtin_nontemporal_store(p, reinterpret_cast<v2d*>(at));
}

// ---- phase 1: products, column block by column block ------------------------------------------------------------------------------
__global__ __launch_bounds__(kStThreads) void k_staged_products(const StPiece* __restrict__ pieces, const double* __restrict__ val,
                                                                 const uint16_t* __restrict__ lcol, const double* __restrict__ x,
                                                                 double* __restrict__ prod, int64_t ncols, const int* status)
{
    __shared__ double xs[kStCols];  // 64 KiB: two workgroups per CU
    if (status && *status != 0)
        return;
    const StPiece pc = pieces[blockIdx.x];
    const int tid = threadIdx.x;
    const int64_t c0 = int64_t(pc.colblock) << kStColBits;
#pragma unroll
    for (int k = 0; k < kStCols / kStThreads; k++)
    {
        const int64_t col = c0 + k * kStThreads + tid;
        xs[k * kStThreads + tid] = (col < ncols) ? x[col] : 0.0;
    }
    __syncthreads();
    // two entries per thread and step (16-byte value loads, 4-byte index loads, 16-byte product stores), four steps in flight
    constexpr int64_t kStep = 2 * kStThreads;
    int64_t i = pc.begin + 2 * tid;
    for (; i + 3 * kStep < pc.end; i += 4 * kStep)
    {
        v2d a[4];
        uint32_t c[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            a[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(val + i + u * kStep));
            c[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(lcol + i + u * kStep));
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            v2d p;
            p.x = a[u].x * xs[c[u] & 0xFFFFu];
            p.y = a[u].y * xs[c[u] >> 16];
            st_store(p, prod + i + u * kStep);
        }
    }
    for (; i < pc.end; i += kStep)
    {
        const v2d a = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(val + i));
        const uint32_t c = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(lcol + i));
        v2d p;
        p.x = a.x * xs[c & 0xFFFFu];
        p.y = a.y * xs[c >> 16];
        st_store(p, prod + i);
    }
}

// ---- phase 2: row sums, bin by bin ------------------------------------------------------------------------------------------------
struct StLoad  // what a thread holds of one batch
{
    double p;
    uint32_t rr;  // row | rank << kStRowBits, or 0xFFFFFFFF: no entry
};

// (profiles/r07r: 8 chunk loads = 4 batches in flight; with one chunk per wavefront and batch, r07g, eight were too many)
constexpr int kStAhead = 8;                 // chunk loads of a wavefront in flight while a batch is being added
constexpr int kStDescAhead = 2 * kStAhead;  // ... and chunk descriptors (the entry loads depend on them)
static_assert(kStAhead % kStPerWave == 0, "staged format: the ring holds whole batches");

// A batch = one chunk per wavefront (<= 64 entries that are contiguous in phase-1 order, i.e. a piece of one bin's share of one
// column block); desc = phase-1 position | entries << 32 | rounds of the batch << 40.  Nothing of a batch passes through LDS
// tables and no load sits between two barriers: per batch the workgroup only meets for the rank rounds.
template <bool EPI>
__global__ __launch_bounds__(kStRowThreads) void k_staged_rows(const int32_t* __restrict__ bin_batch, const uint64_t* __restrict__ desc,
                                                             const uint16_t* __restrict__ rowrank, const double* __restrict__ prod,
                                                             double* __restrict__ y, int64_t nrows, int nblocks256, int bin_rows, SpmvEpilogue epi)
{
    __shared__ double acc[kStRows];  // 64 KiB: two workgroups per CU
    __shared__ double red[(kStRows / kStRowThreads) * (kStRowThreads / 64)];  // wave sums of the epilogue
    if (EPI && epi.status && *epi.status != 0)
        return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bin = int(blockIdx.x);
#pragma unroll
    for (int k = 0; k < kStRows / kStRowThreads; k++)
        acc[k * kStRowThreads + tid] = 0.0;
    // chunk slots of this wavefront: kStPerWave per batch
    const int q0 = bin_batch[bin] * kStPerWave, q1 = bin_batch[bin + 1] * kStPerWave;
    constexpr int kWaves = kStRowThreads / 64;

    auto load_desc = [&](int q) -> uint64_t { return (q < q1) ? desc[int64_t(q) * kWaves + w] : 0ull; };
    auto fetch = [&](uint64_t d) {
        StLoad L;
        L.p = 0.0;
        const int cnt = int((d >> 32) & 0xFFu);
        uint32_t rr = 0x1FFFFu;  // no entry: a rank no round reaches (bit 16)
        static_assert(kStRowBits + kStRankBits == 16 && kStRows % kStRowThreads == 0 && kStRowThreads % 256 == 0, "staged format: field layout");
        if (lane < cnt)
        {
            L.p = __builtin_nontemporal_load(prod + (int64_t(uint32_t(d)) + lane));
            rr = uint32_t(__builtin_nontemporal_load(rowrank + (int64_t(uint32_t(d)) + lane)));  // stored at the entry's phase-1 position
        }
        L.rr = rr | (uint32_t((d >> 40) & 0xFFu) << 20);
        return L;
    };

    uint64_t dq[kStDescAhead];
    StLoad ring[kStAhead];
#pragma unroll
    for (int d = 0; d < kStDescAhead; d++)
        dq[d] = load_desc(q0 + d);
#pragma unroll
    for (int d = 0; d < kStAhead; d++)
        ring[d] = fetch(dq[d]);
    __syncthreads();  // accumulators zeroed
    for (int q = q0; q < q1; q += kStDescAhead)
    {
#pragma unroll
        for (int ub = 0; ub < kStDescAhead / kStPerWave; ub++)
        {
            const int qq = q + ub * kStPerWave;
            if (qq >= q1)
                break;
            StLoad cur[kStPerWave];
#pragma unroll
            for (int c = 0; c < kStPerWave; c++)
            {
                const int u = ub * kStPerWave + c;
                cur[c] = ring[u % kStAhead];
                ring[u % kStAhead] = fetch(dq[(u + kStAhead) % kStDescAhead]);
                dq[u] = load_desc(qq + c + kStDescAhead);
            }
            const int rounds = __builtin_amdgcn_readfirstlane(int(cur[0].rr >> 20));  // the same in every descriptor of the batch
            for (int r = 0; r < rounds; r++)
            {
#pragma unroll
                for (int c = 0; c < kStPerWave; c++)
                {
                    const int row = int(cur[c].rr & uint32_t(kStRows - 1));
                    const int rank = int((cur[c].rr >> kStRowBits) & uint32_t((2 << kStRankBits) - 1));  // > kStMaxRank: no entry
                    if (rank == r)
                    {
#pragma clang fp contract(off)
                        acc[row] = acc[row] + cur[c].p;  // entries of one rank address distinct rows
                    }
                }
                __syncthreads();
            }
        }
    }
    if (q0 == q1)
        __syncthreads();

    // rows of the bin -> y, in the 256-row records of the CSR kernels (identical alpha partials: wave sums by the same shuffle
    // tree, then (w0 + w1) + (w2 + w3) per record)
    const int64_t row0 = int64_t(bin) * bin_rows;
    constexpr int kWavesPer = kStRowThreads / 64;
#pragma unroll
    for (int j = 0; j < kStRows / kStRowThreads; j++)
    {
        const int lr = j * kStRowThreads + tid;
        const int64_t row = row0 + lr;
        double contrib = 0.0;
        if (lr < bin_rows && row < nrows)
        {
            double yv = acc[lr];
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
            const double t = st_wave_sum(contrib);
            if (lane == 0)
                red[j * kWavesPer + w] = t;
        }
    }
    if (EPI)
    {
        __syncthreads();
        if (tid < (kStRows / 256))
        {
            const int lr0 = tid * 256;  // first row of the record inside the bin
            const int64_t blk = (row0 >> 8) + tid;
            if (lr0 < bin_rows && blk < nblocks256)
            {
                const double* q = red + tid * 4;  // record tid = waves 4 tid .. 4 tid + 3 of the (j, wave) grid
                epi.partials[blk] = (q[0] + q[1]) + (q[2] + q[3]);
            }
        }
    }
}

}  // namespace

// ---- host image ---------------------------------------------------------------------------------------------------------------------
bool build_staged(int64_t nrows, int64_t ncols, const int32_t* rowptr, const int32_t* colind, const double* val, HostStaged& out,
                  int resident)
{
    const int64_t nnz = int64_t(rowptr[nrows]) - rowptr[0];
    // bin height: 8192 rows — except for matrices with fewer such bins than the device holds workgroups (`resident`): their rows
    // are spread over `resident` bins (a multiple of 256 rows each: the alpha records of the fused epilogue are per 256 rows), so
    // that every CU has work.  (Spreading larger matrices over whole rounds of resident workgroups was measured too, r07f/r07g:
    // the smaller tiles fill the 64-entry chunks worse and the gain of the evened-out last round is lost again.)
    const int64_t min_bins = (nrows + kStRows - 1) >> kStRowBits;
    int64_t R = kStRows;
    if (min_bins < resident)
    {
        R = (nrows + resident - 1) / std::max(resident, 1);
        R = std::min<int64_t>(kStRows, std::max<int64_t>(256, (R + 255) / 256 * 256));
    }
    const int64_t ncb = (ncols + kStCols - 1) >> kStColBits, nbins = (nrows + R - 1) / R;
    if (nnz <= 0 || nnz + ncb >= (int64_t(1) << 32) - 2)
        return false;
    const int nt = int(std::max<int64_t>(1, std::min<int64_t>(ingest_threads(), nnz / (1 << 20))));
    // rows per thread: contiguous ranges; per-thread histograms over the column blocks; rows must be sorted by column
    std::vector<std::vector<int64_t>> hist(static_cast<size_t>(nt), std::vector<int64_t>(static_cast<size_t>(ncb), 0));
    std::vector<char> unsorted(static_cast<size_t>(nt), 0);
    std::vector<int64_t> rb(static_cast<size_t>(nt) + 1);
    for (int t = 0; t <= nt; t++)
        rb[size_t(t)] = nrows * t / nt;
    parallel_ranges(nt, nt, [&](int, int64_t tb, int64_t te) {
        for (int64_t t = tb; t < te; t++)
        {
            std::vector<int64_t>& h = hist[size_t(t)];
            char bad = 0;
            for (int64_t i = rb[size_t(t)]; i < rb[size_t(t) + 1]; i++)
                for (int64_t p = rowptr[i]; p < rowptr[i + 1]; p++)
                {
                    h[size_t(colind[p] >> kStColBits)]++;
                    bad |= char(p > rowptr[i] && colind[p] <= colind[p - 1]);
                }
            unsorted[size_t(t)] = bad;
        }
    });
    for (int t = 0; t < nt; t++)
        if (unsorted[size_t(t)])
            return false;  // the row sums would not follow the storage order
    // column-block segments (padded to an even number of slots) and each thread's first slot inside every segment
    std::vector<int64_t> cb_ptr(static_cast<size_t>(ncb) + 1, 0);
    for (int64_t c = 0; c < ncb; c++)
    {
        int64_t tot = 0;
        for (int t = 0; t < nt; t++)
        {
            const int64_t h = hist[size_t(t)][size_t(c)];
            hist[size_t(t)][size_t(c)] = cb_ptr[size_t(c)] + tot;  // becomes the thread's write cursor
            tot += h;
        }
        cb_ptr[size_t(c) + 1] = cb_ptr[size_t(c)] + ((tot + 1) & ~int64_t(1));
    }
    const int64_t slots = cb_ptr[size_t(ncb)];
    out.nrows = nrows;
    out.ncols = ncols;
    out.nnz = nnz;
    out.slots = slots;
    out.nbins = nbins;
    out.ncb = ncb;
    out.bin_rows = int(R);
    out.val.resize_uninitialized(size_t(slots));
    out.lcol.resize_uninitialized(size_t(slots));
    RawVec<int32_t> grow;  // row of every slot (-1: padding); temporary
    grow.resize_uninitialized(size_t(slots));
    // padding slots (at most one per segment, the last one) first, the scatter overwrites nothing else
    for (int64_t c = 0; c < ncb; c++)
        if (cb_ptr[size_t(c) + 1] > cb_ptr[size_t(c)])
        {
            const int64_t last = cb_ptr[size_t(c) + 1] - 1;
            out.val[size_t(last)] = 0.0;
            out.lcol[size_t(last)] = 0;
            grow[size_t(last)] = -1;
        }
    parallel_ranges(nt, nt, [&](int, int64_t tb, int64_t te) {
        for (int64_t t = tb; t < te; t++)
        {
            std::vector<int64_t>& cur = hist[size_t(t)];
            for (int64_t i = rb[size_t(t)]; i < rb[size_t(t) + 1]; i++)
                for (int64_t p = rowptr[i]; p < rowptr[i + 1]; p++)
                {
                    const int32_t col = colind[p];
                    const int64_t s = cur[size_t(col >> kStColBits)]++;
                    out.val[size_t(s)] = val[p];
                    out.lcol[size_t(s)] = uint16_t(col & (kStCols - 1));
                    grow[size_t(s)] = int32_t(i);
                }
        }
    });
    // phase-1 work list
    out.pieces.clear();
    for (int64_t c = 0; c < ncb; c++)
        for (int64_t b = cb_ptr[size_t(c)]; b < cb_ptr[size_t(c) + 1]; b += kStPiece)
            out.pieces.push_back(StPiece{b, std::min(b + kStPiece, cb_ptr[size_t(c) + 1]), int32_t(c), 0});
    // first slot of every (column block, bin) tile: rows ascend inside a segment.  The table is ncb x (nbins + 1) — quadratic in
    // n: 6 MB at n = 1e7, 600 MB at 1e8 — so the format declines matrices beyond a fixed budget (they keep the tiles / CSR kernels)
    if (double(ncb) * double(nbins + 1) > 64.0 * 1024.0 * 1024.0)
        return false;
    std::vector<uint32_t> tile(static_cast<size_t>(ncb) * size_t(nbins + 1));
    parallel_ranges(ncb, nt, [&](int, int64_t cb0, int64_t cb1) {
        for (int64_t c = cb0; c < cb1; c++)
        {
            uint32_t* T = tile.data() + size_t(c) * size_t(nbins + 1);
            int64_t s = cb_ptr[size_t(c)];
            const int64_t e = cb_ptr[size_t(c) + 1];
            for (int64_t bin = 0; bin <= nbins; bin++)
            {
                const int64_t first_row = bin * R;
                while (s < e && grow[size_t(s)] >= 0 && grow[size_t(s)] < first_row)
                    s++;
                T[bin] = uint32_t(s);  // (a padding slot, row -1, ends the segment: s stops there for every later bin)
            }
        }
    });
    out.nnz = nnz;
    // phase 2, bin by bin: chunks of <= 64 phase-1-contiguous entries, kStWaves chunks per batch, ranks inside the batch
    out.rowrank.resize_uninitialized(size_t(slots));  // row inside the bin | rank, at the entry's phase-1 position (padding: never read)
    for (int64_t c = 0; c < ncb; c++)
        if (cb_ptr[size_t(c) + 1] > cb_ptr[size_t(c)])
            out.rowrank[size_t(cb_ptr[size_t(c) + 1] - 1)] = 0;
    struct BinOut
    {
        std::vector<uint64_t> desc;     // kStBatchChunks per batch
        int64_t chunks = 0;
    };
    std::vector<BinOut> bins(static_cast<size_t>(nbins));
    parallel_ranges(nbins, nt, [&](int, int64_t bin0, int64_t bin1) {
        std::vector<uint16_t> seen(static_cast<size_t>(kStRows), 0);   // rank counter per row of the bin ...
        std::vector<uint32_t> stamp(static_cast<size_t>(kStRows), 0);  // ... valid for the batch with this number
        uint32_t batch_no = 0;
        for (int64_t bin = bin0; bin < bin1; bin++)
        {
            BinOut& B = bins[size_t(bin)];
            const int64_t first_row = bin * R;
            int nch = 0, cnt = 0, maxrank = 0;  // of the open batch: chunks started, entries in the open chunk, largest rank
            int64_t last_slot = -2;
            size_t base = 0;  // the open batch's first descriptor
            auto open = [&]() {
                base = B.desc.size();
                B.desc.resize(base + kStBatchChunks, 0ull);
                nch = 0;
                cnt = 0;
                maxrank = 0;
                last_slot = -2;
                batch_no++;
            };
            auto close = [&]() {  // the rounds of the batch into every descriptor
                for (int g = 0; g < kStBatchChunks; g++)
                    B.desc[base + size_t(g)] |= uint64_t(maxrank + 1) << 40;
            };
            bool is_open = false;
            for (int64_t c = 0; c < ncb; c++)
            {
                const uint32_t* T = tile.data() + size_t(c) * size_t(nbins + 1);
                for (int64_t s = T[bin]; s < int64_t(T[bin + 1]); s++)
                {
                    const int r = int(grow[size_t(s)] - first_row);
                    if (!is_open)
                    {
                        open();
                        is_open = true;
                    }
                    int rank = (stamp[size_t(r)] == batch_no) ? int(seen[size_t(r)]) : 0;
                    bool new_chunk = (s != last_slot + 1) || cnt == kStChunk;
                    if (rank > kStMaxRank || (new_chunk && nch == kStBatchChunks))
                    {
                        close();
                        open();
                        rank = 0;
                        new_chunk = true;
                    }
                    if (new_chunk)
                    {
                        B.desc[base + size_t(nch)] = uint64_t(uint32_t(s));
                        nch++;
                        cnt = 0;
                        B.chunks++;
                    }
                    out.rowrank[size_t(s)] = uint16_t(r | (rank << kStRowBits));  // a slot belongs to one bin: no two threads meet
                    cnt++;
                    B.desc[base + size_t(nch - 1)] = (B.desc[base + size_t(nch - 1)] & 0xFFFFFFFFull) | (uint64_t(cnt) << 32);
                    stamp[size_t(r)] = batch_no;
                    seen[size_t(r)] = uint16_t(rank + 1);
                    maxrank = std::max(maxrank, rank);
                    last_slot = s;
                }
            }
            if (is_open)
                close();
        }
    });
    // concatenate
    out.bin_batch.assign(static_cast<size_t>(nbins) + 1, 0);
    int64_t nb = 0;
    out.nchunks = 0;
    for (int64_t bin = 0; bin < nbins; bin++)
    {
        nb += int64_t(bins[size_t(bin)].desc.size()) / kStBatchChunks;
        out.bin_batch[size_t(bin) + 1] = int32_t(nb);
        out.nchunks += bins[size_t(bin)].chunks;
    }
    out.nbatches = nb;
    // A row with more than 8 entries inside one batch closes the batch (3-bit ranks): a few dense or heavy rows among scattered
    // ones produce one nearly empty batch per 8 of their entries, all of them serial barrier rounds of one workgroup — the product
    // stays correct but takes orders of magnitude longer.  `well_filled` is false when fewer than a quarter of the
    // chunk slots of the batches are in use: the automatic format choice (csr.hip upload_rows) then declines the image.
    out.well_filled = !(double(nb) * double(kStBatchChunks) > 4.0 * double(out.nchunks) + 4.0 * double(kStBatchChunks) * double(nbins));
    out.desc.resize_uninitialized(size_t(nb) * kStBatchChunks);
    parallel_ranges(nbins, nt, [&](int, int64_t bin0, int64_t bin1) {
        for (int64_t bin = bin0; bin < bin1; bin++)
        {
            const BinOut& B = bins[size_t(bin)];
            const size_t at = size_t(out.bin_batch[size_t(bin)]);
            if (!B.desc.empty())
            {
                std::memcpy(out.desc.data() + at * kStBatchChunks, B.desc.data(), B.desc.size() * sizeof(uint64_t));
            }
        }
    });
    return true;
}

void staged_spmv_host(const HostStaged& S, const double* x, double* y)
{
    // phase 1
    std::vector<double> prod(static_cast<size_t>(S.slots));
    for (const StPiece& pc : S.pieces)
    {
        const int64_t c0 = int64_t(pc.colblock) << kStColBits;
        for (int64_t s = pc.begin; s < pc.end; s++)
        {
            const int64_t col = c0 + S.lcol[size_t(s)];
            prod[size_t(s)] = S.val[size_t(s)] * (col < S.ncols ? x[col] : 0.0);
        }
    }
    // phase 2: batch after batch, rank after rank
    std::vector<double> acc(static_cast<size_t>(kStRows));
    for (int64_t bin = 0; bin < S.nbins; bin++)
    {
        std::fill(acc.begin(), acc.end(), 0.0);
        for (int64_t b = S.bin_batch[size_t(bin)]; b < S.bin_batch[size_t(bin) + 1]; b++)
        {
            const int rounds = int((S.desc[size_t(b) * kStBatchChunks] >> 40) & 0xFFu);
            for (int r = 0; r < rounds; r++)
                for (int g = 0; g < kStBatchChunks; g++)
                {
                    const uint64_t d = S.desc[size_t(b) * kStBatchChunks + size_t(g)];
                    const int cnt = int((d >> 32) & 0xFFu);
                    for (int t = 0; t < cnt; t++)
                    {
                        const uint16_t rr = S.rowrank[size_t(uint32_t(d)) + size_t(t)];
                        if ((rr >> kStRowBits) == r)
                            acc[size_t(rr & (kStRows - 1))] += prod[size_t(uint32_t(d)) + size_t(t)];
                    }
                }
        }
        const int64_t row0 = bin * S.bin_rows;
        for (int64_t r = 0; r < S.bin_rows && row0 + r < S.nrows; r++)
            y[row0 + r] = acc[size_t(r)];
    }
}

// ---- device image ---------------------------------------------------------------------------------------------------------------------
void DevStaged::swap(DevStaged& o)
{
    val.swap(o.val);
    prod.swap(o.prod);
    lcol.swap(o.lcol);
    rowrank.swap(o.rowrank);
    pieces.swap(o.pieces);
    bin_batch.swap(o.bin_batch);
    desc.swap(o.desc);
    std::swap(nnz, o.nnz);
    std::swap(slots, o.slots);
    std::swap(nbins, o.nbins);
    std::swap(ncb, o.ncb);
    std::swap(npieces, o.npieces);
    std::swap(nbatches, o.nbatches);
    std::swap(nchunks, o.nchunks);
    std::swap(bin_rows, o.bin_rows);
}

void upload_staged(const HostStaged& H, hipStream_t stream, DevStaged& D)
{
    auto up = [&](auto& dst, const auto* src, size_t count) {
        dst.alloc(count);
        if (count)
            MISPEC_HIP(hipMemcpyAsync(dst.p, src, count * sizeof(*src), hipMemcpyHostToDevice, stream));
    };
    up(D.val, H.val.data(), H.val.size());
    up(D.lcol, H.lcol.data(), H.lcol.size());
    up(D.rowrank, H.rowrank.data(), H.rowrank.size());
    up(D.pieces, H.pieces.data(), H.pieces.size());
    up(D.bin_batch, H.bin_batch.data(), H.bin_batch.size());
    up(D.desc, H.desc.data(), H.desc.size());
    D.prod.alloc(size_t(H.slots));
    D.nnz = H.nnz;
    D.slots = H.slots;
    D.nbins = H.nbins;
    D.ncb = H.ncb;
    D.npieces = int64_t(H.pieces.size());
    D.nbatches = H.nbatches;
    D.bin_rows = H.bin_rows;
    D.nchunks = H.nchunks;
    MISPEC_HIP(hipStreamSynchronize(stream));
}

void launch_spmv_staged(const DevStaged& S, hipStream_t stream, const double* x, double* y, int64_t nrows, int64_t ncols, int nblocks256,
                        const SpmvEpilogue* epi, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    const SpmvEpilogue e = epi ? *epi : SpmvEpilogue();
    const int* status = epi ? epi->status : nullptr;
    if (ev_start)
        MISPEC_HIP(hipEventRecord(ev_start, stream));
    hipLaunchKernelGGL(k_staged_products, dim3(unsigned(S.npieces)), dim3(kStThreads), 0, stream, S.pieces.p, S.val.p, S.lcol.p, x, S.prod.p,
                       ncols, status);
    if (epi)
        hipLaunchKernelGGL((k_staged_rows<true>), dim3(unsigned(S.nbins)), dim3(kStRowThreads), 0, stream, S.bin_batch.p, S.desc.p, S.rowrank.p, S.prod.p, y, nrows,
                           nblocks256, S.bin_rows, e);
    else
        hipLaunchKernelGGL((k_staged_rows<false>), dim3(unsigned(S.nbins)), dim3(kStRowThreads), 0, stream, S.bin_batch.p, S.desc.p, S.rowrank.p, S.prod.p, y, nrows,
                           nblocks256, S.bin_rows, e);
    if (ev_stop)
        MISPEC_HIP(hipEventRecord(ev_stop, stream));
    MISPEC_HIP(hipGetLastError());
}

}  // namespace mispec

// Host-only test hook (no device): builds the staged image of a CSR matrix and multiplies with it on the host in the two kernels'
// order.  *built = 0 when the format does not apply.  stats: [bins, phase-1 slots, batches, chunks, largest rank + 1 of a batch]
extern "C" int mispec_staged_spmv_host(int64_t nrows, int64_t ncols, const int32_t* rowptr, const int32_t* colind, const double* val,
                                       const double* x, double* y, int* built, int64_t* stats)
{
    return mispec::guarded([&] {
        MISPEC_REQUIRE(rowptr && x && y && built, "mispec_staged_spmv_host: NULL argument");
        mispec::HostStaged S;
        *built = mispec::build_staged(nrows, ncols, rowptr, colind, val, S) ? 1 : 0;
        if (!*built)
            return;
        if (!S.well_filled)
            *built = 2;  // the image is correct, but the automatic format choice declines it (nearly empty batches)
        mispec::staged_spmv_host(S, x, y);
        if (stats)
        {
            stats[0] = S.nbins;
            stats[1] = S.slots;
            stats[2] = S.nbatches;
            stats[3] = S.nchunks;
            int64_t rounds = 0;
            for (int64_t b = 0; b < S.nbatches; b++)
                rounds = std::max<int64_t>(rounds, int64_t((S.desc[size_t(b) * mispec::kStBatchChunks] >> 40) & 0xFFu));
            stats[4] = rounds;
        }
    });
}
