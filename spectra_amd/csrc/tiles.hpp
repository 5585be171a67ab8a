// Column-blocked "tile" storage for matrices whose x gathers are scattered over the whole vector and that no reordering can
// localise (expander-like patterns: SURVEY.md 8d's M-rand).  See tiles.hip.
#pragma once
#include <vector>

#include "common.hpp"

namespace mispec {

constexpr int kTileRowBits = 12;                 // rows per segment: 4096 (32 KiB of fp64 accumulators in LDS)
constexpr int kTileRows = 1 << kTileRowBits;
constexpr int kTileColBits = 17;                 // columns per block: 131072 (1 MiB of x: a quarter of an XCD's L2)
constexpr int kTileCols = 1 << kTileColBits;
constexpr int kTileRunBits = 3;                  // entries of one row inside one tile: at most 7
constexpr int kTileMaxRun = (1 << kTileRunBits) - 1;
constexpr int kTileChunk = 1024;                 // entries handled between two barriers (4 per thread, 256 threads)
constexpr uint32_t kTileSkip = 0xFFFFFFFFu;      // padding entry

// idx = row_local << 20 | col_local << 3 | run      run = 0: continuation of the run started by an earlier entry;
//                                                   run = k >= 1: first of k consecutive entries of this row in this tile
struct TileChunk
{
    int32_t offset;   // first entry, relative to the segment's first entry
    uint16_t count;   // <= kTileChunk
    uint16_t colblock;
};

struct HostTiles
{
    std::vector<int64_t> seg_entry;   // nseg + 1: first entry of every segment
    std::vector<int32_t> seg_chunk;   // nseg + 1: first chunk of every segment
    std::vector<TileChunk> chunks;
    std::vector<double> val;          // padded entries carry 0.0
    std::vector<uint32_t> idx;
    int64_t padding = 0;              // padding entries inserted
    int64_t ncb = 0;                  // column blocks
};

// Build the tiles of rows [0, nrows) of a CSR matrix (rows sorted by column, no duplicates).  Returns false (nothing
// built) when the format does not apply: a row with more than kTileMaxRun entries inside one column block, unsorted rows,
// more than 65535 column blocks.
bool build_tiles(int64_t nrows, int64_t ncols, const int32_t* rowptr, const int32_t* colind, const double* val, HostTiles& out);

// Device image of the tiles + launcher (y = A x, optional fused Lanczos epilogue on the same 256-row records as the CSR kernels)
struct DevTiles
{
    DevBuf<int64_t> seg_entry;
    DevBuf<int32_t> seg_chunk;
    DevBuf<TileChunk> chunks;
    DevBuf<double> val;
    DevBuf<uint32_t> idx;
    DevBuf<unsigned int> sync_counters;  // the loose per-XCD barrier of the persistent variant (MISPEC_TILES_SYNC)
    int64_t nseg = 0, entries = 0, nchunks = 0, padding = 0, ncb = 0;
    int sync_period = 0;                 // 0: free-running workgroups; k: persistent workgroups meeting every k column blocks
    bool present() const { return nseg > 0; }
    void swap(DevTiles& o)
    {
        seg_entry.swap(o.seg_entry);
        seg_chunk.swap(o.seg_chunk);
        chunks.swap(o.chunks);
        val.swap(o.val);
        idx.swap(o.idx);
        sync_counters.swap(o.sync_counters);
        std::swap(ncb, o.ncb);
        std::swap(sync_period, o.sync_period);
        std::swap(nseg, o.nseg);
        std::swap(entries, o.entries);
        std::swap(nchunks, o.nchunks);
        std::swap(padding, o.padding);
    }
};
void upload_tiles(const HostTiles& H, hipStream_t stream, DevTiles& D);
void calibrate_tiles(DevTiles& T, hipStream_t stream, int64_t nrows, int64_t ncols, int nblocks256);
struct SpmvEpilogue;
void launch_spmv_tiles(const DevTiles& T, hipStream_t stream, const double* x, double* y, int64_t nrows, int nblocks256,
                       const SpmvEpilogue* epi, hipEvent_t ev_start, hipEvent_t ev_stop);

// y = A x from the host image of the tiles, one "thread": the summation order the kernel uses (= CSR storage order).
void tiles_spmv_host(const HostTiles& T, int64_t nrows, const double* x, double* y);

}  // namespace mispec
