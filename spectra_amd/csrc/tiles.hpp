// Column-blocked "tile" storage for matrices whose x gathers are scattered over the whole vector and that no reordering can
// localise (expander-like patterns: SURVEY.md 8d's M-rand).  See tiles.hip.
#pragma once
#include <vector>

#include "common.hpp"

namespace mispec {

// Geometry (compile-time; the defaults are the measured best of profiles/rounds_1_2/r02_mrand_tile_geometry_sweep.jsonl, M-rand n = 1e7):
//   segments of 8192 rows (64 KiB of accumulators: two workgroups per CU), column blocks of 65536 columns (512 KiB of x),
//   512 threads per workgroup, chunks of 1024 entries.  4096 x 131072 x 256 threads: 1.87 ms; 8192 x 65536 x 256: 1.50;
//   8192 x 65536 x 512: 1.29-1.32; 8192 x 65536 x 1024: 1.40; 8192 x 131072 (2-bit runs): 1.65-1.70; 16384 x 32768: 1.60-1.99;
//   2048 x 131072: 2.27 (int32 CSR kernel on the same matrix: 2.67).
#ifndef MISPEC_TILE_ROW_BITS
#define MISPEC_TILE_ROW_BITS 13
#endif
#ifndef MISPEC_TILE_COL_BITS
#define MISPEC_TILE_COL_BITS 16
#endif
#ifndef MISPEC_TILE_THREADS
#define MISPEC_TILE_THREADS 512
#endif
constexpr int kTileRowBits = MISPEC_TILE_ROW_BITS;  // rows per segment: 8192 (64 KiB of fp64 accumulators in LDS)
constexpr int kTileRows = 1 << kTileRowBits;
constexpr int kTileColBits = MISPEC_TILE_COL_BITS;  // columns per block: 65536 (512 KiB of x)
constexpr int kTileThreads = MISPEC_TILE_THREADS;   // threads of the workgroup that owns a segment
constexpr int kTileCols = 1 << kTileColBits;
constexpr int kTileRunBits = 32 - MISPEC_TILE_ROW_BITS - MISPEC_TILE_COL_BITS;  // run length of a head entry: the bits that are left
constexpr int kTileMaxRun = (1 << kTileRunBits) - 1;
#ifndef MISPEC_TILE_CHUNK
#define MISPEC_TILE_CHUNK 1024
#endif
constexpr int kTileChunk = MISPEC_TILE_CHUNK;       // entries handled between two barriers (kTileChunk / kTileThreads per thread)
static_assert(kTileRunBits >= 2 && kTileRunBits <= 8, "tile index packing: row + column bits must leave 2..8 bits for the run length");
constexpr uint32_t kTileSkip = 0xFFFFFFFFu;      // padding entry
constexpr int kTileSlack = 2 * kTileChunk;        // padding entries behind the last segment: the kernel prefetches two chunks ahead

// idx = row_local << (col bits + 3) | col_local << 3 | run      run = 0: continuation of the run started by an earlier entry;
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
    RawVec<TileChunk> chunks;
    RawVec<double> val;               // padded entries carry 0.0
    RawVec<uint32_t> idx;
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
    int64_t nseg = 0, entries = 0, nchunks = 0, padding = 0, ncb = 0;
    bool present() const { return nseg > 0; }
    void swap(DevTiles& o)
    {
        seg_entry.swap(o.seg_entry);
        seg_chunk.swap(o.seg_chunk);
        chunks.swap(o.chunks);
        val.swap(o.val);
        idx.swap(o.idx);
        std::swap(ncb, o.ncb);
        std::swap(nseg, o.nseg);
        std::swap(entries, o.entries);
        std::swap(nchunks, o.nchunks);
        std::swap(padding, o.padding);
    }
};
void upload_tiles(const HostTiles& H, hipStream_t stream, DevTiles& D);
struct SpmvEpilogue;
void launch_spmv_tiles(const DevTiles& T, hipStream_t stream, const double* x, double* y, int64_t nrows, int nblocks256,
                       const SpmvEpilogue* epi, hipEvent_t ev_start, hipEvent_t ev_stop);

// y = A x from the host image of the tiles, one "thread": the summation order the kernel uses (= CSR storage order).
void tiles_spmv_host(const HostTiles& T, int64_t nrows, const double* x, double* y);

}  // namespace mispec
