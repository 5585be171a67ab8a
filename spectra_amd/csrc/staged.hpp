// "Staged" storage for scattered patterns (fifth SpMV format): the product in two streaming phases with x AND y in LDS, for
// matrices whose x gathers no cache level can serve (SURVEY.md 8d's M-rand: the one-phase kernels are bound by the rate at which
// a CU moves 8-byte gathers through its L1, about 110 G/s for the device — DESIGN.md 3.1).  See staged.hip.
//
//   phase 1  entries in COLUMN-BLOCK-major order (blocks of 8192 columns): a workgroup stages its block of x in LDS (64 KiB),
//            streams value + 16-bit local column of its entries and writes the products, in the same order — a pure stream.
//   phase 2  a workgroup owns a BIN of 8192 rows (64 KiB of accumulators in LDS) and adds the bin's products in batches: two
//            CHUNKS per wavefront, a chunk being <= 64 entries that are contiguous in phase-1 order (a piece of the bin's share
//            of one column block), chunks and batches following the column blocks.  Inside a batch the entries of one row are
//            ranked in column order and applied rank by rank with a barrier in between: every row is summed in ascending column
//            order, one accumulator, i.e. exactly the CSR row sum of the other kernels and of the oracle (bit-identical results).
// Bytes per entry: phase 1 reads 10 and writes 8, phase 2 reads 8 + 2 (row | rank, stored at the entry's phase-1 position like
// the product) + the descriptors (8 per chunk), about 28.2 in all, against 12 + a gather.
#pragma once
#include <vector>

#include "common.hpp"

namespace mispec {

constexpr int kStRowBits = 13;                  // rows per bin: 8192 (64 KiB of fp64 accumulators); fewer (a multiple of 256) for
constexpr int kStRows = 1 << kStRowBits;        // matrices that would otherwise have fewer bins than the device holds workgroups
constexpr int kStColBits = 13;                  // columns per block: 8192 (64 KiB of x)
constexpr int kStCols = 1 << kStColBits;
constexpr int kStThreads = 1024;                // phase 1: two workgroups per CU
constexpr int kStRowThreads = 1024;             // phase 2 (4096-row bins with 512 threads: the same time, profiles/r07q)
constexpr int kStWaves = kStRowThreads / 64;
// chunks a wavefront adds per batch.  Phase 2 is bound by its chain of rank rounds (one LDS read-add-write and a barrier each,
// about 3 per batch whatever its size up to here): two chunks per wavefront halve the rounds per entry — 0.468 -> 0.402 ms on
// M-rand; four gain nothing more (profiles/r07r)
constexpr int kStPerWave = 2;
constexpr int kStBatchChunks = kStWaves * kStPerWave;  // chunks per batch: chunk j of a batch goes to wavefront j % kStWaves
constexpr int kStChunk = 64;                    // entries per chunk: one per lane
constexpr int kStRankBits = 16 - kStRowBits;    // 3: an entry's rank among the entries of its row inside its batch
constexpr int kStMaxRank = (1 << kStRankBits) - 1;
constexpr int64_t kStPiece = int64_t(1) << 16;  // phase 1: entries per workgroup and staging of an x block (2^18: 5 % slower, r07g)

struct StPiece
{
    int64_t begin, end;  // phase-1 positions (multiples of 2)
    int32_t colblock;
    int32_t pad;
};

struct HostStaged
{
    int64_t nrows = 0, ncols = 0, nnz = 0, slots = 0;  // slots: phase-1 positions (nnz + padding of the column blocks to even counts)
    int64_t nbins = 0, ncb = 0;
    int bin_rows = kStRows;        // rows per bin (multiple of 256, <= kStRows)
    RawVec<double> val;            // [slots] phase-1 order; padding slots carry 0.0
    RawVec<uint16_t> lcol;         // [slots] column inside the block
    std::vector<StPiece> pieces;   // phase-1 work list
    // phase 2: batch b of a bin = chunks [b * kStBatchChunks, (b + 1) * kStBatchChunks)
    std::vector<int32_t> bin_batch;    // nbins + 1
    RawVec<uint64_t> desc;         // [batches * kStBatchChunks] phase-1 position of the chunk | entries << 32 | rounds of its batch << 40
    RawVec<uint16_t> rowrank;      // [slots] phase-1 order like val: row inside the bin | rank << kStRowBits
    int64_t nbatches = 0, nchunks = 0;  // nchunks: chunks that hold entries
    bool well_filled = true;            // false: heavy rows split the batches into nearly empty ones (the automatic choice declines)
};

// Build the image of rows [0, nrows) of a CSR matrix (any pattern; rows need not be sorted: the row sums follow the storage
// order either way).  Returns false (nothing built) when the format does not apply: more than 2^32 - 2 stored entries.
// resident: workgroups of phase 2 the device holds at a time (2 per CU): a matrix with fewer 8192-row bins than that gets lower
// bins, so that every CU has one.
bool build_staged(int64_t nrows, int64_t ncols, const int32_t* rowptr, const int32_t* colind, const double* val, HostStaged& out,
                  int resident = 512);

struct DevStaged
{
    DevBuf<double> val, prod;
    DevBuf<uint16_t> lcol, rowrank;
    DevBuf<StPiece> pieces;
    DevBuf<int32_t> bin_batch;
    DevBuf<uint64_t> desc;
    int64_t nnz = 0, slots = 0, nbins = 0, ncb = 0, npieces = 0, nbatches = 0, nchunks = 0;
    int bin_rows = kStRows;
    bool present() const { return nbins > 0; }
    void swap(DevStaged& o);
    // bytes one product has to move (both phases, incl. the product array's round trip and the tables)
    double stored_bytes(int64_t n_rows, int64_t n_cols) const
    {
        return 18.0 * double(slots) + 10.0 * double(nnz) + 8.0 * double(kStBatchChunks) * double(nbatches) + 16.0 * double(npieces) +
               8.0 * double(n_cols) + 8.0 * double(n_rows);
    }
};
void upload_staged(const HostStaged& H, hipStream_t stream, DevStaged& D);
struct SpmvEpilogue;
// ev_start / ev_stop bracket BOTH launches (start of phase 1, completion of phase 2)
void launch_spmv_staged(const DevStaged& S, hipStream_t stream, const double* x, double* y, int64_t nrows, int64_t ncols, int nblocks256,
                        const SpmvEpilogue* epi, hipEvent_t ev_start, hipEvent_t ev_stop);

// y = A x from the host image, one "thread", in the order the two kernels use.
void staged_spmv_host(const HostStaged& S, const double* x, double* y);

}  // namespace mispec
