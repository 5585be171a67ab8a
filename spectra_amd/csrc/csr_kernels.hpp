// Device helpers shared by the SpMV kernels of csr.hip (CSR-stream), csr_win.hip (int32 CSR with x windows) and csr_dia.hip
// (diagonal storage): vector types, the LDS chunk sizes, the offset-code descriptor and the fixed-order block sum.  Internal.
#pragma once
#include "csr.hpp"
#include "krylov.hpp"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

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

}  // namespace

namespace mispec {

// What launch_spmv_raw (csr.hip) hands to the per-format launchers: the grid over the row-blocks [first_block, first_block +
// nblocks) of the shard (XCD-aware block map inside the kernels), the epilogue with first_block filled in, the event pair of a
// timed launch (the dispatch's own completion signal), the operands.
struct SpmvLaunch
{
    dim3 grid, block;
    int64_t nloc;
    int nblocks;
    const SpmvEpilogue* epi;  // nullptr: plain product
    SpmvEpilogue e;
    hipEvent_t ev_start, ev_stop;
    const double* x_dev;
    double* y_dev;
};
// csr_dia.hip — diagonal storage (format 2): build at ingest, launch
void build_dia(mispec_csr& A, const std::vector<int32_t>& dict);
void launch_spmv_dia(const mispec_csr& A, const SpmvLaunch& L);
// csr_win.hip — int32 CSR with the x entries of a row-block staged through LDS windows (format 0 with a window table)
void build_windows(mispec_csr& A);
void launch_spmv_csr_win(const mispec_csr& A, const SpmvLaunch& L);

}  // namespace mispec
