// Dense operators in HBM: the matrix behind DenseSymMatProd / DenseGenMatProd, and the row-major GEMV that the
// dense shift-solve and Cholesky paths share.
#pragma once
#include "common.hpp"

struct mispec_dense
{
    mispec_ctx* ctx = nullptr;
    int64_t rows = 0, cols = 0;
    int64_t ld = 0;            // row stride (cols rounded up to even, padding is zero)
    mispec::DevBuf<double> a;  // row-major rows x ld
    mutable mispec::DevBuf<double> stage_x, stage_y;  // host-pointer paths
    double algorithmic_bytes() const { return 8.0 * double(rows) * double(cols) + 8.0 * double(cols) + 8.0 * double(rows); }
};

namespace mispec {
// y[rows] = M x, M row-major with row stride ld (device pointers, enqueued on the context's stream).  One wavefront per
// row, fixed summation order: deterministic, independent of the launch geometry.
// storage_order_if_small: matrices of at most 128 x 128 entries are multiplied with one lane per row, each row summed in storage
// order like a CPU row-dot and the sparse kernels (the dense product operators ask for it; see k_row_gemv_serial)
void launch_row_gemv(const mispec_ctx& ctx, const double* M, int64_t ld, int64_t rows, int64_t cols, const double* x, double* y,
                     bool storage_order_if_small = false);
}  // namespace mispec
