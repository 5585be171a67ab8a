// Dense matrices in HBM and y = M x for them (DenseSymMatProd.h:81-86, DenseGenMatProd.h:78-83 of the reference are
// `m_mat.selfadjointView<Uplo>() * x` and `m_mat * x` through Eigen).
//
// Layout: row-major, row stride rounded up to even so that every row starts on a 16-byte boundary; a symmetric input
// given by one triangle is mirrored at upload (the other triangle of the input is ignored, as selfadjointView does).
// Kernel: one wavefront per row.  A lane streams 16-byte pieces of the row (four in flight), x comes from the L2
// (it is read by every row), a shuffle tree finishes the dot product.  Bound: HBM, 8 bytes per matrix entry.
#include "dense.hpp"

#include <cstring>
#include <memory>
#include <vector>

using namespace mispec;

namespace {

constexpr int kGemvThreads = 256;  // four rows per workgroup
constexpr int64_t kSerialGemvEntries = 128 * 128;  // up to this many entries: storage-order row sums (k_row_gemv_serial)

// VEC: row stride even and x 16-byte aligned -> double2 loads; else a scalar walk with the same reduction tree shape.
template <bool VEC>
__global__ __launch_bounds__(kGemvThreads) void k_row_gemv(int64_t rows, int64_t cols, int64_t ld, const double* __restrict__ M,
                                                            const double* __restrict__ x, double* __restrict__ y)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = int64_t(blockIdx.x) * (kGemvThreads / 64) + (threadIdx.x >> 6);
    if (r >= rows)
        return;
    const double* a = M + r * ld;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    if (VEC)
    {
        const double2* a2 = reinterpret_cast<const double2*>(a);
        const double2* x2 = reinterpret_cast<const double2*>(x);
        const int64_t pairs = cols >> 1;
        int64_t p = lane;
        for (; p + 192 < pairs; p += 256)
        {
            const double2 m0 = a2[p], m1 = a2[p + 64], m2 = a2[p + 128], m3 = a2[p + 192];
            const double2 v0 = x2[p], v1 = x2[p + 64], v2 = x2[p + 128], v3 = x2[p + 192];
            acc0 += m0.x * v0.x + m0.y * v0.y;
            acc1 += m1.x * v1.x + m1.y * v1.y;
            acc2 += m2.x * v2.x + m2.y * v2.y;
            acc3 += m3.x * v3.x + m3.y * v3.y;
        }
        for (; p < pairs; p += 64)
        {
            const double2 m0 = a2[p];
            const double2 v0 = x2[p];
            acc0 += m0.x * v0.x + m0.y * v0.y;
        }
        if ((cols & 1) && lane == 0)
            acc1 += a[cols - 1] * x[cols - 1];
    }
    else
    {
        int64_t c = lane;
        for (; c + 192 < cols; c += 256)
        {
            acc0 += a[c] * x[c];
            acc1 += a[c + 64] * x[c + 64];
            acc2 += a[c + 128] * x[c + 128];
            acc3 += a[c + 192] * x[c + 192];
        }
        for (; c < cols; c += 64)
            acc0 += a[c] * x[c];
    }
    double acc = (acc0 + acc1) + (acc2 + acc3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        acc += __shfl_down(acc, off, 64);
    if (lane == 0)
        y[r] = acc;
}

// Small matrices: one lane per row, the row summed in STORAGE ORDER with the product rounded before it is added — the order
// of a plain CPU row-dot and of the sparse kernels (csr.hip), whose results this kernel therefore reproduces bit for bit on
// the same entries (an entry that is 0 adds +-0).  The wavefront tree above rounds differently, which is harmless except
// where a test of the reference sits exactly on a rounding boundary: test/Example1.cpp (20, 5, 12) exhausts its Krylov space
// and the surviving noise (1e-15) lands on either side of the breakdown clamp (Lanczos.h:163-168) depending on that order.
// A launch this small is latency-bound either way.
__global__ __launch_bounds__(64) void k_row_gemv_serial(int64_t rows, int64_t cols, int64_t ld, const double* __restrict__ M,
                                                         const double* __restrict__ x, double* __restrict__ y)
{
#pragma clang fp contract(off)
    const int64_t r = int64_t(blockIdx.x) * 64 + threadIdx.x;
    if (r >= rows)
        return;
    const double* a = M + r * ld;
    double acc = 0.0;
    for (int64_t c = 0; c < cols; c++)
        acc += a[c] * x[c];
    y[r] = acc;
}

void ensure_stage(const mispec_dense& D)
{
    if (D.stage_x.n < size_t(D.cols) + 2)
        D.stage_x.alloc(size_t(D.cols) + 2);
    if (D.stage_y.n < size_t(D.rows) + 2)
        D.stage_y.alloc(size_t(D.rows) + 2);
}

}  // namespace

namespace mispec {

void launch_row_gemv(const mispec_ctx& ctx, const double* M, int64_t ld, int64_t rows, int64_t cols, const double* x, double* y,
                     bool storage_order_if_small)
{
    if (rows <= 0)
        return;
    if (storage_order_if_small && rows * cols <= kSerialGemvEntries)
    {
        hipLaunchKernelGGL(k_row_gemv_serial, dim3(unsigned((rows + 63) / 64)), dim3(64), 0, ctx.stream, rows, cols, ld, M, x, y);
        MISPEC_HIP(hipGetLastError());
        return;
    }
    const dim3 grid(unsigned((rows + kGemvThreads / 64 - 1) / (kGemvThreads / 64))), block(kGemvThreads);
    const bool vec = (ld % 2 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(M) % 16 == 0);
    if (vec)
        hipLaunchKernelGGL(k_row_gemv<true>, grid, block, 0, ctx.stream, rows, cols, ld, M, x, y);
    else
        hipLaunchKernelGGL(k_row_gemv<false>, grid, block, 0, ctx.stream, rows, cols, ld, M, x, y);
    MISPEC_HIP(hipGetLastError());
}

}  // namespace mispec

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" int mispec_dense_upload(mispec_ctx* ctx, int64_t rows, int64_t cols, const double* data_host, int64_t ld_host,
                                   int row_major, char uplo, mispec_dense** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && out && rows >= 0 && cols >= 0, "mispec_dense_upload: bad argument");
        MISPEC_REQUIRE(data_host || rows * cols == 0, "mispec_dense_upload: NULL matrix");
        MISPEC_REQUIRE(ld_host >= (row_major ? cols : rows), "mispec_dense_upload: leading dimension too small");
        const bool sym = (uplo == 'L' || uplo == 'l' || uplo == 'U' || uplo == 'u');
        MISPEC_REQUIRE(sym || uplo == 0 || uplo == 'G' || uplo == 'g', "mispec_dense_upload: uplo must be 'L', 'U' or 0 (general)");
        MISPEC_REQUIRE(!sym || rows == cols, "mispec_dense_upload: a symmetric matrix must be square");
        MISPEC_REQUIRE(ctx->comm.allgather == nullptr, "mispec_dense_upload: dense operators cannot be row-sharded");
        const bool lower = (uplo == 'L' || uplo == 'l');
        ctx->make_current();
        auto D = std::make_unique<mispec_dense>();
        D->ctx = ctx;
        D->rows = rows;
        D->cols = cols;
        D->ld = round_up(std::max<int64_t>(cols, 1), 2);
        std::vector<double> rm(size_t(std::max<int64_t>(rows, 1)) * size_t(D->ld), 0.0);
        auto in = [&](int64_t i, int64_t j) { return row_major ? data_host[i * ld_host + j] : data_host[j * ld_host + i]; };
        for (int64_t i = 0; i < rows; i++)
            for (int64_t j = 0; j < cols; j++)
            {
                double v;
                if (!sym)
                    v = in(i, j);
                else if (lower)
                    v = i >= j ? in(i, j) : in(j, i);  // selfadjointView<Lower>: the strict upper triangle is never read
                else
                    v = i <= j ? in(i, j) : in(j, i);
                rm[size_t(i) * size_t(D->ld) + size_t(j)] = v;
            }
        D->a.alloc(rm.size());
        MISPEC_HIP(hipMemcpy(D->a.p, rm.data(), rm.size() * sizeof(double), hipMemcpyHostToDevice));
        *out = D.release();
    });
}

extern "C" int mispec_dense_destroy(mispec_dense* D)
{
    return guarded([&] {
        if (D)
        {
            D->ctx->make_current();
            delete D;
        }
    });
}
extern "C" int64_t mispec_dense_rows(const mispec_dense* D) { return D ? D->rows : 0; }
extern "C" int64_t mispec_dense_cols(const mispec_dense* D) { return D ? D->cols : 0; }

extern "C" int mispec_dense_gemv(const mispec_dense* D, const double* x_dev, double* y_dev)
{
    return guarded([&] {
        MISPEC_REQUIRE(D && x_dev && y_dev, "mispec_dense_gemv: NULL argument");
        D->ctx->make_current();
        launch_row_gemv(*D->ctx, D->a.p, D->ld, D->rows, D->cols, x_dev, y_dev, true);
    });
}

extern "C" int mispec_dense_gemm_host(const mispec_dense* D, const double* X_host, int64_t ldx, int k, double* Y_host, int64_t ldy)
{
    return guarded([&] {
        MISPEC_REQUIRE(D && X_host && Y_host && k >= 0, "mispec_dense_gemm_host: bad argument");
        MISPEC_REQUIRE(ldx >= D->cols && ldy >= D->rows, "mispec_dense_gemm_host: leading dimension too small");
        D->ctx->make_current();
        hipStream_t s = D->ctx->stream;
        ensure_stage(*D);
        for (int c = 0; c < k; c++)
        {
            MISPEC_HIP(hipMemcpyAsync(D->stage_x.p, X_host + int64_t(c) * ldx, size_t(D->cols) * sizeof(double), hipMemcpyHostToDevice, s));
            launch_row_gemv(*D->ctx, D->a.p, D->ld, D->rows, D->cols, D->stage_x.p, D->stage_y.p, true);
            MISPEC_HIP(hipMemcpyAsync(Y_host + int64_t(c) * ldy, D->stage_y.p, size_t(D->rows) * sizeof(double), hipMemcpyDeviceToHost, s));
            MISPEC_HIP(hipStreamSynchronize(s));
        }
    });
}

extern "C" int mispec_dense_gemv_host(const mispec_dense* D, const double* x_host, double* y_host)
{
    return mispec_dense_gemm_host(D, x_host, D ? D->cols : 0, 1, y_host, D ? D->rows : 0);
}

extern "C" int mispec_dense_coeff(const mispec_dense* D, int64_t i, int64_t j, double* out)
{
    return guarded([&] {
        MISPEC_REQUIRE(D && out, "mispec_dense_coeff: NULL argument");
        MISPEC_REQUIRE(i >= 0 && i < D->rows && j >= 0 && j < D->cols, "mispec_dense_coeff: index out of range");
        D->ctx->make_current();
        MISPEC_HIP(hipMemcpy(out, D->a.p + i * D->ld + j, sizeof(double), hipMemcpyDeviceToHost));
    });
}

extern "C" int mispec_dense_gemv_time(const mispec_dense* D, const double* x_dev, double* y_dev, int reps, float* ms_per_launch)
{
    return guarded([&] {
        MISPEC_REQUIRE(D && x_dev && y_dev && reps > 0 && ms_per_launch, "mispec_dense_gemv_time: bad argument");
        D->ctx->make_current();
        hipEvent_t e0, e1;
        MISPEC_HIP(hipEventCreate(&e0));
        MISPEC_HIP(hipEventCreate(&e1));
        MISPEC_HIP(hipEventRecord(e0, D->ctx->stream));
        for (int i = 0; i < reps; i++)
            launch_row_gemv(*D->ctx, D->a.p, D->ld, D->rows, D->cols, x_dev, y_dev, true);
        MISPEC_HIP(hipEventRecord(e1, D->ctx->stream));
        MISPEC_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        MISPEC_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void) hipEventDestroy(e0);
        (void) hipEventDestroy(e1);
        *ms_per_launch = ms / float(reps);
    });
}
