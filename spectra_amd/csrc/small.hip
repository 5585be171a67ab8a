// ncv x ncv kernels of the implicit restart: one workgroup (one wavefront), everything in LDS.
// They are latency-bound (no roofline): what they buy is that H, the rotations and Q never leave
// the device between the factorisation and compress_V (Q is consumed by the V*Q kernel straight
// from HBM, 12.8 KB at ncv = 40), and only 2*ncv doubles cross PCIe per restart.
#include "small.hpp"

#include <Spectra/internal/SmallDense.h>

using namespace mispec;

namespace {

// diag/subd: in = tridiagonal H, out = eigenvalues (diag).  evecs: n x n column-major.
__global__ __launch_bounds__(64) void k_tridiag_eigen(int n, const double* __restrict__ diag_in,
                                                       const double* __restrict__ subd_in, double* __restrict__ evals,
                                                       double* __restrict__ evecs, int* __restrict__ info)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* diag = sm;
    double* subd = sm + n;
    double* Q = sm + 2 * n;
    const int lane = threadIdx.x;
    for (int i = lane; i < n; i += 64)
    {
        diag[i] = diag_in[i];
        subd[i] = (i < n - 1) ? subd_in[i] : 0.0;
    }
    for (int idx = lane; idx < n * n; idx += 64)
        Q[idx] = (idx / n == idx % n) ? 1.0 : 0.0;
    __syncthreads();
    const int rc = small::tridiag_eigen(n, diag, subd, Q, n, small::Lanes{lane, 64});
    __syncthreads();
    for (int i = lane; i < n; i += 64)
        evals[i] = diag[i];
    for (int idx = lane; idx < n * n; idx += 64)
        evecs[idx] = Q[idx];
    if (lane == 0)
        *info = rc;
}

// Applies nshift shifted-QR steps to the tridiagonal (diag, subd) and accumulates Q (m x m).
__global__ __launch_bounds__(64) void k_restart_sym(int m, double* __restrict__ diag_io, double* __restrict__ subd_io,
                                                     ShiftList shifts, int nshift, double* __restrict__ Qout)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* diag = sm;
    double* subd = sm + m;
    double* work = sm + 2 * m;  // 4m
    double* Q = sm + 6 * m;     // m*m
    const int lane = threadIdx.x;
    for (int i = lane; i < m; i += 64)
    {
        diag[i] = diag_io[i];
        subd[i] = (i < m - 1) ? subd_io[i] : 0.0;
    }
    for (int idx = lane; idx < m * m; idx += 64)
        Q[idx] = (idx / m == idx % m) ? 1.0 : 0.0;
    __syncthreads();
    for (int s = 0; s < nshift; s++)
        small::tridiag_shifted_qr(m, diag, subd, shifts.mu[s], Q, m, m, work, small::Lanes{lane, 64});
    __syncthreads();
    for (int i = lane; i < m; i += 64)
    {
        diag_io[i] = diag[i];
        if (i < m - 1)
            subd_io[i] = subd[i];
    }
    for (int idx = lane; idx < m * m; idx += 64)
        Qout[idx] = Q[idx];
}

}  // namespace

namespace mispec {

void launch_tridiag_eigen(const mispec_ctx& ctx, int n, const double* diag, const double* subd, double* evals, double* evecs,
                          int* info)
{
    MISPEC_REQUIRE(n >= 1 && n <= kMaxSmallDim, "tridiag_eigen kernel: dimension out of range");
    const size_t lds = (size_t(2) * n + size_t(n) * n) * sizeof(double);
    MISPEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tridiag_eigen), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   int(lds)));
    hipLaunchKernelGGL(k_tridiag_eigen, dim3(1), dim3(64), lds, ctx.stream, n, diag, subd, evals, evecs, info);
    MISPEC_HIP(hipGetLastError());
}

void launch_restart_sym(const mispec_ctx& ctx, int m, double* diag, double* subd, const double* shifts_host, int nshift,
                        double* Q)
{
    MISPEC_REQUIRE(m >= 2 && m <= kMaxSmallDim, "restart kernel: dimension out of range");
    MISPEC_REQUIRE(nshift >= 0 && nshift <= kMaxShifts, "restart kernel: too many shifts");
    ShiftList sl;
    for (int i = 0; i < nshift; i++)
        sl.mu[i] = shifts_host[i];
    const size_t lds = (size_t(6) * m + size_t(m) * m) * sizeof(double);
    MISPEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_restart_sym), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   int(lds)));
    hipLaunchKernelGGL(k_restart_sym, dim3(1), dim3(64), lds, ctx.stream, m, diag, subd, sl, nshift, Q);
    MISPEC_HIP(hipGetLastError());
}

}  // namespace mispec

// ---- stand-alone unit-test entry points (mirror test/QR.cpp "QR of real tridiagonal matrix" and
// ---- test/Eigen.cpp "Eigen decomposition of symmetric real tridiagonal matrix") ---------------------
namespace {
struct SmallBufs
{
    DevBuf<double> diag, subd, evals, mat;
    DevBuf<int> info;
};
void split_tridiag(int n, const double* T, std::vector<double>& d, std::vector<double>& e)
{
    d.resize(size_t(n));
    e.assign(size_t(n), 0.0);
    for (int i = 0; i < n; i++)
        d[size_t(i)] = T[size_t(i) * n + i];
    for (int i = 0; i < n - 1; i++)
        e[size_t(i)] = T[size_t(i) * n + i + 1];  // T(i+1, i): column i, row i+1
}
}  // namespace

extern "C" int mispec_tridiag_qr(mispec_ctx* ctx, int n, const double* T_host, double shift, double* Q_host, double* QtHQ_host)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && T_host && n >= 2, "mispec_tridiag_qr: bad argument");
        ctx->make_current();
        std::vector<double> d, e;
        split_tridiag(n, T_host, d, e);
        SmallBufs b;
        b.diag.alloc(size_t(n));
        b.subd.alloc(size_t(n));
        b.mat.alloc(size_t(n) * n);
        MISPEC_HIP(hipMemcpyAsync(b.diag.p, d.data(), size_t(n) * 8, hipMemcpyHostToDevice, ctx->stream));
        MISPEC_HIP(hipMemcpyAsync(b.subd.p, e.data(), size_t(n) * 8, hipMemcpyHostToDevice, ctx->stream));
        launch_restart_sym(*ctx, n, b.diag.p, b.subd.p, &shift, 1, b.mat.p);
        MISPEC_HIP(hipMemcpyAsync(d.data(), b.diag.p, size_t(n) * 8, hipMemcpyDeviceToHost, ctx->stream));
        MISPEC_HIP(hipMemcpyAsync(e.data(), b.subd.p, size_t(n) * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (Q_host)
            MISPEC_HIP(hipMemcpyAsync(Q_host, b.mat.p, size_t(n) * n * 8, hipMemcpyDeviceToHost, ctx->stream));
        MISPEC_HIP(hipStreamSynchronize(ctx->stream));
        if (QtHQ_host)
        {
            for (size_t i = 0; i < size_t(n) * n; i++)
                QtHQ_host[i] = 0.0;
            for (int i = 0; i < n; i++)
                QtHQ_host[size_t(i) * n + i] = d[size_t(i)];
            for (int i = 0; i < n - 1; i++)
            {
                QtHQ_host[size_t(i) * n + i + 1] = e[size_t(i)];
                QtHQ_host[size_t(i + 1) * n + i] = e[size_t(i)];
            }
        }
    });
}

extern "C" int mispec_tridiag_eigen(mispec_ctx* ctx, int n, const double* T_host, double* evals_host, double* evecs_host)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && T_host && evals_host && n >= 1, "mispec_tridiag_eigen: bad argument");
        ctx->make_current();
        std::vector<double> d, e;
        split_tridiag(n, T_host, d, e);
        SmallBufs b;
        b.diag.alloc(size_t(n));
        b.subd.alloc(size_t(n));
        b.evals.alloc(size_t(n));
        b.mat.alloc(size_t(n) * n);
        b.info.alloc(1);
        MISPEC_HIP(hipMemcpyAsync(b.diag.p, d.data(), size_t(n) * 8, hipMemcpyHostToDevice, ctx->stream));
        MISPEC_HIP(hipMemcpyAsync(b.subd.p, e.data(), size_t(n) * 8, hipMemcpyHostToDevice, ctx->stream));
        launch_tridiag_eigen(*ctx, n, b.diag.p, b.subd.p, b.evals.p, b.mat.p, b.info.p);
        int info = 0;
        MISPEC_HIP(hipMemcpyAsync(evals_host, b.evals.p, size_t(n) * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (evecs_host)
            MISPEC_HIP(hipMemcpyAsync(evecs_host, b.mat.p, size_t(n) * n * 8, hipMemcpyDeviceToHost, ctx->stream));
        MISPEC_HIP(hipMemcpyAsync(&info, b.info.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        MISPEC_HIP(hipStreamSynchronize(ctx->stream));
        if (info != 0)
            throw Error(MISPEC_ERUNTIME, "TridiagEigen: eigen decomposition failed");
    });
}
