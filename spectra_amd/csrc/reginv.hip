// SparseRegularInverse on the GPU — the B operator of SymGEigsSolver<..., GEigsMode::RegularInverse>
// (replaces MatOp/SparseRegularInverse.h:55-127, which holds an Eigen::ConjugateGradient<SparseMatrix>):
//   perform_op  y = selfadjointView<Uplo>(B) x          -> the CSR-stream SpMV on the mirrored triangle
//   solve       y = B^{-1} x, conjugate gradient with the defaults the reference inherits from Eigen 3.4.0
//               (IterativeLinearSolvers/ConjugateGradient.h): Jacobi preconditioner, start vector 0, tolerance
//               epsilon on |r|/|b| with |r|^2 taken from the recurrence, at most 2n iterations.
// Each iteration is one SpMV and two fused vector kernels; the dot products are two-stage fixed-order reductions
// (reproducible), read back once per iteration for alpha and the stopping test.
#include "reginv.hpp"

#include <cmath>
#include <limits>
#include <memory>

#include "krylov.hpp"

using namespace mispec;

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ double block_sum(double v, double* red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// r = rhs ; x = 0 ; p = invdiag .* r ; partials: [0][b] = sum r^2, [1][b] = sum r p
__global__ __launch_bounds__(kThreads) void k_cg_start(const double* __restrict__ rhs, const double* __restrict__ invdiag,
                                                        double* __restrict__ r, double* __restrict__ p, double* __restrict__ x,
                                                        int64_t n, double* __restrict__ partials)
{
    __shared__ double red[4];
    double rr = 0.0, rp = 0.0;
    for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < n; i += int64_t(gridDim.x) * kThreads)
    {
        const double ri = rhs[i], pi = invdiag[i] * ri;
        r[i] = ri;
        p[i] = pi;
        x[i] = 0.0;
        rr += ri * ri;
        rp += ri * pi;
    }
    rr = block_sum(rr, red);
    rp = block_sum(rp, red);
    if (threadIdx.x == 0)
    {
        partials[blockIdx.x] = rr;
        partials[gridDim.x + blockIdx.x] = rp;
    }
}

// partials[0][b] = sum a b
__global__ __launch_bounds__(kThreads) void k_dot(const double* __restrict__ a, const double* __restrict__ b, int64_t n,
                                                   double* __restrict__ partials)
{
    __shared__ double red[4];
    double s = 0.0;
    for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < n; i += int64_t(gridDim.x) * kThreads)
        s += a[i] * b[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0)
    {
        partials[blockIdx.x] = s;
        partials[gridDim.x + blockIdx.x] = 0.0;
    }
}

// x += alpha p ; r -= alpha t ; z = invdiag .* r ; partials: sum r^2, sum r z
__global__ __launch_bounds__(kThreads) void k_cg_update(double alpha, const double* __restrict__ p, const double* __restrict__ t,
                                                         const double* __restrict__ invdiag, double* __restrict__ x,
                                                         double* __restrict__ r, double* __restrict__ z, int64_t n,
                                                         double* __restrict__ partials)
{
    __shared__ double red[4];
    double rr = 0.0, rz = 0.0;
    for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < n; i += int64_t(gridDim.x) * kThreads)
    {
        x[i] += alpha * p[i];
        const double ri = r[i] - alpha * t[i];
        const double zi = invdiag[i] * ri;
        r[i] = ri;
        z[i] = zi;
        rr += ri * ri;
        rz += ri * zi;
    }
    rr = block_sum(rr, red);
    rz = block_sum(rz, red);
    if (threadIdx.x == 0)
    {
        partials[blockIdx.x] = rr;
        partials[gridDim.x + blockIdx.x] = rz;
    }
}

// p = z + beta p
__global__ __launch_bounds__(kThreads) void k_cg_direction(double beta, const double* __restrict__ z, double* __restrict__ p,
                                                            int64_t n)
{
    for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < n; i += int64_t(gridDim.x) * kThreads)
        p[i] = z[i] + beta * p[i];
}

// out[s] = sum_b partials[s][b] for the two slots, fixed order
__global__ __launch_bounds__(1024) void k_sum2(const double* __restrict__ partials, int nblocks, double* __restrict__ out)
{
    __shared__ double sh[2][16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 1024)
    {
        a += partials[i];
        b += partials[nblocks + i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
    {
        a += __shfl_down(a, off, 64);
        b += __shfl_down(b, off, 64);
    }
    if (lane == 0)
    {
        sh[0][w] = a;
        sh[1][w] = b;
    }
    __syncthreads();
    if (threadIdx.x < 2)
    {
        double s = 0.0;
        for (int k = 0; k < 16; k++)
            s += sh[threadIdx.x][k];
        out[threadIdx.x] = s;
    }
}

// diagonal of a CSR matrix -> 1/diag (1 where zero or absent): Eigen's DiagonalPreconditioner
__global__ __launch_bounds__(kThreads) void k_inv_diag(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                                                        const double* __restrict__ val, int64_t n, double* __restrict__ invdiag)
{
    const int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x;
    if (i >= n)
        return;
    double d = 0.0;
    for (int32_t p = rowptr[i]; p < rowptr[i + 1]; p++)
        if (colind[p] == i)
            d += val[p];
    invdiag[i] = (d != 0.0) ? 1.0 / d : 1.0;
}

// scalar slots of an orthogonalisation record: sum x y and max |x|
__global__ __launch_bounds__(kThreads) void k_dot_record(const double* __restrict__ x, const double* __restrict__ y, int64_t n,
                                                          double* __restrict__ partials, int64_t pstride)
{
    __shared__ double red[4];
    double s = 0.0, mx = 0.0;
    for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < n; i += int64_t(gridDim.x) * kThreads)
    {
        const double xi = x[i];
        s += xi * y[i];
        mx = fmax(mx, fabs(xi));
    }
    s = block_sum(s, red);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        mx = fmax(mx, __shfl_down(mx, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        partials[kSlotBeta2 * pstride + blockIdx.x] = s;
        partials[kSlotMaxAbs * pstride + blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    }
}

int cg_grid(const mispec_ctx& ctx, int64_t n)
{
    const int64_t blocks = (n + kThreads - 1) / kThreads;
    return int(std::max<int64_t>(1, std::min<int64_t>(blocks, int64_t(ctx.num_cu) * 4)));
}

// the two sums of the last kernel -> host
void read_sums(const mispec_reginv& R, int grid, double& a, double& b)
{
    hipLaunchKernelGGL(k_sum2, dim3(1), dim3(1024), 0, R.ctx->stream, R.partials.p, grid, R.scal.p);
    MISPEC_HIP(hipGetLastError());
    MISPEC_HIP(hipMemcpyAsync(R.h_scal.p, R.scal.p, 2 * sizeof(double), hipMemcpyDeviceToHost, R.ctx->stream));
    MISPEC_HIP(hipStreamSynchronize(R.ctx->stream));
    a = R.h_scal.p[0];
    b = R.h_scal.p[1];
}

}  // namespace

mispec_reginv::~mispec_reginv()
{
    if (B)
        (void) mispec_csr_destroy(B);
}

namespace mispec {

void launch_dot_record(const mispec_ctx& ctx, const double* x, const double* y, int64_t n, double* partials, int64_t pstride,
                       int nrec)
{
    hipLaunchKernelGGL(k_dot_record, dim3(unsigned(nrec)), dim3(kThreads), 0, ctx.stream, x, y, n, partials, pstride);
    MISPEC_HIP(hipGetLastError());
}

void reginv_solve(const mispec_reginv& R, const double* rhs, double* x)
{
    const mispec_ctx& ctx = *R.ctx;
    const int64_t n = R.n;
    const int grid = cg_grid(ctx, n);
    const dim3 g(static_cast<unsigned>(grid)), b(kThreads);
    R.last_iterations = 0;
    hipLaunchKernelGGL(k_cg_start, g, b, 0, ctx.stream, rhs, R.invdiag.p, R.r.p, R.p.p, x, n, R.partials.p);
    MISPEC_HIP(hipGetLastError());
    double rhs2, abs_new;
    read_sums(R, grid, rhs2, abs_new);
    if (rhs2 == 0.0)
        return;  // x = 0
    const double tol = std::numeric_limits<double>::epsilon();
    const double threshold = std::max(tol * tol * rhs2, std::numeric_limits<double>::min());
    double r2 = rhs2;
    if (r2 < threshold)
        return;
    const int64_t max_iters = 2 * n;
    int64_t it = 0;
    while (it < max_iters)
    {
        launch_spmv(*R.B, R.p.p, R.t.p, nullptr);
        hipLaunchKernelGGL(k_dot, g, b, 0, ctx.stream, R.p.p, R.t.p, n, R.partials.p);
        MISPEC_HIP(hipGetLastError());
        double pt, unused;
        read_sums(R, grid, pt, unused);
        const double alpha = abs_new / pt;
        hipLaunchKernelGGL(k_cg_update, g, b, 0, ctx.stream, alpha, R.p.p, R.t.p, R.invdiag.p, x, R.r.p, R.z.p, n, R.partials.p);
        MISPEC_HIP(hipGetLastError());
        double rz;
        read_sums(R, grid, r2, rz);
        if (r2 < threshold)
            break;
        const double abs_old = abs_new;
        abs_new = rz;
        const double beta = abs_new / abs_old;
        hipLaunchKernelGGL(k_cg_direction, g, b, 0, ctx.stream, beta, R.z.p, R.p.p, n);
        MISPEC_HIP(hipGetLastError());
        it++;
    }
    R.last_iterations = it;
    if (!(r2 < threshold))
        throw Error(MISPEC_ERUNTIME, "SparseRegularInverse: CG solver does not converge");  // SparseRegularInverse.h:113-114
}

}  // namespace mispec

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" int mispec_reginv_create(mispec_ctx* ctx, int64_t n, const int32_t* outer, const int32_t* inner, const double* val,
                                    char uplo, int row_major, mispec_reginv** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && out && outer && n >= 1, "mispec_reginv_create: bad argument");
        MISPEC_REQUIRE(ctx->comm.allgather == nullptr, "mispec_reginv_create: the B operator cannot be row-sharded");
        auto R = std::make_unique<mispec_reginv>();
        R->ctx = ctx;
        R->n = n;
        if (mispec_csr_from_triangle(ctx, n, outer, inner, val, uplo, row_major, &R->B) != MISPEC_OK)
            throw Error(MISPEC_EINVAL, mispec_last_error());
        ctx->make_current();
        const size_t np = size_t(n) + 2;
        R->invdiag.alloc(np);
        R->r.alloc(np);
        R->p.alloc(np);
        R->z.alloc(np);
        R->t.alloc(np);
        const int grid = cg_grid(*ctx, n);
        R->partials.alloc(2 * size_t(grid));
        R->scal.alloc(2);
        R->h_scal.alloc(2);
        hipLaunchKernelGGL(k_inv_diag, dim3(unsigned((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream, R->B->rowptr.p,
                           R->B->colind.p, R->B->val.p, n, R->invdiag.p);
        MISPEC_HIP(hipGetLastError());
        if (R->B->reordered())
        {
            // stored matrix = P B P': entry i of what was just computed belongs to row perm[i]; the CG iteration works in the
            // caller's order (launch_spmv), so the Jacobi preconditioner has to as well
            launch_from_stored_order(*R->B, R->invdiag.p, R->z.p);
            MISPEC_HIP(hipMemcpyAsync(R->invdiag.p, R->z.p, size_t(n) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        }
        MISPEC_HIP(hipStreamSynchronize(ctx->stream));
        *out = R.release();
    });
}

extern "C" int mispec_reginv_destroy(mispec_reginv* R)
{
    return guarded([&] {
        if (R)
        {
            R->ctx->make_current();
            delete R;
        }
    });
}

extern "C" int64_t mispec_reginv_rows(const mispec_reginv* R) { return R ? R->n : 0; }
extern "C" int64_t mispec_reginv_last_iterations(const mispec_reginv* R) { return R ? R->last_iterations : 0; }

extern "C" int mispec_reginv_perform_op_host(const mispec_reginv* R, const double* x_host, double* y_host)
{
    return R ? mispec_spmv_host(R->B, x_host, y_host) : MISPEC_EINVAL;
}

extern "C" int mispec_reginv_solve_host(const mispec_reginv* R, const double* x_host, double* y_host)
{
    return guarded([&] {
        MISPEC_REQUIRE(R && x_host && y_host, "mispec_reginv_solve_host: NULL argument");
        R->ctx->make_current();
        if (R->stage_x.n < size_t(R->n))
        {
            R->stage_x.alloc(size_t(R->n) + 2);
            R->stage_y.alloc(size_t(R->n) + 2);
        }
        MISPEC_HIP(hipMemcpyAsync(R->stage_x.p, x_host, size_t(R->n) * sizeof(double), hipMemcpyHostToDevice, R->ctx->stream));
        reginv_solve(*R, R->stage_x.p, R->stage_y.p);
        MISPEC_HIP(hipMemcpyAsync(y_host, R->stage_y.p, size_t(R->n) * sizeof(double), hipMemcpyDeviceToHost, R->ctx->stream));
        MISPEC_HIP(hipStreamSynchronize(R->ctx->stream));
    });
}
