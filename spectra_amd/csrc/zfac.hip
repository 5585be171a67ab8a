// Complex-scalar Arnoldi / Lanczos factorisation on the device: the HIP backend of zfac_flow.hpp and its C entry points
// (include/mispec_extras.h).  OUTSIDE the hot path of SURVEY.md section 8 — the configs are real fp64 — and deliberately plain:
// host-driven steps in the reference's order, one kernel per vector primitive, no fusion.  It exists so that the reference's
// factorisation templates keep their complex instantiations (LinAlg/Arnoldi.h, LinAlg/Lanczos.h over DenseGenMatProd<complex> /
// DenseHermMatProd<complex>; test/Arnoldi.cpp:122-158) with the basis in HBM rather than on a CPU fallback.
//
// Layout: complex numbers interleaved (re, im) = double2, 16-byte loads; V is n x m column-major with leading dimension n; a dense
// operator is stored column-major (one thread per row reads a column slice coalesced), a Hermitian input given by one triangle is
// mirrored at upload with the diagonal's imaginary part dropped, as selfadjointView reads it.
// Every primitive is HBM-bound at 16 bytes per entry touched; reductions are per-column workgroups with a fixed LDS tree
// (deterministic, independent of the launch geometry).
#include <complex>
#include <memory>
#include <vector>

#include "common.hpp"
#include "zfac_flow.hpp"

using namespace mispec;
using cd = std::complex<double>;

namespace {

constexpr int kThreads = 256;

__device__ inline double2 zmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ inline double2 zmulc(double2 a, double2 b) { return make_double2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x); }  // conj(a) b

// out[j] = X[:, j]^H y, one workgroup per column
__global__ __launch_bounds__(kThreads) void k_zdotc(int64_t n, const double2* __restrict__ X, int64_t ldx, const double2* __restrict__ y,
                                                     double2* __restrict__ out)
{
    __shared__ double sre[kThreads], sim[kThreads];
    const double2* x = X + int64_t(blockIdx.x) * ldx;
    double re = 0.0, im = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += kThreads)
    {
        const double2 p = zmulc(x[i], y[i]);
        re += p.x;
        im += p.y;
    }
    sre[threadIdx.x] = re;
    sim[threadIdx.x] = im;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1)
    {
        if (int(threadIdx.x) < s)
        {
            sre[threadIdx.x] += sre[threadIdx.x + s];
            sim[threadIdx.x] += sim[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
        out[blockIdx.x] = make_double2(sre[0], sim[0]);
}

// f = w - V[:, :ncols] h  (w may alias f: each thread reads its own row before it writes it)
__global__ __launch_bounds__(kThreads) void k_zupdate(int64_t n, double2* f, const double2* w, const double2* __restrict__ V, int64_t ldv,
                                                       int ncols, const double2* __restrict__ h)
{
    const int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x;
    if (i >= n)
        return;
    double2 acc = w[i];
    for (int j = 0; j < ncols; j++)
    {
        const double2 p = zmul(V[i + int64_t(j) * ldv], h[j]);
        acc.x -= p.x;
        acc.y -= p.y;
    }
    f[i] = acc;
}

// y = A x, A column-major rows x cols with leading dimension ld: one thread per row
__global__ __launch_bounds__(kThreads) void k_zgemv(int64_t rows, int64_t cols, const double2* __restrict__ A, int64_t ld,
                                                     const double2* __restrict__ x, double2* __restrict__ y)
{
    const int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x;
    if (i >= rows)
        return;
    double2 acc = make_double2(0.0, 0.0);
    for (int64_t j = 0; j < cols; j++)
    {
        const double2 p = zmul(A[i + j * ld], x[j]);
        acc.x += p.x;
        acc.y += p.y;
    }
    y[i] = acc;
}

__global__ __launch_bounds__(kThreads) void k_zscale_copy(int64_t n, double2* dst, const double2* src, double alpha)
{
    const int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x;
    if (i < n)
    {
        const double2 v = src[i];
        dst[i] = make_double2(alpha * v.x, alpha * v.y);
    }
}

__global__ __launch_bounds__(kThreads) void k_zaxpy(int64_t n, double2* __restrict__ y, double2 a, const double2* __restrict__ x)
{
    const int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x;
    if (i < n)
    {
        const double2 p = zmul(a, x[i]);
        y[i] = make_double2(y[i].x + p.x, y[i].y + p.y);
    }
}

// max_i |x_i| (a single workgroup: called once per init)
__global__ __launch_bounds__(kThreads) void k_zabsmax(int64_t n, const double2* __restrict__ x, double* __restrict__ out)
{
    __shared__ double smax[kThreads];
    double m = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += kThreads)
        m = fmax(m, hypot(x[i].x, x[i].y));
    smax[threadIdx.x] = m;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1)
    {
        if (int(threadIdx.x) < s)
            smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0)
        out[0] = smax[0];
}

inline unsigned blocks_for(int64_t n) { return unsigned((n + kThreads - 1) / kThreads); }
inline double2* z2(cd* p) { return reinterpret_cast<double2*>(p); }
inline const double2* z2(const cd* p) { return reinterpret_cast<const double2*>(p); }

}  // namespace

struct mispec_zdense
{
    mispec_ctx* ctx = nullptr;
    int64_t rows = 0, cols = 0;
    DevBuf<double2> a;  // column-major, leading dimension rows
    std::vector<cd> host;  // the same entries, for operator()
    mutable DevBuf<double2> stage_x, stage_y;
};

namespace {

// The primitives of zfac_flow.hpp on the context's stream.
struct HipBackend
{
    mispec_ctx* ctx = nullptr;
    int64_t n = 0;
    const mispec_zdense* dense = nullptr;
    mispec_zop_fn op = nullptr;
    void* op_user = nullptr;
    DevBuf<double2> small;           // reduction results / coefficient vectors on the device
    PinnedBuf<double2> small_host;   // their host images
    PinnedBuf<double2> stage_x, stage_y;  // host-pointer operator
    DevBuf<double> scalar;

    void setup(int m)
    {
        small.alloc(size_t(m) + 1);
        small_host.alloc(size_t(m) + 1);
        scalar.alloc(1);
        if (op)
        {
            stage_x.alloc(size_t(n));
            stage_y.alloc(size_t(n));
        }
    }
    hipStream_t s() const { return ctx->stream; }

    cd* alloc(size_t count)
    {
        void* p = nullptr;
        MISPEC_HIP(hipMalloc(&p, count * sizeof(double2)));
        return static_cast<cd*>(p);
    }
    void release(cd* p)
    {
        if (p)
            (void) hipFree(p);
    }
    void upload(cd* dev, const cd* host, int64_t count)
    {
        MISPEC_HIP(hipMemcpyAsync(dev, host, size_t(count) * sizeof(double2), hipMemcpyHostToDevice, s()));
        MISPEC_HIP(hipStreamSynchronize(s()));
    }
    void download(cd* host, const cd* dev, int64_t count)
    {
        MISPEC_HIP(hipMemcpyAsync(host, dev, size_t(count) * sizeof(double2), hipMemcpyDeviceToHost, s()));
        MISPEC_HIP(hipStreamSynchronize(s()));
    }
    void apply(const cd* x, cd* y)
    {
        if (dense)
        {
            hipLaunchKernelGGL(k_zgemv, dim3(blocks_for(n)), dim3(kThreads), 0, s(), dense->rows, dense->cols, dense->a.p, dense->rows, z2(x),
                               z2(y));
            MISPEC_HIP(hipGetLastError());
            return;
        }
        // the reference's contract: perform_op(const Scalar* x_in, Scalar* y_out) on host pointers
        MISPEC_HIP(hipMemcpyAsync(stage_x.p, x, size_t(n) * sizeof(double2), hipMemcpyDeviceToHost, s()));
        MISPEC_HIP(hipStreamSynchronize(s()));
        if (op(op_user, reinterpret_cast<const double*>(stage_x.p), reinterpret_cast<double*>(stage_y.p)) != 0)
            throw Error(MISPEC_ERUNTIME, "complex factorisation: the user operator reported an error");
        MISPEC_HIP(hipMemcpyAsync(y, stage_y.p, size_t(n) * sizeof(double2), hipMemcpyHostToDevice, s()));
        MISPEC_HIP(hipStreamSynchronize(s()));
    }
    void dotc(const cd* X, int64_t ldx, int ncols, const cd* y, cd* out_host)
    {
        if (ncols <= 0)
            return;
        hipLaunchKernelGGL(k_zdotc, dim3(unsigned(ncols)), dim3(kThreads), 0, s(), n, z2(X), ldx, z2(y), small.p);
        MISPEC_HIP(hipGetLastError());
        MISPEC_HIP(hipMemcpyAsync(small_host.p, small.p, size_t(ncols) * sizeof(double2), hipMemcpyDeviceToHost, s()));
        MISPEC_HIP(hipStreamSynchronize(s()));
        for (int j = 0; j < ncols; j++)
            out_host[j] = cd(small_host.p[j].x, small_host.p[j].y);
    }
    void update(cd* f, const cd* w, const cd* V, int64_t ldv, int ncols, const cd* h_host)
    {
        for (int j = 0; j < ncols; j++)
            small_host.p[j] = make_double2(h_host[j].real(), h_host[j].imag());
        if (ncols > 0)
            MISPEC_HIP(hipMemcpyAsync(small.p, small_host.p, size_t(ncols) * sizeof(double2), hipMemcpyHostToDevice, s()));
        hipLaunchKernelGGL(k_zupdate, dim3(blocks_for(n)), dim3(kThreads), 0, s(), n, z2(f), z2(w), z2(V), ldv, ncols, small.p);
        MISPEC_HIP(hipGetLastError());
        MISPEC_HIP(hipStreamSynchronize(s()));  // small_host is rewritten by the next call
    }
    void scale_copy(cd* dst, const cd* src, double alpha)
    {
        hipLaunchKernelGGL(k_zscale_copy, dim3(blocks_for(n)), dim3(kThreads), 0, s(), n, z2(dst), z2(src), alpha);
        MISPEC_HIP(hipGetLastError());
    }
    void axpy(cd* y, cd a, const cd* x)
    {
        hipLaunchKernelGGL(k_zaxpy, dim3(blocks_for(n)), dim3(kThreads), 0, s(), n, z2(y), make_double2(a.real(), a.imag()), z2(x));
        MISPEC_HIP(hipGetLastError());
    }
    double norm(const cd* x)
    {
        cd r;
        dotc(x, n, 1, x, &r);
        return std::sqrt(r.real());
    }
    double absmax(const cd* x)
    {
        hipLaunchKernelGGL(k_zabsmax, dim3(1), dim3(kThreads), 0, s(), n, z2(x), scalar.p);
        MISPEC_HIP(hipGetLastError());
        double v = 0.0;
        MISPEC_HIP(hipMemcpyAsync(&v, scalar.p, sizeof(double), hipMemcpyDeviceToHost, s()));
        MISPEC_HIP(hipStreamSynchronize(s()));
        return v;
    }
    void zero(cd* x) { MISPEC_HIP(hipMemsetAsync(x, 0, size_t(n) * sizeof(double2), s())); }
};

}  // namespace

struct mispec_zfac
{
    HipBackend be;
    std::unique_ptr<ZFacFlow<HipBackend>> flow;
};

namespace {

mispec_zfac* make_zfac(mispec_ctx* ctx, int64_t n, int ncv, int hermitian, const mispec_zdense* D, mispec_zop_fn op, void* user)
{
    MISPEC_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<mispec_zfac> F(new mispec_zfac);
    F->be.ctx = ctx;
    F->be.n = n;
    F->be.dense = D;
    F->be.op = op;
    F->be.op_user = user;
    F->be.setup(ncv);
    F->flow.reset(new ZFacFlow<HipBackend>(F->be, n, ncv, hermitian != 0));
    return F.release();
}

}  // namespace

// =================================================================================================
// C ABI (include/mispec_extras.h)
// =================================================================================================
extern "C" int mispec_zdense_upload(mispec_ctx* ctx, int64_t rows, int64_t cols, const double* data_host, int64_t ld_host,
                                    int row_major, char uplo, mispec_zdense** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && out && rows >= 0 && cols >= 0, "mispec_zdense_upload: bad argument");
        MISPEC_REQUIRE(data_host || rows * cols == 0, "mispec_zdense_upload: NULL matrix");
        MISPEC_REQUIRE(ld_host >= (row_major ? cols : rows), "mispec_zdense_upload: leading dimension too small");
        MISPEC_REQUIRE(uplo == 0 || uplo == 'L' || uplo == 'U', "mispec_zdense_upload: uplo must be 0, 'L' or 'U'");
        MISPEC_REQUIRE(uplo == 0 || rows == cols, "mispec_zdense_upload: a Hermitian matrix must be square");
        MISPEC_HIP(hipSetDevice(ctx->device));
        std::unique_ptr<mispec_zdense> D(new mispec_zdense);
        D->ctx = ctx;
        D->rows = rows;
        D->cols = cols;
        D->host.resize(size_t(rows) * size_t(cols));
        zdense_expand(rows, cols, reinterpret_cast<const cd*>(data_host), ld_host, row_major != 0, uplo, D->host.data());
        D->a.alloc(D->host.size());
        if (!D->host.empty())
            MISPEC_HIP(hipMemcpy(D->a.p, D->host.data(), D->host.size() * sizeof(double2), hipMemcpyHostToDevice));
        *out = D.release();
    });
}

extern "C" int mispec_zdense_destroy(mispec_zdense* D)
{
    return guarded([&] { delete D; });
}

extern "C" int64_t mispec_zdense_rows(const mispec_zdense* D) { return D ? D->rows : 0; }
extern "C" int64_t mispec_zdense_cols(const mispec_zdense* D) { return D ? D->cols : 0; }

extern "C" int mispec_zdense_gemv_host(const mispec_zdense* D, const double* x_host, double* y_host)
{
    return guarded([&] {
        MISPEC_REQUIRE(D && x_host && y_host, "mispec_zdense_gemv_host: NULL argument");
        MISPEC_HIP(hipSetDevice(D->ctx->device));
        if (D->stage_x.n < size_t(D->cols))
            D->stage_x.alloc(size_t(D->cols));
        if (D->stage_y.n < size_t(D->rows))
            D->stage_y.alloc(size_t(D->rows));
        hipStream_t s = D->ctx->stream;
        MISPEC_HIP(hipMemcpyAsync(D->stage_x.p, x_host, size_t(D->cols) * sizeof(double2), hipMemcpyHostToDevice, s));
        if (D->rows > 0)
        {
            hipLaunchKernelGGL(k_zgemv, dim3(blocks_for(D->rows)), dim3(kThreads), 0, s, D->rows, D->cols, D->a.p, D->rows, D->stage_x.p,
                               D->stage_y.p);
            MISPEC_HIP(hipGetLastError());
        }
        MISPEC_HIP(hipMemcpyAsync(y_host, D->stage_y.p, size_t(D->rows) * sizeof(double2), hipMemcpyDeviceToHost, s));
        MISPEC_HIP(hipStreamSynchronize(s));
    });
}

extern "C" int mispec_zdense_coeff(const mispec_zdense* D, int64_t i, int64_t j, double* out_re_im)
{
    return guarded([&] {
        MISPEC_REQUIRE(D && out_re_im, "mispec_zdense_coeff: NULL argument");
        MISPEC_REQUIRE(i >= 0 && i < D->rows && j >= 0 && j < D->cols, "mispec_zdense_coeff: index out of range");
        const cd v = D->host[size_t(j) * size_t(D->rows) + size_t(i)];
        out_re_im[0] = v.real();
        out_re_im[1] = v.imag();
    });
}

extern "C" int mispec_zfac_create_dense(mispec_ctx* ctx, const mispec_zdense* D, int ncv, int hermitian, mispec_zfac** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && D && out, "mispec_zfac_create_dense: NULL argument");
        MISPEC_REQUIRE(D->rows == D->cols, "mispec_zfac_create_dense: the matrix must be square");
        *out = make_zfac(ctx, D->rows, ncv, hermitian, D, nullptr, nullptr);
    });
}

extern "C" int mispec_zfac_create_op(mispec_ctx* ctx, mispec_zop_fn op, void* op_user, int64_t n, int ncv, int hermitian,
                                     mispec_zfac** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && op && out, "mispec_zfac_create_op: NULL argument");
        *out = make_zfac(ctx, n, ncv, hermitian, nullptr, op, op_user);
    });
}

extern "C" int mispec_zfac_destroy(mispec_zfac* F)
{
    return guarded([&] { delete F; });
}

extern "C" int mispec_zfac_init(mispec_zfac* F, const double* v0_host, int64_t* op_counter)
{
    return guarded([&] {
        MISPEC_REQUIRE(F && v0_host && op_counter, "mispec_zfac_init: NULL argument");
        MISPEC_HIP(hipSetDevice(F->be.ctx->device));
        F->flow->init(reinterpret_cast<const cd*>(v0_host), *op_counter);
    });
}

extern "C" int mispec_zfac_factorize(mispec_zfac* F, int from_k, int to_m, int64_t* op_counter)
{
    return guarded([&] {
        MISPEC_REQUIRE(F && op_counter, "mispec_zfac_factorize: NULL argument");
        MISPEC_HIP(hipSetDevice(F->be.ctx->device));
        F->flow->factorize_from(from_k, to_m, *op_counter);
    });
}

extern "C" int mispec_zfac_subspace_dim(const mispec_zfac* F) { return F ? F->flow->subspace_dim() : 0; }

extern "C" int mispec_zfac_f_norm(const mispec_zfac* F, double* out)
{
    return guarded([&] {
        MISPEC_REQUIRE(F && out, "mispec_zfac_f_norm: NULL argument");
        *out = F->flow->f_norm();
    });
}

extern "C" int mispec_zfac_get_H(const mispec_zfac* F, double* H_host)
{
    return guarded([&] {
        MISPEC_REQUIRE(F && H_host, "mispec_zfac_get_H: NULL argument");
        const std::vector<cd>& H = F->flow->matrix_H();
        std::copy(H.begin(), H.end(), reinterpret_cast<cd*>(H_host));
    });
}

extern "C" int mispec_zfac_get_V(const mispec_zfac* F, int ncols, double* V_host)
{
    return guarded([&] {
        MISPEC_REQUIRE(F && V_host && ncols >= 0 && ncols <= F->flow->max_dim(), "mispec_zfac_get_V: bad argument");
        MISPEC_HIP(hipSetDevice(F->be.ctx->device));
        F->flow->get_V(reinterpret_cast<cd*>(V_host), ncols);
    });
}

extern "C" int mispec_zfac_get_f(const mispec_zfac* F, double* f_host)
{
    return guarded([&] {
        MISPEC_REQUIRE(F && f_host, "mispec_zfac_get_f: NULL argument");
        MISPEC_HIP(hipSetDevice(F->be.ctx->device));
        F->flow->get_f(reinterpret_cast<cd*>(f_host));
    });
}
