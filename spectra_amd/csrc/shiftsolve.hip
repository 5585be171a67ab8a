// y = (A - sigma I)^{-1} x on the GPU for symmetric A — the operator behind SymEigsShiftSolver
// (replaces MatOp/SparseSymShiftSolve.h:85-109, which delegates to Eigen::SparseLU).
//
// Two factorisations, chosen from the half-bandwidth b of A - sigma I at set_shift():
//   * banded (b <= 8): a recursive "partition + Schur complement" LDL' — the parallel form of a band
//     solve.  The rows are cut into chunks of L rows; the last b rows of every chunk form a separator,
//     the rest (the interior) of different chunks are decoupled.  Factor (host, once per shift): banded
//     LDL' of every interior block, the spikes W = M_II^{-1} M_IS, and the Schur complement of the
//     separators, which is again banded (half-bandwidth 2b-1) and is factored the same way, recursively,
//     until it fits one chunk.  Solve (device, every Lanczos step), per level three kernels:
//        k_chunk_solve  one thread per chunk: forward/backward substitution on its interior block
//                       (factors stored chunk-interleaved => coalesced across the threads of a wave)
//        k_sep_rhs      g_S = f_S - M_SI y_I
//        k_back_subst   x_I = y_I - W x_S   (one thread per row, fully parallel)
//     The sequential depth per level is L rows instead of n.  No pivoting: stable when A - sigma I is
//     definite (sigma outside the spectrum, config 5); a vanishing pivot is reported as a failed
//     factorisation, like the reference does for a singular shift (SparseSymShiftSolve.h:93-94).
//   * dense (n <= 4096, any sparsity — the reference's own test fixtures are of this kind): LU with partial
//     pivoting of the dense A - sigma I on the host, explicit inverse, and a dense GEMV kernel per step.
// Anything else (large n with large bandwidth) is rejected: a general sparse LU on the GPU is out of scope.
#include "shiftsolve.hpp"
#include "dense.hpp"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstring>
#include <string>

using namespace mispec;

namespace {

constexpr int kThreads = 256;

// ---------------------------------------------------------------------------------------------------
// host-side band matrix (symmetric, lower band incl. diagonal): a(i, d) = M(i, i - d), 0 <= d <= b
// ---------------------------------------------------------------------------------------------------
struct HostBand
{
    int64_t n = 0;
    int b = 0;
    std::vector<double> a;  // n x (b + 1), row-major
    // read-only view of an unshifted band kept elsewhere (the operator's resident copy): M = *view - shift * I.
    // Used for the top level when it is factored on the device, so that no shifted host copy has to be built.
    const std::vector<double>* view = nullptr;
    const double* view_dev = nullptr;  // the same unshifted band in device memory
    const std::vector<double>* viewB = nullptr;  // pencil: M = *view - shift * *viewB (same layout); nullptr: B = I
    const double* viewB_dev = nullptr;
    double shift = 0.0;
    double& at(int64_t i, int d) { return a[size_t(i) * (b + 1) + d]; }
    double at(int64_t i, int d) const
    {
        if (view)
            return (*view)[size_t(i) * (b + 1) + d] - shift * (viewB ? (*viewB)[size_t(i) * (b + 1) + d] : (d == 0 ? 1.0 : 0.0));
        return a[size_t(i) * (b + 1) + d];
    }
    double get(int64_t i, int64_t j) const  // symmetric access, 0 outside the band
    {
        if (i < j)
            std::swap(i, j);
        const int64_t d = i - j;
        return d <= b ? at(i, int(d)) : 0.0;
    }
};

// ---- kernels ------------------------------------------------------------------------------------------
// Chunk p owns rows [p*L, min((p+1)*L, N)); its interior is the chunk minus the last b rows (the last chunk
// has no separator).  Lf(k, d, p) multiplies z_{k-d-1}; everything chunk-interleaved: index (k*b + d)*P + p.
//
// One thread per chunk.  The recurrence is sequential in k, so the only latency that may sit on the critical
// path is the FMA chain itself: the last B unknowns live in registers (never re-read from memory), and the
// factor entries / right-hand sides of the next U rows are loaded as one batch before they are needed.
constexpr int kChunkThreads = 64;  // one wavefront per workgroup: the chunks spread over all CUs, 512 VGPRs per lane
template <int B, int U>
__global__ __launch_bounds__(kChunkThreads) void k_chunk_solve(int64_t N, int b, int64_t L, int64_t P, const double* __restrict__ Lf,
                                                           const double* __restrict__ Dinv, const double* __restrict__ f,
                                                           double* __restrict__ y)
{
    const int64_t p = int64_t(blockIdx.x) * kChunkThreads + threadIdx.x;
    if (p >= P)
        return;
    const int64_t row0 = p * L;
    const int64_t m = ((p == P - 1) ? N : (row0 + L - b)) - row0;  // interior rows
    double hist[B];                                                // hist[d] = unknown k-d-1 (forward) / k+d+1 (backward)
#pragma unroll
    for (int d = 0; d < B; d++)
        hist[d] = 0.0;
    // forward: z_k = f_k - sum_d Lf(k,d) z_{k-d-1}
    for (int64_t k0 = 0; k0 < m; k0 += U)
    {
        double fk[U], lf[U][B];
#pragma unroll
        for (int u = 0; u < U; u++)
        {
            const int64_t k = (k0 + u < m) ? k0 + u : m - 1;
            fk[u] = f[row0 + k];
#pragma unroll
            for (int d = 0; d < B; d++)
                lf[u][d] = (d < b) ? Lf[(k * b + d) * P + p] : 0.0;  // rows k < b hold zeros for the missing neighbours
        }
#pragma unroll
        for (int u = 0; u < U; u++)
        {
            if (k0 + u < m)
            {
                double acc = fk[u];
#pragma unroll
                for (int d = B - 1; d >= 0; d--)  // the most recent unknown (d = 0) enters last
                    acc -= lf[u][d] * hist[d];
#pragma unroll
                for (int d = B - 1; d > 0; d--)
                    hist[d] = hist[d - 1];
                hist[0] = acc;
                y[row0 + k0 + u] = acc;
            }
        }
    }
    // diagonal and backward: y_k = z_k / D_k - sum_d Lf(k+d+1, d) y_{k+d+1}
#pragma unroll
    for (int d = 0; d < B; d++)
        hist[d] = 0.0;
    for (int64_t k0 = m - 1; k0 >= 0; k0 -= U)
    {
        double zk[U], di[U], lf[U][B];
#pragma unroll
        for (int u = 0; u < U; u++)
        {
            const int64_t k = (k0 - u >= 0) ? k0 - u : 0;
            zk[u] = y[row0 + k];
            di[u] = Dinv[k * P + p];
#pragma unroll
            for (int d = 0; d < B; d++)
                lf[u][d] = (d < b && k + d + 1 < m) ? Lf[((k + d + 1) * b + d) * P + p] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; u++)
        {
            if (k0 - u >= 0)
            {
                double acc = zk[u] * di[u];
#pragma unroll
                for (int d = B - 1; d >= 0; d--)
                    acc -= lf[u][d] * hist[d];
#pragma unroll
                for (int d = B - 1; d > 0; d--)
                    hist[d] = hist[d - 1];
                hist[0] = acc;
                y[row0 + k0 - u] = acc;
            }
        }
    }
}

// separator rows: s = p*b + c  <->  global row (p+1)*L - b + c.  g[s] = f[r] - sum over interior neighbours M(r,j) y[j]
__global__ __launch_bounds__(kThreads) void k_sep_rhs(int64_t N, int b, int64_t L, int64_t P, const double* __restrict__ band,
                                                       const double* __restrict__ f, const double* __restrict__ y,
                                                       double* __restrict__ g)
{
    const int64_t s = int64_t(blockIdx.x) * kThreads + threadIdx.x;
    if (s >= (P - 1) * b)
        return;
    const int64_t p = s / b;
    const int64_t sep0 = (p + 1) * L - b, r = sep0 + (s % b);
    double acc = f[r];
    for (int d = 1; d <= b; d++)
    {
        const int64_t ju = r - d;  // above: interior of chunk p unless still inside this separator
        if (ju >= 0 && ju < sep0)
            acc -= band[r * (b + 1) + d] * y[ju];
        const int64_t jl = r + d;  // below: interior of chunk p+1 unless still inside this separator
        if (jl < N && jl >= sep0 + b)
            acc -= band[jl * (b + 1) + d] * y[jl];
    }
    g[s] = acc;
}

// x_I = y_I - W [x_S(p-1); x_S(p)], x_S copied into place.  W: N x 2b row-major (zero rows for separators)
__global__ __launch_bounds__(kThreads) void k_back_subst(int64_t N, int b, int64_t L, int64_t P, const double* __restrict__ W,
                                                          const double* __restrict__ y, const double* __restrict__ xs,
                                                          double* __restrict__ x)
{
    const int64_t r = int64_t(blockIdx.x) * kThreads + threadIdx.x;
    if (r >= N)
        return;
    int64_t p = r / L;
    if (p > P - 1)
        p = P - 1;
    const int64_t sep0 = (p + 1) * L - b;
    if (p < P - 1 && r >= sep0)
    {
        x[r] = xs[p * b + (r - sep0)];
        return;
    }
    double acc = y[r];
    const double* w = W + r * (2 * b);
    if (p > 0)
        for (int c = 0; c < b; c++)
            acc -= w[c] * xs[(p - 1) * b + c];
    if (p < P - 1)
        for (int c = 0; c < b; c++)
            acc -= w[b + c] * xs[p * b + c];
    x[r] = acc;
}

// dst = src with sigma subtracted from the diagonal entries (column 0 of the n x (b+1) row-major band)
// (pencil: dst = src - sigma * srcB entry by entry, the bands share one layout)
__global__ __launch_bounds__(kThreads) void k_band_shift(int64_t total, int bw, double sigma, const double* __restrict__ src,
                                                          const double* __restrict__ srcB, double* __restrict__ dst)
{
    for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < total; i += int64_t(gridDim.x) * kThreads)
        dst[i] = src[i] - sigma * (srcB ? srcB[i] : ((i % bw) == 0 ? 1.0 : 0.0));
}

// ---- factorisation of the top level on the device -------------------------------------------------------
// One lane per chunk, the three steps of the host routine below (factor_level) in the same order of operations:
//   1. banded LDL' of the chunk's interior block (the last B rows of L and D kept in registers),
//   2. the 2b spikes  W = M_II^{-1} M_IS  (forward/backward substitution with the factor just written),
//   3. the chunk's (2b x 2b) contribution  M_SI W  to the Schur complement of its two separators.
// band: N x (b+1) row-major, band[i*(b+1)+d] = M(i, i-d).  W (N x 2b row-major) and C (P x 2b x 2b) must be
// zero on entry.  *fail is set when a pivot vanishes.
template <int B>
__global__ __launch_bounds__(kChunkThreads) void k_chunk_factor(int64_t N, int b, int64_t L, int64_t P, double tiny,
                                                                 const double* __restrict__ band, double* __restrict__ Lf,
                                                                 double* __restrict__ Dinv, double* __restrict__ W,
                                                                 double* __restrict__ C, int* __restrict__ fail)
{
    const int64_t p = int64_t(blockIdx.x) * kChunkThreads + threadIdx.x;
    if (p >= P)
        return;
    const int64_t row0 = p * L;
    const int64_t m = ((p == P - 1) ? N : (row0 + L - b)) - row0;  // interior rows
    const int bw = b + 1;
    // ---- 1. LDL' ---------------------------------------------------------------------------------------
    {
        double Lw[B][B], Dw[B];  // Lw[i][d] = L(k-1-i, k-1-i-d-1), Dw[i] = D(k-1-i)
#pragma unroll
        for (int i = 0; i < B; i++)
        {
            Dw[i] = 1.0;
#pragma unroll
            for (int d = 0; d < B; d++)
                Lw[i][d] = 0.0;
        }
        for (int64_t k = 0; k < m; k++)
        {
            const int dk = int(k < b ? k : b);
            const double* mrow = band + (row0 + k) * bw;
            double lrow[B];
#pragma unroll
            for (int d = B - 1; d >= 0; d--)
            {
                lrow[d] = 0.0;
                if (d < dk)
                {
                    double v = mrow[d + 1];
#pragma unroll
                    for (int e = B - 1; e > d; e--)
                        if (e < dk)
                            v -= lrow[e] * Dw[e] * Lw[d][e - d - 1];
                    lrow[d] = v / Dw[d];
                }
            }
            double dv = mrow[0];
#pragma unroll
            for (int d = 0; d < B; d++)
                if (d < dk)
                    dv -= lrow[d] * lrow[d] * Dw[d];
            if (!(fabs(dv) > tiny))
            {
                *fail = 1;
                dv = 1.0;
            }
            Dinv[k * P + p] = 1.0 / dv;
#pragma unroll
            for (int d = 0; d < B; d++)
                if (d < b)
                    Lf[(k * b + d) * P + p] = lrow[d];
#pragma unroll
            for (int i = B - 1; i > 0; i--)
            {
                Dw[i] = Dw[i - 1];
#pragma unroll
                for (int d = 0; d < B; d++)
                    Lw[i][d] = Lw[i - 1][d];
            }
            Dw[0] = dv;
#pragma unroll
            for (int d = 0; d < B; d++)
                Lw[0][d] = lrow[d];
        }
    }
    if (P == 1)
        return;
    // ---- 2. spikes: side 0 couples to separator p-1 (rows row0-b..row0-1), side 1 to separator p --------
    const int w2 = 2 * b;
    for (int side = 0; side < 2; side++)
    {
        if ((side == 0 && p == 0) || (side == 1 && p == P - 1))
            continue;
        double hist[B][B];  // hist[c][d] = unknown k-d-1 (forward) / k+d+1 (backward) of right-hand side c
#pragma unroll
        for (int c = 0; c < B; c++)
#pragma unroll
            for (int d = 0; d < B; d++)
                hist[c][d] = 0.0;
        for (int64_t k = 0; k < m; k++)
        {
            double lf[B];
#pragma unroll
            for (int d = 0; d < B; d++)
                lf[d] = (d < b) ? Lf[(k * b + d) * P + p] : 0.0;
            const int dk = int(k < b ? k : b);
#pragma unroll
            for (int c = 0; c < B; c++)
            {
                if (c >= b)
                    continue;
                double acc = 0.0;
                if (side == 0)
                {
                    if (k <= c)
                        acc = band[(row0 + k) * bw + (k + b - c)];
                }
                else if (k >= m + c - b)
                    acc = band[(row0 + m + c) * bw + (m + c - k)];
#pragma unroll
                for (int d = B - 1; d >= 0; d--)
                    if (d < dk)
                        acc -= lf[d] * hist[c][d];
#pragma unroll
                for (int d = B - 1; d > 0; d--)
                    hist[c][d] = hist[c][d - 1];
                hist[c][0] = acc;
                W[(row0 + k) * w2 + side * b + c] = acc;
            }
        }
#pragma unroll
        for (int c = 0; c < B; c++)
#pragma unroll
            for (int d = 0; d < B; d++)
                hist[c][d] = 0.0;
        for (int64_t k = m - 1; k >= 0; k--)
        {
            const double di = Dinv[k * P + p];
            double lf[B];
#pragma unroll
            for (int d = 0; d < B; d++)
                lf[d] = (d < b && k + d + 1 < m) ? Lf[((k + d + 1) * b + d) * P + p] : 0.0;
#pragma unroll
            for (int c = 0; c < B; c++)
            {
                if (c >= b)
                    continue;
                double acc = W[(row0 + k) * w2 + side * b + c] * di;
#pragma unroll
                for (int d = B - 1; d >= 0; d--)
                    acc -= lf[d] * hist[c][d];
#pragma unroll
                for (int d = B - 1; d > 0; d--)
                    hist[c][d] = hist[c][d - 1];
                hist[c][0] = acc;
                W[(row0 + k) * w2 + side * b + c] = acc;
            }
        }
    }
    // ---- 3. C(s1, s2) = sum_k M(separator row s1, interior k) * spike_s2[k] --------------------------------
    double* Cp = C + p * int64_t(w2) * w2;
    for (int side1 = 0; side1 < 2; side1++)
    {
        if ((side1 == 0 && p == 0) || (side1 == 1 && p == P - 1))
            continue;
        for (int c1 = 0; c1 < b; c1++)
            for (int s2 = 0; s2 < w2; s2++)
            {
                const int side2 = s2 / b;
                if ((side2 == 0 && p == 0) || (side2 == 1 && p == P - 1))
                    continue;
                double acc = 0.0;
                if (side1 == 0)
                {
                    const int64_t kend = (b < m) ? b : m;
                    for (int64_t k = 0; k < kend; k++)
                        if (k <= c1)
                            acc += band[(row0 + k) * bw + (k + b - c1)] * W[(row0 + k) * w2 + s2];
                }
                else
                {
                    for (int64_t k = (m - b > 0 ? m - b : 0); k < m; k++)
                        if (k >= m + c1 - b)
                            acc += band[(row0 + m + c1) * bw + (m + c1 - k)] * W[(row0 + k) * w2 + s2];
                }
                Cp[(side1 * b + c1) * w2 + s2] = acc;
            }
    }
}

// column-major host inverse -> row-major device copy
void upload_row_major(const std::vector<double>& inv, int64_t n, DevBuf<double>& dst)
{
    std::vector<double> rm(inv.size());
    for (int64_t c = 0; c < n; c++)
        for (int64_t r = 0; r < n; r++)
            rm[size_t(r) * n + c] = inv[size_t(c) * n + r];
    dst.alloc(rm.size());
    MISPEC_HIP(hipMemcpy(dst.p, rm.data(), rm.size() * sizeof(double), hipMemcpyHostToDevice));
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// one level of the recursive factorisation
// ---------------------------------------------------------------------------------------------------
struct mispec::BandLevel
{
    int64_t N = 0, L = 0, P = 1;
    int b = 0;
    DevBuf<double> Lf, Dinv, W, band, y, g, xs;
    DevBuf<double> inv;  // last level only: explicit inverse (N x N), applied by a dense GEMV
    std::unique_ptr<BandLevel> next;
};

namespace {

constexpr int64_t kChunk = 128;         // rows per chunk
constexpr int64_t kSingleChunk = 2048;  // a level this small is not partitioned any further: its inverse is formed
                                        // explicitly (banded LDL' solves of the unit vectors) and applied as a GEMV

void throw_singular()
{
    throw Error(MISPEC_EINVAL, "SparseSymShiftSolve: factorization failed with the given shift");
}

// chunk length and chunk count of a level
void plan_level(int64_t N, int b, int64_t& L, int64_t& P)
{
    L = std::max<int64_t>(kChunk, 4 * int64_t(b));
    P = (N <= std::max<int64_t>(kSingleChunk, 8 * int64_t(b))) ? 1 : N / L;
    if (P < 2)
    {
        P = 1;
        L = N;
    }
}

// whether a level of this shape is factored by k_chunk_factor (MISPEC_SHIFT_FACTOR=host keeps everything on the host)
bool factored_on_device(int64_t N, int b)
{
    static const bool host_only = getenv("MISPEC_SHIFT_FACTOR") && std::string(getenv("MISPEC_SHIFT_FACTOR")) == "host";
    int64_t L, P;
    plan_level(N, b, L, P);
    return P > 1 && b <= 8 && !host_only;
}

// Factor the band matrix M (destroyed) into `lev`, recursively.
void factor_level(mispec_ctx* ctx, HostBand& M, BandLevel& lev)
{
    const int64_t N = M.n;
    const int b = M.b;
    MISPEC_REQUIRE(b <= 64, "internal: band wider than the chunk kernel supports");
    lev.N = N;
    lev.b = b;
    int64_t L, P;
    plan_level(N, b, L, P);
    lev.L = L;
    lev.P = P;
    const int64_t mmax = (P == 1) ? N : std::max<int64_t>(L - b, N - (P - 1) * L);  // longest interior
    double scale = 0.0;
    for (int64_t i = 0; i < N; i++)
        scale = std::max(scale, std::fabs(static_cast<const HostBand&>(M).at(i, 0)));
    const double tiny = scale * 1e-14 + 1e-300;

    const bool on_device = factored_on_device(N, b);
    MISPEC_REQUIRE(on_device || !M.view, "internal: a band view is only valid for a level factored on the device");
    const size_t lf_size = size_t(mmax) * std::max(b, 1) * P, dinv_size = size_t(mmax) * P, w_size = size_t(N) * 2 * std::max(b, 1);
    std::vector<double> Lf, Dinv, W;  // host images of the factor (host path only)
    if (!on_device)
    {
        Lf.assign(lf_size, 0.0);
        Dinv.assign(dinv_size, 0.0);
        W.assign(w_size, 0.0);
    }
    const int64_t nsep = (P - 1) * b;
    HostBand S;  // Schur complement of the separators
    S.n = nsep;
    S.b = (P > 1) ? std::min<int64_t>(2 * b - 1, std::max<int64_t>(nsep - 1, 0)) : 0;
    S.a.assign(size_t(std::max<int64_t>(nsep, 1)) * (S.b + 1), 0.0);

    // ---- the top level of a large matrix is factored on the device (one lane per chunk, k_chunk_factor); the
    // ---- Schur complement comes back as per-chunk blocks and is assembled here, in the order of the host loop
    if (on_device)
    {
        ctx->make_current();
        const int w2 = 2 * b;
        lev.band.alloc(size_t(N) * (b + 1));
        if (M.view_dev)
        {
            const int64_t total = N * (b + 1);
            hipLaunchKernelGGL(k_band_shift, dim3(unsigned(std::min<int64_t>((total + kThreads - 1) / kThreads, 4096))), dim3(kThreads),
                               0, ctx->stream, total, b + 1, M.shift, M.view_dev, M.viewB_dev, lev.band.p);
            MISPEC_HIP(hipGetLastError());
        }
        else
            MISPEC_HIP(hipMemcpyAsync(lev.band.p, M.a.data(), M.a.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        lev.Lf.alloc(lf_size);
        lev.Dinv.alloc(dinv_size);
        lev.W.alloc(w_size);
        DevBuf<double> Cdev;
        Cdev.alloc(size_t(P) * w2 * w2);
        DevBuf<int> fail;
        fail.alloc(1);
        MISPEC_HIP(hipMemsetAsync(lev.Lf.p, 0, lf_size * sizeof(double), ctx->stream));
        MISPEC_HIP(hipMemsetAsync(lev.Dinv.p, 0, dinv_size * sizeof(double), ctx->stream));
        MISPEC_HIP(hipMemsetAsync(lev.W.p, 0, w_size * sizeof(double), ctx->stream));
        MISPEC_HIP(hipMemsetAsync(Cdev.p, 0, Cdev.n * sizeof(double), ctx->stream));
        MISPEC_HIP(hipMemsetAsync(fail.p, 0, sizeof(int), ctx->stream));
        const dim3 grid(unsigned((P + kChunkThreads - 1) / kChunkThreads));
        if (b <= 4)
            hipLaunchKernelGGL((k_chunk_factor<4>), grid, dim3(kChunkThreads), 0, ctx->stream, N, b, L, P, tiny, lev.band.p,
                               lev.Lf.p, lev.Dinv.p, lev.W.p, Cdev.p, fail.p);
        else
            hipLaunchKernelGGL((k_chunk_factor<8>), grid, dim3(kChunkThreads), 0, ctx->stream, N, b, L, P, tiny, lev.band.p,
                               lev.Lf.p, lev.Dinv.p, lev.W.p, Cdev.p, fail.p);
        MISPEC_HIP(hipGetLastError());
        std::vector<double> Cc(Cdev.n);
        int failed = 0;
        MISPEC_HIP(hipMemcpyAsync(Cc.data(), Cdev.p, Cc.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        MISPEC_HIP(hipMemcpyAsync(&failed, fail.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        MISPEC_HIP(hipStreamSynchronize(ctx->stream));
        if (failed)
            throw_singular();
        for (int64_t p = 0; p < P; p++)
            for (int side1 = 0; side1 < 2; side1++)
            {
                if ((side1 == 0 && p == 0) || (side1 == 1 && p == P - 1))
                    continue;
                for (int c1 = 0; c1 < b; c1++)
                {
                    const int64_t s1 = (side1 == 0 ? (p - 1) : p) * b + c1;
                    for (int side2 = 0; side2 < 2; side2++)
                    {
                        if ((side2 == 0 && p == 0) || (side2 == 1 && p == P - 1))
                            continue;
                        for (int c2 = 0; c2 < b; c2++)
                        {
                            const int64_t s2 = (side2 == 0 ? (p - 1) : p) * b + c2;
                            if (s2 > s1)
                                continue;  // lower triangle only
                            const double acc = Cc[(size_t(p) * w2 + size_t(side1 * b + c1)) * w2 + size_t(side2 * b + c2)];
                            if (acc != 0.0)
                            {
                                MISPEC_REQUIRE(s1 - s2 <= S.b, "internal: Schur complement wider than expected");
                                S.at(s1, int(s1 - s2)) -= acc;
                            }
                        }
                    }
                }
            }
    }

    std::vector<double> D, Lc, rhs, sol;
    for (int64_t p = 0; p < (on_device ? 0 : P); p++)
    {
        const int64_t row0 = p * L;
        const int64_t m = ((p == P - 1) ? N : (row0 + L - b)) - row0;
        // ---- banded LDL' of the interior block (no pivoting) --------------------------------------
        D.assign(size_t(m), 0.0);
        Lc.assign(size_t(m) * std::max(b, 1), 0.0);  // Lc[k*b + d] = L(k, k-d-1)
        for (int64_t k = 0; k < m; k++)
        {
            const int dk = int(std::min<int64_t>(k, b));
            // row k of L: for j = k-dk .. k-1
            for (int d = dk - 1; d >= 0; d--)
            {
                const int64_t j = k - d - 1;
                double v = M.at(row0 + k, d + 1);
                // subtract sum_{t<j} L(k,t) D_t L(j,t), t within both bands
                const int64_t tlo = std::max<int64_t>(std::max<int64_t>(k - b, j - b), 0);
                for (int64_t t = tlo; t < j; t++)
                    v -= Lc[size_t(k) * b + (k - t - 1)] * D[size_t(t)] * Lc[size_t(j) * b + (j - t - 1)];
                Lc[size_t(k) * b + d] = v / D[size_t(j)];
            }
            double dv = M.at(row0 + k, 0);
            for (int d = 0; d < dk; d++)
                dv -= Lc[size_t(k) * b + d] * Lc[size_t(k) * b + d] * D[size_t(k - d - 1)];
            if (!(std::fabs(dv) > tiny))
                throw_singular();
            D[size_t(k)] = dv;
        }
        for (int64_t k = 0; k < m; k++)
        {
            Dinv[size_t(k) * P + p] = 1.0 / D[size_t(k)];
            for (int d = 0; d < b; d++)
                Lf[(size_t(k) * b + d) * P + p] = Lc[size_t(k) * b + d];
        }
        if (P == 1)
        {
            // last level: explicit inverse, column by column
            std::vector<double> inv(size_t(N) * N, 0.0), e(static_cast<size_t>(N));
            for (int64_t c = 0; c < N; c++)
            {
                std::fill(e.begin(), e.end(), 0.0);
                e[size_t(c)] = 1.0;
                for (int64_t k = c; k < m; k++)  // forward (zero above c)
                {
                    double acc = e[size_t(k)];
                    const int dk2 = int(std::min<int64_t>(k, b));
                    for (int d = dk2 - 1; d >= 0; d--)
                        acc -= Lc[size_t(k) * b + d] * e[size_t(k - d - 1)];
                    e[size_t(k)] = acc;
                }
                for (int64_t k = m - 1; k >= 0; k--)
                {
                    double acc = e[size_t(k)] / D[size_t(k)];
                    const int dk2 = int(std::min<int64_t>(m - 1 - k, b));
                    for (int d = dk2 - 1; d >= 0; d--)
                        acc -= Lc[size_t(k + d + 1) * b + d] * e[size_t(k + d + 1)];
                    e[size_t(k)] = acc;
                }
                std::copy(e.begin(), e.end(), inv.begin() + size_t(c) * N);
            }
            ctx->make_current();
            upload_row_major(inv, N, lev.inv);
            break;
        }
        // ---- spikes W = M_II^{-1} M_IS and their contribution to the Schur complement -------------------
        auto solve_block = [&](std::vector<double>& v) {
            for (int64_t k = 0; k < m; k++)
            {
                double acc = v[size_t(k)];
                const int dk = int(std::min<int64_t>(k, b));
                for (int d = dk - 1; d >= 0; d--)
                    acc -= Lc[size_t(k) * b + d] * v[size_t(k - d - 1)];
                v[size_t(k)] = acc;
            }
            for (int64_t k = m - 1; k >= 0; k--)
            {
                double acc = v[size_t(k)] / D[size_t(k)];
                const int dk = int(std::min<int64_t>(m - 1 - k, b));
                for (int d = dk - 1; d >= 0; d--)
                    acc -= Lc[size_t(k + d + 1) * b + d] * v[size_t(k + d + 1)];
                v[size_t(k)] = acc;
            }
        };
        // side 0: separator p-1 (rows row0-b .. row0-1); side 1: separator p (rows row0+m .. row0+m+b-1)
        std::vector<std::vector<double>> spike(size_t(2 * b));
        for (int side = 0; side < 2; side++)
        {
            if ((side == 0 && p == 0) || (side == 1 && p == P - 1))
                continue;
            for (int c = 0; c < b; c++)
            {
                const int64_t sr = (side == 0) ? (row0 - b + c) : (row0 + m + c);
                rhs.assign(size_t(m), 0.0);
                bool any = false;
                for (int64_t k = (side == 0 ? 0 : std::max<int64_t>(m - b, 0)); k < (side == 0 ? std::min<int64_t>(b, m) : m); k++)
                {
                    const double e = M.get(row0 + k, sr);
                    rhs[size_t(k)] = e;
                    any = any || (e != 0.0);
                }
                if (any)
                    solve_block(rhs);
                spike[size_t(side * b + c)] = rhs;
                for (int64_t k = 0; k < m; k++)
                    W[size_t(row0 + k) * 2 * b + side * b + c] = rhs[size_t(k)];
            }
        }
        // S(s1, s2) -= sum_k M(sep row s1, interior k) * spike_s2[k]
        for (int side1 = 0; side1 < 2; side1++)
        {
            if ((side1 == 0 && p == 0) || (side1 == 1 && p == P - 1))
                continue;
            for (int c1 = 0; c1 < b; c1++)
            {
                const int64_t sr1 = (side1 == 0) ? (row0 - b + c1) : (row0 + m + c1);
                const int64_t s1 = (side1 == 0 ? (p - 1) : p) * b + c1;
                for (int side2 = 0; side2 < 2; side2++)
                {
                    if ((side2 == 0 && p == 0) || (side2 == 1 && p == P - 1))
                        continue;
                    for (int c2 = 0; c2 < b; c2++)
                    {
                        const int64_t s2 = (side2 == 0 ? (p - 1) : p) * b + c2;
                        if (s2 > s1)
                            continue;  // lower triangle only
                        const std::vector<double>& sp = spike[size_t(side2 * b + c2)];
                        double acc = 0.0;
                        for (int64_t k = (side1 == 0 ? 0 : std::max<int64_t>(m - b, 0)); k < (side1 == 0 ? std::min<int64_t>(b, m) : m);
                             k++)
                            acc += M.get(row0 + k, sr1) * sp[size_t(k)];
                        if (acc != 0.0)
                        {
                            MISPEC_REQUIRE(s1 - s2 <= S.b, "internal: Schur complement wider than expected");
                            S.at(s1, int(s1 - s2)) -= acc;
                        }
                    }
                }
            }
        }
    }
    if (P > 1)
    {
        // + M_SS itself (entries inside one separator; different separators are more than b rows apart)
        for (int64_t p = 0; p < P - 1; p++)
            for (int c1 = 0; c1 < b; c1++)
                for (int c2 = 0; c2 <= c1; c2++)
                    S.at(p * b + c1, c1 - c2) += static_cast<const HostBand&>(M).at((p + 1) * L - b + c1, c1 - c2);
    }

    // ---- upload this level ---------------------------------------------------------------------------
    ctx->make_current();
    auto up = [&](DevBuf<double>& dst, const std::vector<double>& src) {
        dst.alloc(std::max<size_t>(src.size(), 1));
        if (!src.empty())
            MISPEC_HIP(hipMemcpyAsync(dst.p, src.data(), src.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    };
    if (!on_device)
    {
        up(lev.Lf, Lf);
        up(lev.Dinv, Dinv);
    }
    lev.y.alloc(size_t(N));
    if (P > 1)
    {
        if (!on_device)
        {
            up(lev.W, W);
            up(lev.band, M.a);
        }
        lev.g.alloc(size_t(nsep));
        lev.xs.alloc(size_t(nsep));
    }
    MISPEC_HIP(hipStreamSynchronize(ctx->stream));
    M.a.clear();
    M.a.shrink_to_fit();
    if (P > 1)
    {
        lev.next = std::make_unique<BandLevel>();
        factor_level(ctx, S, *lev.next);
    }
}

void launch_chunk_solve(const mispec_ctx& ctx, const BandLevel& lev, dim3 grid, const double* f, double* y)
{
#define MISPEC_CHUNK(B, U)                                                                                                  \
    hipLaunchKernelGGL((k_chunk_solve<B, U>), grid, dim3(kChunkThreads), 0, ctx.stream, lev.N, lev.b, lev.L, lev.P, lev.Lf.p, \
                       lev.Dinv.p, f, y)
    // U rows of factor entries are in flight per lane and batch ((B + 2) * U doubles): as deep as the register
    // file allows, because with one lane per chunk nothing else hides the load latency
    if (lev.b <= 4)
        MISPEC_CHUNK(4, 32);
    else if (lev.b <= 8)
        MISPEC_CHUNK(8, 16);
    else if (lev.b <= 16)
        MISPEC_CHUNK(16, 8);
    else
        MISPEC_CHUNK(64, 1);
#undef MISPEC_CHUNK
    MISPEC_HIP(hipGetLastError());
}

void solve_level(const mispec_ctx& ctx, const BandLevel& lev, const double* f, double* x)
{
    const auto blocks = [](int64_t n) { return dim3(unsigned((n + kThreads - 1) / kThreads)); };
    if (lev.P == 1)
    {
        if (lev.inv.p)
        {
            // explicit inverse, row-major: one wavefront per row (dense.hip) — n waves in flight instead of n/256
            // workgroups that each walk all the columns (the first version: 233 us at n = 1830, a third of a banded solve)
            launch_row_gemv(ctx, lev.inv.p, lev.N, lev.N, lev.N, f, x);
        }
        else
            launch_chunk_solve(ctx, lev, dim3(1), f, x);
        return;
    }
    launch_chunk_solve(ctx, lev, dim3(unsigned((lev.P + kChunkThreads - 1) / kChunkThreads)), f, lev.y.p);
    hipLaunchKernelGGL(k_sep_rhs, blocks((lev.P - 1) * lev.b), dim3(kThreads), 0, ctx.stream, lev.N, lev.b, lev.L, lev.P,
                       lev.band.p, f, lev.y.p, lev.g.p);
    MISPEC_HIP(hipGetLastError());
    solve_level(ctx, *lev.next, lev.g.p, lev.xs.p);
    hipLaunchKernelGGL(k_back_subst, blocks(lev.N), dim3(kThreads), 0, ctx.stream, lev.N, lev.b, lev.L, lev.P, lev.W.p, lev.y.p,
                       lev.xs.p, x);
    MISPEC_HIP(hipGetLastError());
}

// dense LU with partial pivoting -> explicit inverse (column-major), host
template <typename T>
void dense_inverse(int n, std::vector<T>& A, std::vector<T>& inv)
{
    std::vector<int> piv(static_cast<size_t>(n));
    auto a = [&](int i, int j) -> T& { return A[size_t(j) * n + i]; };
    for (int k = 0; k < n; k++)
    {
        int pr = k;
        double best = std::abs(a(k, k));
        for (int i = k + 1; i < n; i++)
            if (std::abs(a(i, k)) > best)
            {
                best = std::abs(a(i, k));
                pr = i;
            }
        if (!(best > 0.0))
            throw_singular();
        piv[size_t(k)] = pr;
        if (pr != k)
            for (int j = 0; j < n; j++)
                std::swap(a(k, j), a(pr, j));
        const T d = a(k, k);
        for (int i = k + 1; i < n; i++)
            a(i, k) /= d;
        for (int j = k + 1; j < n; j++)
        {
            const T akj = a(k, j);
            if (akj == T(0))
                continue;
            T* col = &A[size_t(j) * n];
            const T* lk = &A[size_t(k) * n];
            for (int i = k + 1; i < n; i++)
                col[i] -= lk[i] * akj;
        }
    }
    inv.assign(size_t(n) * n, T(0));
    std::vector<T> e(static_cast<size_t>(n));
    for (int c = 0; c < n; c++)
    {
        std::fill(e.begin(), e.end(), T(0));
        e[size_t(c)] = T(1);
        for (int k = 0; k < n; k++)
            std::swap(e[size_t(k)], e[size_t(piv[size_t(k)])]);
        for (int k = 0; k < n; k++)  // L y = P e
        {
            const T ek = e[size_t(k)];
            if (ek != T(0))
                for (int i = k + 1; i < n; i++)
                    e[size_t(i)] -= a(i, k) * ek;
        }
        for (int k = n - 1; k >= 0; k--)  // U x = y
        {
            e[size_t(k)] /= a(k, k);
            const T ek = e[size_t(k)];
            if (ek != T(0))
                for (int i = 0; i < k; i++)
                    e[size_t(i)] -= a(i, k) * ek;
        }
        std::copy(e.begin(), e.end(), inv.begin() + size_t(c) * n);
    }
}

}  // namespace

namespace mispec {

void launch_shiftsolve(const mispec_symshift& S, const double* x_dev, double* y_dev)
{
    if (!S.factored)
        throw Error(MISPEC_ELOGIC, "SparseSymShiftSolve: need to call set_shift() first");
    if (S.dense)
        launch_row_gemv(*S.ctx, S.inverse.p, S.n, S.n, S.n, x_dev, y_dev);
    else
        solve_level(*S.ctx, *S.top, x_dev, y_dev);
}

}  // namespace mispec

mispec_symshift::~mispec_symshift() {}

// =================================================================================================
// C ABI
// =================================================================================================
namespace {
struct TriangleInput
{
    const int32_t* outer;
    const int32_t* inner;
    const double* val;
    bool lower;
    bool row_major;
};

// calls fn(row >= col, value) for every entry of the selected triangle, like selfadjointView<Uplo>
template <typename Fn>
void for_each_entry(const TriangleInput& T, int64_t n, Fn&& fn)
{
    for (int64_t o = 0; o < n; o++)
        for (int32_t p = T.outer[o]; p < T.outer[o + 1]; p++)
        {
            const int64_t in = T.inner[p];
            MISPEC_REQUIRE(in >= 0 && in < n, "mispec_symshift_create: index out of range");
            const int64_t r = T.row_major ? o : in, c = T.row_major ? in : o;
            if (T.lower ? (r >= c) : (r <= c))
                fn(r >= c ? r : c, r >= c ? c : r, T.val[p]);
        }
}

int symshift_create_impl(mispec_ctx* ctx, int64_t n, const TriangleInput& A, const TriangleInput* B, mispec_symshift** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && out && A.outer && n >= 1, "mispec_symshift_create: bad argument");
        MISPEC_REQUIRE(ctx->comm.allgather == nullptr, "mispec_symshift_create: shift-and-invert operators cannot be row-sharded");
        auto S = std::make_unique<mispec_symshift>();
        S->ctx = ctx;
        S->n = n;
        S->pencil = (B != nullptr);
        // first the bandwidth of the triangle(s) ...
        int64_t countA = 0, countB = 0;
        for_each_entry(A, n, [&](int64_t r, int64_t c, double) {
            S->half_bandwidth = std::max<int64_t>(S->half_bandwidth, r - c);
            countA++;
        });
        if (B)
            for_each_entry(*B, n, [&](int64_t r, int64_t c, double) {
                S->half_bandwidth = std::max<int64_t>(S->half_bandwidth, r - c);
                countB++;
            });
        if (S->half_bandwidth <= kMaxBandwidth)
        {
            // ... then, for a band, the band itself (assembled once; every set_shift() starts from it)
            S->band_b = int(std::max<int64_t>(1, std::min<int64_t>(S->half_bandwidth, n - 1)));  // a diagonal matrix: width 1, zeros
            const size_t bw = size_t(S->band_b) + 1;
            S->band0.assign(size_t(n) * bw, 0.0);
            for_each_entry(A, n, [&](int64_t r, int64_t c, double v) { S->band0[size_t(r) * bw + size_t(r - c)] += v; });
            if (B)
            {
                S->bandB0.assign(size_t(n) * bw, 0.0);
                for_each_entry(*B, n, [&](int64_t r, int64_t c, double v) { S->bandB0[size_t(r) * bw + size_t(r - c)] += v; });
            }
            if (factored_on_device(n, S->band_b))
            {
                ctx->make_current();
                S->band0_dev.alloc(S->band0.size());
                MISPEC_HIP(hipMemcpy(S->band0_dev.p, S->band0.data(), S->band0.size() * sizeof(double), hipMemcpyHostToDevice));
                if (B)
                {
                    S->bandB0_dev.alloc(S->bandB0.size());
                    MISPEC_HIP(hipMemcpy(S->bandB0_dev.p, S->bandB0.data(), S->bandB0.size() * sizeof(double), hipMemcpyHostToDevice));
                }
            }
        }
        else
        {
            // ... or (row >= col) triplets for the dense path
            S->rows.reserve(size_t(countA));
            S->cols.reserve(size_t(countA));
            S->vals.reserve(size_t(countA));
            for_each_entry(A, n, [&](int64_t r, int64_t c, double v) {
                S->rows.push_back(r);
                S->cols.push_back(c);
                S->vals.push_back(v);
            });
            if (B)
                for_each_entry(*B, n, [&](int64_t r, int64_t c, double v) {
                    S->rowsB.push_back(r);
                    S->colsB.push_back(c);
                    S->valsB.push_back(v);
                });
        }
        *out = S.release();
    });
}

bool parse_uplo(char uplo, bool& lower)
{
    lower = (uplo == 'L' || uplo == 'l');
    return lower || uplo == 'U' || uplo == 'u';
}
}  // namespace

extern "C" int mispec_symshift_create(mispec_ctx* ctx, int64_t n, const int32_t* outer, const int32_t* inner, const double* val,
                                      char uplo, int row_major, mispec_symshift** out)
{
    bool lower;
    if (!parse_uplo(uplo, lower))
    {
        set_last_error("mispec_symshift_create: uplo must be 'L' or 'U'");
        return MISPEC_EINVAL;
    }
    return symshift_create_impl(ctx, n, TriangleInput{outer, inner, val, lower, row_major != 0}, nullptr, out);
}

// SparseGenRealShiftSolve (MatOp/SparseGenRealShiftSolve.h:33-99): y = (A - sigma I)^{-1} x for a general sparse A.
// The reference factors with Eigen::SparseLU; here the dense path is used (LU with partial pivoting, explicit
// inverse, GEMV), so n <= 4096.
extern "C" int mispec_symshift_create_general(mispec_ctx* ctx, int64_t n, const int32_t* outer, const int32_t* inner, const double* val,
                                              int row_major, mispec_symshift** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && out && outer && n >= 1, "mispec_symshift_create_general: bad argument");
        MISPEC_REQUIRE(n <= kMaxDense,
                       "SparseGenRealShiftSolve: only n <= 4096 is supported on the GPU (the reference uses a general sparse LU)");
        MISPEC_REQUIRE(ctx->comm.allgather == nullptr, "mispec_symshift_create_general: shift-and-invert operators cannot be row-sharded");
        auto S = std::make_unique<mispec_symshift>();
        S->ctx = ctx;
        S->n = n;
        S->general = true;
        S->half_bandwidth = n;  // never the banded (symmetric) path
        for (int64_t o = 0; o < n; o++)
            for (int32_t p = outer[o]; p < outer[o + 1]; p++)
            {
                const int64_t in = inner[p];
                MISPEC_REQUIRE(in >= 0 && in < n, "mispec_symshift_create_general: index out of range");
                S->rows.push_back(row_major ? o : in);
                S->cols.push_back(row_major ? in : o);
                S->vals.push_back(val[p]);
            }
        *out = S.release();
    });
}

extern "C" int mispec_symshift_create_pencil(mispec_ctx* ctx, int64_t n, const int32_t* a_outer, const int32_t* a_inner,
                                             const double* a_val, char a_uplo, int a_row_major, const int32_t* b_outer,
                                             const int32_t* b_inner, const double* b_val, char b_uplo, int b_row_major,
                                             mispec_symshift** out)
{
    bool la, lb;
    if (!parse_uplo(a_uplo, la) || !parse_uplo(b_uplo, lb) || !b_outer)
    {
        set_last_error("mispec_symshift_create_pencil: bad argument (uplo must be 'L' or 'U', B must be given)");
        return MISPEC_EINVAL;
    }
    const TriangleInput B{b_outer, b_inner, b_val, lb, b_row_major != 0};
    return symshift_create_impl(ctx, n, TriangleInput{a_outer, a_inner, a_val, la, a_row_major != 0}, &B, out);
}

extern "C" int mispec_symshift_destroy(mispec_symshift* S)
{
    return guarded([&] {
        if (S)
        {
            S->ctx->make_current();
            delete S;
        }
    });
}

extern "C" int64_t mispec_symshift_rows(const mispec_symshift* S) { return S ? S->n : 0; }

extern "C" int mispec_symshift_set_shift(mispec_symshift* S, double sigma)
{
    return guarded([&] {
        MISPEC_REQUIRE(S, "mispec_symshift_set_shift: NULL argument");
        S->ctx->make_current();
        S->factored = false;
        S->sigma = sigma;
        const int64_t n = S->n, b = S->half_bandwidth;
        if (!S->general && b <= kMaxBandwidth)
        {
            HostBand M;
            M.n = n;
            M.b = S->band_b;
            if (S->band0_dev.p && factored_on_device(n, M.b))
            {
                M.view = &S->band0;  // A - sigma I (or A - sigma B) is formed on the device from the resident band(s)
                M.view_dev = S->band0_dev.p;
                if (S->pencil)
                {
                    M.viewB = &S->bandB0;
                    M.viewB_dev = S->bandB0_dev.p;
                }
                M.shift = sigma;
            }
            else
            {
                M.a = S->band0;
                if (S->pencil)
                    for (size_t e = 0; e < M.a.size(); e++)
                        M.a[e] -= sigma * S->bandB0[e];
                else
                    for (int64_t i = 0; i < n; i++)
                        M.at(i, 0) -= sigma;
            }
            S->top = std::make_unique<BandLevel>();
            factor_level(S->ctx, M, *S->top);
            S->dense = false;
        }
        else if (n <= kMaxDense)
        {
            std::vector<double> A(size_t(n) * n, 0.0), inv;
            for (size_t e = 0; e < S->vals.size(); e++)
            {
                A[size_t(S->cols[e]) * n + S->rows[e]] += S->vals[e];
                if (!S->general && S->rows[e] != S->cols[e])  // symmetric operators keep one triangle: mirror it
                    A[size_t(S->rows[e]) * n + S->cols[e]] += S->vals[e];
            }
            if (S->pencil)
                for (size_t e = 0; e < S->valsB.size(); e++)
                {
                    A[size_t(S->colsB[e]) * n + S->rowsB[e]] -= sigma * S->valsB[e];
                    if (S->rowsB[e] != S->colsB[e])
                        A[size_t(S->rowsB[e]) * n + S->colsB[e]] -= sigma * S->valsB[e];
                }
            else
                for (int64_t i = 0; i < n; i++)
                    A[size_t(i) * n + i] -= sigma;
            dense_inverse<double>(int(n), A, inv);
            upload_row_major(inv, n, S->inverse);
            S->dense = true;
        }
        else
            throw Error(MISPEC_EINVAL,
                        "SparseSymShiftSolve: only banded matrices (half-bandwidth <= 8) or n <= 4096 are supported on the GPU "
                        "(the reference uses a general sparse LU)");
        S->factored = true;
    });
}

// SparseGenComplexShiftSolve::set_shift(sigmar, sigmai) (MatOp/SparseGenComplexShiftSolve.h:74-99): the operator becomes
// y = Re((A - sigma I)^{-1} x) for real x.  Dense path only: complex LU with partial pivoting on the host, the REAL PART of
// the explicit inverse goes to HBM and is applied by the same GEMV kernel (Re(M x) = Re(M) x for real x).
extern "C" int mispec_symshift_set_shift_complex(mispec_symshift* S, double sigmar, double sigmai)
{
    if (S && sigmai == 0.0)
        return mispec_symshift_set_shift(S, sigmar);
    return guarded([&] {
        MISPEC_REQUIRE(S, "mispec_symshift_set_shift_complex: NULL argument");
        MISPEC_REQUIRE(S->general && !S->pencil, "mispec_symshift_set_shift_complex: complex shifts are for the general (non-symmetric) operator");
        S->ctx->make_current();
        S->factored = false;
        S->sigma = sigmar;
        const int64_t n = S->n;
        MISPEC_REQUIRE(n <= kMaxDense, "SparseGenComplexShiftSolve: only n <= 4096 is supported on the GPU");
        typedef std::complex<double> Cx;
        std::vector<Cx> A(size_t(n) * n, Cx(0.0, 0.0)), inv;
        for (size_t e = 0; e < S->vals.size(); e++)
            A[size_t(S->cols[e]) * n + S->rows[e]] += S->vals[e];
        for (int64_t i = 0; i < n; i++)
            A[size_t(i) * n + i] -= Cx(sigmar, sigmai);
        dense_inverse<Cx>(int(n), A, inv);
        std::vector<double> re(inv.size());
        for (size_t e = 0; e < inv.size(); e++)
            re[e] = inv[e].real();
        upload_row_major(re, n, S->inverse);
        S->dense = true;
        S->factored = true;
    });
}

extern "C" int mispec_symshift_solve(const mispec_symshift* S, const double* x_dev, double* y_dev)
{
    return guarded([&] {
        MISPEC_REQUIRE(S && x_dev && y_dev, "mispec_symshift_solve: NULL argument");
        S->ctx->make_current();
        launch_shiftsolve(*S, x_dev, y_dev);
    });
}

extern "C" int mispec_symshift_solve_host(const mispec_symshift* S, const double* x_host, double* y_host)
{
    return guarded([&] {
        MISPEC_REQUIRE(S && x_host && y_host, "mispec_symshift_solve_host: NULL argument");
        S->ctx->make_current();
        if (S->stage_x.n < size_t(S->n))
        {
            S->stage_x.alloc(size_t(S->n));
            S->stage_y.alloc(size_t(S->n));
        }
        hipStream_t s = S->ctx->stream;
        MISPEC_HIP(hipMemcpyAsync(S->stage_x.p, x_host, size_t(S->n) * sizeof(double), hipMemcpyHostToDevice, s));
        launch_shiftsolve(*S, S->stage_x.p, S->stage_y.p);
        MISPEC_HIP(hipMemcpyAsync(y_host, S->stage_y.p, size_t(S->n) * sizeof(double), hipMemcpyDeviceToHost, s));
        MISPEC_HIP(hipStreamSynchronize(s));
    });
}
