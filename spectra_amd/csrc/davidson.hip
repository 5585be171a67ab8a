// Block Davidson solver with the diagonal (DPR) correction on the GPU — replaces DavidsonSymEigsSolver.h:18-90 on top
// of JDSymEigsBase.h:28-187, LinAlg/SearchSpace.h:24-93, LinAlg/RitzPairs.h:24-127 and LinAlg/Orthogonalization.h.
//
// What stays in HBM: the search space V (n x cap) and its image AV = A V, the Ritz vectors / A-images of the wanted
// pairs, the diagonal of A.  What the host does: the dense symmetric eigenproblem of the projected matrix
// G = V' A V (at most 128 x 128: Householder tridiagonalisation + the TridiagEigen restatement), sorting and the
// convergence decision.  Every length-n operation reuses the factorisation kernels: G's new columns are V'(A v)
// (k_orth VTF), Ritz vectors and restarts are V Y (k_vq), new directions are orthogonalised against the whole space by
// two Gram-Schmidt passes per column (k_orth CORRECT_VTF / CORRECT_ONLY) — the reference projects the block and
// then takes a Householder QR of it, twice (Orthogonalization.h:107-137); both produce an orthonormal basis of the same
// space, so Ritz values agree to rounding and Ritz vectors up to sign.
#include <algorithm>
#include <cmath>
#include <memory>
#include <numeric>
#include <vector>

#include <Spectra/Util/SelectionRule.h>
#include <Spectra/internal/SmallDense.h>

#include "csr.hpp"
#include "dense.hpp"
#include "krylov.hpp"

using namespace mispec;

namespace {
constexpr int kDavidsonMaxCols = 256;  // search-space columns held on the device (the projected problem is solved densely on the host)

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void k_csr_diag(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                                                        const double* __restrict__ val, int64_t row_begin, int64_t nloc,
                                                        double* __restrict__ diag)
{
    const int64_t r = int64_t(blockIdx.x) * kThreads + threadIdx.x;
    if (r >= nloc)
        return;
    double d = 0.0;
    for (int p = rowptr[r]; p < rowptr[r + 1]; p++)
        if (colind[p] == row_begin + r)
            d += val[p];
    diag[r] = d;
}

__global__ __launch_bounds__(kThreads) void k_dense_diag(const double* __restrict__ a, int64_t ld, int64_t n, double* __restrict__ diag)
{
    const int64_t r = int64_t(blockIdx.x) * kThreads + threadIdx.x;
    if (r < n)
        diag[r] = a[r * ld + r];
}

// V[idx[k], k] = 1 for k < count (the initial search space: unit vectors, DavidsonSymEigsSolver.h:47-58)
__global__ void k_set_units(double* __restrict__ V, int64_t ldv, const int64_t* __restrict__ idx, int count)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < count)
        V[idx[k] + int64_t(k) * ldv] = 1.0;
}

// t = (y - theta x) / (theta - d): the residual of the Ritz pair (RitzPairs.h:122) divided by the diagonal
// preconditioner (DavidsonSymEigsSolver.h:69-74)
__global__ __launch_bounds__(kThreads) void k_dpr_correction(const double* __restrict__ y, const double* __restrict__ x, double theta,
                                                              const double* __restrict__ diag, int64_t n, double* __restrict__ t)
{
    const int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x;
    if (i < n)
        t[i] = (y[i] - theta * x[i]) / (theta - diag[i]);
}

dim3 blocks_for(int64_t n) { return dim3(unsigned((n + kThreads - 1) / kThreads)); }

// All eigenpairs of the symmetric s x s matrix G (column-major, destroyed): evals[s], evecs s x s column-major.
// Householder reduction to tridiagonal form with the transformation accumulated, then the implicit-shift QR of
// internal/SmallDense.h applied to that accumulated matrix.  Returns false if the QR iteration did not converge.
bool symmetric_eigen(int s, std::vector<double>& G, std::vector<double>& evals, std::vector<double>& evecs)
{
    auto g = [&](int i, int j) -> double& { return G[size_t(j) * s + i]; };
    evecs.assign(size_t(s) * s, 0.0);
    auto q = [&](int i, int j) -> double& { return evecs[size_t(j) * s + i]; };
    for (int i = 0; i < s; i++)
        q(i, i) = 1.0;
    std::vector<double> v(static_cast<size_t>(s)), w(static_cast<size_t>(s));
    for (int k = 0; k + 2 < s; k++)
    {
        double norm2 = 0.0;
        for (int i = k + 1; i < s; i++)
            norm2 += g(i, k) * g(i, k);
        const double x0 = g(k + 1, k);
        if (!(norm2 - x0 * x0 > 0.0))
            continue;  // already tridiagonal in this column
        const double alpha = x0 > 0.0 ? -std::sqrt(norm2) : std::sqrt(norm2);
        std::fill(v.begin(), v.end(), 0.0);
        for (int i = k + 1; i < s; i++)
            v[size_t(i)] = g(i, k);
        v[size_t(k) + 1] -= alpha;
        double vn = 0.0;
        for (int i = k + 1; i < s; i++)
            vn += v[size_t(i)] * v[size_t(i)];
        vn = std::sqrt(vn);
        if (!(vn > 0.0))
            continue;
        for (int i = k + 1; i < s; i++)
            v[size_t(i)] /= vn;
        // G <- H G H with H = I - 2 v v':  w = G v ; K = v'w ; G -= 2 (v w' + w v') - 4 K v v'
        for (int i = 0; i < s; i++)
        {
            double acc = 0.0;
            for (int j = k + 1; j < s; j++)
                acc += g(i, j) * v[size_t(j)];
            w[size_t(i)] = acc;
        }
        double K = 0.0;
        for (int i = k + 1; i < s; i++)
            K += v[size_t(i)] * w[size_t(i)];
        for (int j = 0; j < s; j++)
            for (int i = 0; i < s; i++)
                g(i, j) -= 2.0 * (v[size_t(i)] * w[size_t(j)] + w[size_t(i)] * v[size_t(j)]) - 4.0 * K * v[size_t(i)] * v[size_t(j)];
        // Q <- Q H
        for (int i = 0; i < s; i++)
        {
            double acc = 0.0;
            for (int j = k + 1; j < s; j++)
                acc += q(i, j) * v[size_t(j)];
            for (int j = k + 1; j < s; j++)
                q(i, j) -= 2.0 * acc * v[size_t(j)];
        }
    }
    evals.resize(static_cast<size_t>(s));
    std::vector<double> subd(static_cast<size_t>(std::max(s - 1, 1)), 0.0);
    for (int i = 0; i < s; i++)
        evals[size_t(i)] = g(i, i);
    for (int i = 0; i + 1 < s; i++)
        subd[size_t(i)] = 0.5 * (g(i + 1, i) + g(i, i + 1));
    return small::tridiag_eigen(s, evals.data(), subd.data(), evecs.data(), s, small::Lanes{0, 1}) == 0;
}

}  // namespace

struct mispec_davidson
{
    mispec_ctx* ctx = nullptr;
    const mispec_csr* A = nullptr;
    const mispec_dense* D = nullptr;
    mispec_device_op_fn dop = nullptr;
    void* dop_user = nullptr;
    int64_t n = 0, ldv = 0;
    int nev = 0, init_size = 0, max_size = 0, corr_size = 0;  // JDSymEigsBase.h:37-42
    int cap = 0;                                              // columns allocated for V / AV
    DevBuf<double> V, AV, X, AX, W, diag, partials, red, Ydev;
    DevBuf<int64_t> idx;
    PinnedBuf<double> h_red;
    int64_t pstride = 0;
    std::vector<double> diag_host, theta, Y, G;
    std::vector<char> converged;
    int size = 0;    // current search-space dimension
    int niter = 0, info = 1 /* CompInfo::NotComputed */;
    int64_t nops = 0;
    bool computed = false;

    hipStream_t stream() const { return ctx->stream; }
    void apply(const double* x, double* y)
    {
        if (A)
            launch_spmv(*A, x, y, nullptr);
        else if (D)
            launch_row_gemv(*ctx, D->a.p, D->ld, D->rows, D->cols, x, y, true);
        else if (dop(dop_user, x, y, static_cast<void*>(stream())) != 0)
            throw Error(MISPEC_ERUNTIME, "user device operator callback reported failure");
        nops++;
    }
    double* vcol(int j) { return V.p + int64_t(j) * ldv; }
    double* avcol(int j) { return AV.p + int64_t(j) * ldv; }

    OrthArgs orth(int ncol) const
    {
        OrthArgs a;
        a.V = V.p;
        a.ldv = ldv;
        a.ncol = ncol;
        a.n = n;
        a.partials = partials.p;
        a.pstride = pstride;
        return a;
    }
    // sum the records of the last launch into red (device) and bring them to h_red
    void reduce_host(int nrec, int ncol)
    {
        FinishArgs fin;
        launch_reduce_partials(*ctx, partials.p, pstride, nrec, ncol, red.p, fin);
        MISPEC_HIP(hipMemcpyAsync(h_red.p, red.p, kPartialLd * sizeof(double), hipMemcpyDeviceToHost, stream()));
        MISPEC_HIP(hipStreamSynchronize(stream()));
    }
    void reduce_device(int nrec, int ncol, double* dst)
    {
        FinishArgs fin;
        launch_reduce_partials(*ctx, partials.p, pstride, nrec, ncol, dst, fin);
    }
};

namespace {

// Orthonormalise column `j` of V against columns [0, j): two Gram-Schmidt passes ("twice is enough",
// Orthogonalization.h:130-137), then normalise.  Returns false if the column vanished (it is then left out).
bool orthonormalise_column(mispec_davidson& S, int j)
{
    double* t = S.vcol(j);
    double* c1 = S.red.p;               // coefficients of the first pass (device)
    double* c2 = S.red.p + kPartialLd;  // ... and of the second
    if (j > 0)
    {
        OrthArgs a = S.orth(j);
        a.src = t;
        S.reduce_device(launch_orth(*S.ctx, ORTH_VTF, a), j, c1);  // c = V't
        OrthArgs b = S.orth(j);
        b.src = t;
        b.dst = t;
        b.c_in = c1;
        S.reduce_device(launch_orth(*S.ctx, ORTH_CORRECT_VTF, b), j, c2);  // t -= V c ; c' = V't
        OrthArgs c = S.orth(j);
        c.src = t;
        c.dst = t;
        c.c_in = c2;
        const int nrec = launch_orth(*S.ctx, ORTH_CORRECT_ONLY, c);  // t -= V c' ; |t|^2
        S.reduce_host(nrec, 1);
    }
    else  // first column: only its norm (|t - 0 * t|^2 lands in the norm slot)
        S.reduce_host(launch_resid_norms(*S.ctx, t, t, 0.0, S.n, S.partials.p, S.pstride), 1);
    const double nrm = std::sqrt(S.h_red.p[kSlotBeta2]);
    if (!(nrm > 0.0) || !std::isfinite(nrm))
        return false;
    launch_scale(*S.ctx, t, t, S.ldv, nrm);
    return true;
}

void check_sizes(mispec_davidson& S)
{
    MISPEC_REQUIRE(S.init_size >= 1 && S.corr_size >= 1 && S.max_size >= S.init_size,
                   "DavidsonSymEigsSolver: need 1 <= initial search space <= maximum search space and a positive correction size");
    // The device search space holds 128 vectors.  A larger maximum (the reference's default is 10 * nev) is lowered to what
    // fits — the solver then restarts earlier, which changes the iteration count, not the result; a space that cannot even
    // hold the initial vectors plus one correction block is an error.
    MISPEC_REQUIRE(S.init_size + S.corr_size <= kDavidsonMaxCols,
                   "DavidsonSymEigsSolver: the device search space holds at most 128 vectors (initial space + correction size <= 256)");
    if (S.max_size + S.corr_size > kDavidsonMaxCols)
        S.max_size = kDavidsonMaxCols - S.corr_size;
    MISPEC_REQUIRE(S.init_size >= S.nev && S.corr_size <= S.init_size,
                   "DavidsonSymEigsSolver: the initial search space must hold at least nev vectors and the correction block");
}

void allocate(mispec_davidson& S)
{
    const int cap = S.max_size + S.corr_size;
    if (cap <= S.cap)
        return;
    S.cap = cap;
    S.V.alloc(size_t(S.ldv) * cap);
    S.AV.alloc(size_t(S.ldv) * cap);
    S.X.alloc(size_t(S.ldv) * cap);
    S.AX.alloc(size_t(S.ldv) * cap);
    S.Ydev.alloc(size_t(cap) * cap);
    S.idx.alloc(size_t(cap));
}

std::unique_ptr<mispec_davidson> make_solver(mispec_ctx* ctx, int64_t n, int64_t nev, int64_t nvec_init, int64_t nvec_max)
{
    MISPEC_REQUIRE(ctx, "mispec_davidson_create: NULL context");
    MISPEC_REQUIRE(ctx->comm.allgather == nullptr, "DavidsonSymEigsSolver: the operator cannot be row-sharded");
    // JDSymEigsBase.h:49-53
    MISPEC_REQUIRE(nev >= 1 && nev <= n - 1, "nev must satisfy 1 <= nev <= n - 1, n is the size of matrix");
    ctx->make_current();
    auto S = std::make_unique<mispec_davidson>();
    S->ctx = ctx;
    S->n = n;
    S->ldv = round_up(std::max<int64_t>(n, 1), 2);
    S->nev = int(nev);
    // JDSymEigsBase.h:70-78 and initialize() :55-66
    int64_t max_size = nvec_max < n ? nvec_max : 10 * nev;
    int64_t init_size = nvec_init < n ? nvec_init : 2 * nev;
    int64_t corr = nev;
    if (n < max_size)
        max_size = n;
    if (n < init_size + corr)
    {
        init_size = n / 3;
        corr = n / 3;
    }
    S->max_size = int(std::min<int64_t>(max_size, INT32_MAX));
    S->init_size = int(std::min<int64_t>(init_size, INT32_MAX));
    S->corr_size = int(std::min<int64_t>(corr, INT32_MAX));
    S->diag.alloc(size_t(S->ldv));
    S->W.alloc(size_t(S->ldv));
    const int64_t max_rec = int64_t(ctx->num_cu) * 8 + 8;
    S->pstride = max_rec;
    S->partials.alloc(size_t(max_rec) * kPartialLd);
    MISPEC_HIP(hipMemsetAsync(S->partials.p, 0, S->partials.n * sizeof(double), ctx->stream));
    S->red.alloc(2 * kPartialLd);
    MISPEC_HIP(hipMemsetAsync(S->red.p, 0, S->red.n * sizeof(double), ctx->stream));
    S->h_red.alloc(kPartialLd + 8);
    S->diag_host.resize(size_t(n));
    return S;
}

void fetch_diag(mispec_davidson& S)
{
    MISPEC_HIP(hipMemcpyAsync(S.diag_host.data(), S.diag.p, size_t(S.n) * sizeof(double), hipMemcpyDeviceToHost, S.stream()));
    MISPEC_HIP(hipStreamSynchronize(S.stream()));
}

}  // namespace

extern "C" int mispec_davidson_create(mispec_ctx* ctx, const mispec_csr* A, int64_t nev, int64_t nvec_init, int64_t nvec_max,
                                      mispec_davidson** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(A && out, "mispec_davidson_create: NULL argument");
        MISPEC_REQUIRE(A->ctx == ctx && A->n_rows == A->n_cols, "mispec_davidson_create: needs a square matrix of this context");
        auto S = make_solver(ctx, A->n_rows, nev, nvec_init, nvec_max);
        S->A = A;
        hipLaunchKernelGGL(k_csr_diag, blocks_for(S->n), dim3(kThreads), 0, ctx->stream, A->rowptr.p, A->colind.p, A->val.p, A->row_begin,
                           S->n, S->diag.p);
        MISPEC_HIP(hipGetLastError());
        if (A->reordered())
        {
            // the stored matrix is P A P' (reorder.hip): its i-th diagonal entry is A(perm[i], perm[i]), while the operator of this
            // solver (launch_spmv) keeps the caller's order — the preconditioner and the unit start vectors must use that order too
            DevBuf<double> tmp;
            tmp.alloc(size_t(S->n) + 2);
            launch_from_stored_order(*A, S->diag.p, tmp.p);
            MISPEC_HIP(hipMemcpyAsync(S->diag.p, tmp.p, size_t(S->n) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            MISPEC_HIP(hipStreamSynchronize(ctx->stream));
        }
        fetch_diag(*S);
        *out = S.release();
    });
}

extern "C" int mispec_davidson_create_dense(mispec_ctx* ctx, const mispec_dense* D, int64_t nev, int64_t nvec_init, int64_t nvec_max,
                                            mispec_davidson** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(D && out, "mispec_davidson_create_dense: NULL argument");
        MISPEC_REQUIRE(D->ctx == ctx && D->rows == D->cols, "mispec_davidson_create_dense: needs a square matrix of this context");
        auto S = make_solver(ctx, D->rows, nev, nvec_init, nvec_max);
        S->D = D;
        hipLaunchKernelGGL(k_dense_diag, blocks_for(S->n), dim3(kThreads), 0, ctx->stream, D->a.p, D->ld, S->n, S->diag.p);
        MISPEC_HIP(hipGetLastError());
        fetch_diag(*S);
        *out = S.release();
    });
}

extern "C" int mispec_davidson_create_device_op(mispec_ctx* ctx, mispec_device_op_fn op, void* op_user, int64_t n, const double* diag_host,
                                                int64_t nev, int64_t nvec_init, int64_t nvec_max, mispec_davidson** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(op && diag_host && out && n >= 1, "mispec_davidson_create_device_op: bad argument");
        auto S = make_solver(ctx, n, nev, nvec_init, nvec_max);
        S->dop = op;
        S->dop_user = op_user;
        std::copy(diag_host, diag_host + n, S->diag_host.begin());
        MISPEC_HIP(hipMemcpy(S->diag.p, diag_host, size_t(n) * sizeof(double), hipMemcpyHostToDevice));
        *out = S.release();
    });
}

extern "C" int mispec_davidson_destroy(mispec_davidson* S)
{
    return guarded([&] {
        if (S)
        {
            S->ctx->make_current();
            delete S;
        }
    });
}

// JDSymEigsBase.h:86-105 (negative = keep)
extern "C" int mispec_davidson_set_sizes(mispec_davidson* S, int64_t initial_search_space, int64_t max_search_space, int64_t correction)
{
    return guarded([&] {
        MISPEC_REQUIRE(S, "mispec_davidson_set_sizes: NULL argument");
        if (initial_search_space >= 0)
            S->init_size = int(std::min<int64_t>(initial_search_space, INT32_MAX));
        if (max_search_space >= 0)
            S->max_size = int(std::min<int64_t>(max_search_space, INT32_MAX));
        if (correction >= 0)
            S->corr_size = int(std::min<int64_t>(correction, INT32_MAX));
    });
}

extern "C" int mispec_davidson_get_sizes(const mispec_davidson* S, int64_t* initial_search_space, int64_t* max_search_space,
                                         int64_t* correction)
{
    return guarded([&] {
        MISPEC_REQUIRE(S, "mispec_davidson_get_sizes: NULL argument");
        if (initial_search_space)
            *initial_search_space = S->init_size;
        if (max_search_space)
            *max_search_space = S->max_size;
        if (correction)
            *correction = S->corr_size;
    });
}

// compute() / compute_with_guess() — JDSymEigsBase.h:121-184.  guess_host == NULL: the unit-vector start of
// DavidsonSymEigsSolver.h:47-58; else an n x guess_cols column-major block (orthonormalised here: the search space must
// be orthonormal for G = V'AV to be the projected operator; the reference leaves that to the caller).
extern "C" int mispec_davidson_compute(mispec_davidson* Sp, int selection, int64_t maxit, double tol, const double* guess_host,
                                       int64_t guess_cols, int64_t ldg, int64_t* nconv)
{
    return guarded([&] {
        MISPEC_REQUIRE(Sp && nconv, "mispec_davidson_compute: NULL argument");
        mispec_davidson& S = *Sp;
        const Spectra::SortRule rule = static_cast<Spectra::SortRule>(selection);
        (void) Spectra::internal::sort_key(rule, 0.0);  // "unsupported selection rule" (SelectionRule.h) for the complex-only rules
        S.ctx->make_current();
        check_sizes(S);
        allocate(S);
        const int64_t n = S.n;
        hipStream_t st = S.stream();

        // ---- initial search space
        MISPEC_HIP(hipMemsetAsync(S.V.p, 0, S.V.n * sizeof(double), st));
        MISPEC_HIP(hipMemsetAsync(S.AV.p, 0, S.AV.n * sizeof(double), st));
        MISPEC_HIP(hipMemsetAsync(S.X.p, 0, S.X.n * sizeof(double), st));
        MISPEC_HIP(hipMemsetAsync(S.AX.p, 0, S.AX.n * sizeof(double), st));
        MISPEC_HIP(hipMemsetAsync(S.W.p, 0, S.W.n * sizeof(double), st));
        int size = 0;
        if (!guess_host)
        {
            const std::vector<std::ptrdiff_t> order = Spectra::argsort(rule, S.diag_host.data(), std::ptrdiff_t(n));
            std::vector<int64_t> idx(static_cast<size_t>(S.init_size));
            for (int k = 0; k < S.init_size; k++)
                idx[size_t(k)] = int64_t(order[size_t(k)]);
            MISPEC_HIP(hipMemcpyAsync(S.idx.p, idx.data(), idx.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_set_units, dim3(unsigned((S.init_size + 63) / 64)), dim3(64), 0, st, S.V.p, S.ldv, S.idx.p, S.init_size);
            MISPEC_HIP(hipGetLastError());
            MISPEC_HIP(hipStreamSynchronize(st));  // idx is a local
            size = S.init_size;
        }
        else
        {
            MISPEC_REQUIRE(guess_cols >= S.nev && guess_cols <= S.max_size && ldg >= n,
                           "compute_with_guess: the initial space needs nev <= columns <= maximum search space");
            for (int64_t j = 0; j < guess_cols; j++)
            {
                MISPEC_HIP(hipMemcpyAsync(S.vcol(size), guess_host + j * ldg, size_t(n) * sizeof(double), hipMemcpyHostToDevice, st));
                if (orthonormalise_column(S, size))
                    size++;
            }
            MISPEC_REQUIRE(size >= S.nev, "compute_with_guess: the initial space has fewer than nev independent columns");
        }

        int nprod = 0;   // columns of AV that are up to date
        int gvalid = 0;  // leading block of G that is up to date
        S.G.assign(size_t(S.cap) * S.cap, 0.0);
        auto Gat = [&](int i, int j) -> double& { return S.G[size_t(j) * S.cap + i]; };
        const int q = std::max(S.nev, S.corr_size);  // Ritz pairs that are formed explicitly
        S.info = 1;  // NotComputed
        S.niter = 0;
        S.nops = 0;
        S.converged.assign(size_t(S.nev), 0);
        bool have_pairs = false;
        std::vector<double> Gwork, evals, evecs;
        for (S.niter = 0; S.niter < maxit; S.niter++)
        {
            // ---- restart (SearchSpace.h:59-63): V <- V Y[:, :init], AV <- AV Y[:, :init]
            if (size > S.max_size)
            {
                // The Ritz pairs are those of the space BEFORE the last extension (the reference restarts from
                // ritz_pairs of the previous iteration, so the block that was just appended is dropped).
                MISPEC_REQUIRE(have_pairs, "DavidsonSymEigsSolver: restart before the first Ritz pairs");
                const int rows = int(S.theta.size());
                const int keep = std::min(S.init_size, rows);
                MISPEC_HIP(hipMemcpyAsync(S.Ydev.p, S.Y.data(), size_t(rows) * keep * sizeof(double), hipMemcpyHostToDevice, st));
                launch_vq(*S.ctx, S.V.p, S.ldv, rows, S.Ydev.p, rows, keep, S.X.p, S.ldv, n);
                launch_vq(*S.ctx, S.AV.p, S.ldv, rows, S.Ydev.p, rows, keep, S.AX.p, S.ldv, n);
                MISPEC_HIP(hipMemcpyAsync(S.V.p, S.X.p, size_t(S.ldv) * keep * sizeof(double), hipMemcpyDeviceToDevice, st));
                MISPEC_HIP(hipMemcpyAsync(S.AV.p, S.AX.p, size_t(S.ldv) * keep * sizeof(double), hipMemcpyDeviceToDevice, st));
                MISPEC_HIP(hipStreamSynchronize(st));  // Y is about to be overwritten
                size = keep;
                nprod = keep;
                gvalid = 0;
            }
            // ---- AV for the new columns (SearchSpace.h:51-57)
            for (int j = nprod; j < size; j++)
                S.apply(S.vcol(j), S.avcol(j));
            nprod = size;
            // ---- projected matrix G = V' AV (RitzPairs.h:113): new columns by V'(A v_j), the rest by symmetry
            for (int j = gvalid; j < size; j++)
            {
                OrthArgs a = S.orth(size);
                a.src = S.avcol(j);
                S.reduce_host(launch_orth(*S.ctx, ORTH_VTF, a), size);
                for (int i = 0; i < size; i++)
                    Gat(i, j) = S.h_red.p[i];
            }
            for (int j = gvalid; j < size; j++)
            {
                for (int i = 0; i < gvalid; i++)
                    Gat(j, i) = Gat(i, j);  // new rows of the old columns
                for (int i = gvalid; i < j; i++)
                {
                    const double sym = 0.5 * (Gat(i, j) + Gat(j, i));  // both were measured: keep G exactly symmetric
                    Gat(i, j) = sym;
                    Gat(j, i) = sym;
                }
            }
            gvalid = size;
            // ---- small eigenproblem (RitzPairs.h:115-117), sorted by the selection rule (:41-52)
            Gwork.resize(size_t(size) * size);
            for (int j = 0; j < size; j++)
                for (int i = 0; i < size; i++)
                    Gwork[size_t(j) * size + i] = Gat(i, j);
            if (!symmetric_eigen(size, Gwork, evals, evecs))
            {
                S.info = 3;  // CompInfo::NumericalIssue (JDSymEigsBase.h:160-164)
                break;
            }
            const std::vector<std::ptrdiff_t> order = Spectra::argsort(rule, evals.data(), std::ptrdiff_t(size));
            S.theta.resize(size_t(size));
            S.Y.resize(size_t(size) * size);
            for (int k = 0; k < size; k++)
            {
                S.theta[size_t(k)] = evals[size_t(order[size_t(k)])];
                std::copy(evecs.begin() + size_t(order[size_t(k)]) * size, evecs.begin() + size_t(order[size_t(k)] + 1) * size,
                          S.Y.begin() + size_t(k) * size);
            }
            have_pairs = true;
            // ---- Ritz vectors X = V Y and their images A X = AV Y for the first q pairs (RitzPairs.h:119-122)
            const int qq = std::min(q, size);
            MISPEC_HIP(hipMemcpyAsync(S.Ydev.p, S.Y.data(), size_t(size) * qq * sizeof(double), hipMemcpyHostToDevice, st));
            launch_vq(*S.ctx, S.V.p, S.ldv, size, S.Ydev.p, size, qq, S.X.p, S.ldv, n);
            launch_vq(*S.ctx, S.AV.p, S.ldv, size, S.Ydev.p, size, qq, S.AX.p, S.ldv, n);
            // ---- convergence: ||A x - theta x||_2 < tol for the first nev pairs (RitzPairs.h:54-68)
            bool all = true;
            for (int k = 0; k < S.nev; k++)
            {
                const int nrec = launch_resid_norms(*S.ctx, S.AX.p + int64_t(k) * S.ldv, S.X.p + int64_t(k) * S.ldv, S.theta[size_t(k)], n,
                                                    S.partials.p, S.pstride);
                S.reduce_host(nrec, 1);
                const double rn = std::sqrt(S.h_red.p[kSlotBeta2]);
                S.converged[size_t(k)] = rn < tol;
                all = all && rn < tol;
            }
            S.size = size;
            if (all)
            {
                S.info = 0;  // Successful
                break;
            }
            if (S.niter == maxit - 1)
            {
                S.info = 2;  // NotConverging
                break;
            }
            // ---- correction block (DavidsonSymEigsSolver.h:61-76) appended and orthonormalised (SearchSpace.h:65-70)
            const int ncorr = std::min(S.corr_size, qq);
            for (int k = 0; k < ncorr; k++)
            {
                hipLaunchKernelGGL(k_dpr_correction, blocks_for(n), dim3(kThreads), 0, st, S.AX.p + int64_t(k) * S.ldv,
                                   S.X.p + int64_t(k) * S.ldv, S.theta[size_t(k)], S.diag.p, n, S.vcol(size));
                MISPEC_HIP(hipGetLastError());
                if (orthonormalise_column(S, size))
                    size++;
            }
        }
        S.size = size > int(S.theta.size()) ? int(S.theta.size()) : size;
        S.computed = true;
        int64_t cnt = 0;
        for (char c : S.converged)
            cnt += c ? 1 : 0;
        *nconv = cnt;
    });
}

extern "C" int mispec_davidson_info(const mispec_davidson* S) { return S ? S->info : 1; }
extern "C" int64_t mispec_davidson_num_iterations(const mispec_davidson* S) { return S ? S->niter : 0; }
extern "C" int64_t mispec_davidson_num_operations(const mispec_davidson* S) { return S ? S->nops : 0; }

// eigenvalues() / eigenvectors(): the first nev Ritz pairs (JDSymEigsBase.h:112-118)
extern "C" int mispec_davidson_eigenvalues(const mispec_davidson* S, double* out_host)
{
    return guarded([&] {
        MISPEC_REQUIRE(S && out_host, "mispec_davidson_eigenvalues: NULL argument");
        MISPEC_REQUIRE(S->computed && int(S->theta.size()) >= S->nev, "DavidsonSymEigsSolver: compute() has not produced Ritz pairs");
        std::copy(S->theta.begin(), S->theta.begin() + S->nev, out_host);
    });
}

extern "C" int mispec_davidson_eigenvectors(const mispec_davidson* S, double* out_host, int64_t ld)
{
    return guarded([&] {
        MISPEC_REQUIRE(S && out_host && ld >= S->n, "mispec_davidson_eigenvectors: bad argument");
        MISPEC_REQUIRE(S->computed && int(S->theta.size()) >= S->nev, "DavidsonSymEigsSolver: compute() has not produced Ritz pairs");
        S->ctx->make_current();
        MISPEC_HIP(hipMemcpy2D(out_host, size_t(ld) * sizeof(double), S->X.p, size_t(S->ldv) * sizeof(double), size_t(S->n) * sizeof(double),
                               size_t(S->nev), hipMemcpyDeviceToHost));
    });
}
