// SparseCholesky on the GPU — the B operator of SymGEigsSolver<..., GEigsMode::Cholesky>
// (replaces MatOp/SparseCholesky.h:36-128, an Eigen::SimplicialLLT with lower/upper triangular solves).
// The generalized problem A x = lambda B x becomes the standard one L^{-1} A L^{-T} y = lambda y, x = L^{-T} y
// (SymGEigsSolver.h:142-208, MatOp/internal/SymGEigsCholeskyOp.h:63-71).  Any factor G with G G' = B gives the
// same eigenpairs; here G = L is the dense Cholesky factor (no fill-reducing permutation), computed once on the
// host, and the two triangular solves are dense GEMVs with the explicit L^{-1} / L^{-T} held in HBM (n <= 4096).
// Larger B must be BANDED (half-bandwidth <= 64; <= 8 — the usual mass / stiffness matrices of 1-D and structured problems — until round 6): the
// factor is then the one of the partitioned band factorisation of shiftsolve.hip in its nested order (chunk interiors, then
// separators, recursively) — G = [L_II D^{1/2} 0; M_SI L_II^{-T} D^{-1/2} G_S] — and G^{-1} x / G^{-T} x are the two halves of
// the band solve, on the device (launch_band_cholesky_solve).  Other large patterns: regular-inverse mode (conjugate gradient).
#include "cholesky.hpp"
#include "dense.hpp"

#include <cmath>
#include <memory>
#include <vector>

using namespace mispec;

namespace mispec {
void launch_cholesky_solve(const mispec_cholesky& C, bool upper, const double* x, double* y)
{
    if (C.band)
    {
        launch_band_cholesky_solve(*C.band, upper, x, y);
        return;
    }
    launch_row_gemv(*C.ctx, upper ? C.linvt.p : C.linv.p, C.n, C.n, C.n, x, y);  // dense.hip: one wavefront per row
}
}  // namespace mispec

extern "C" int mispec_cholesky_create(mispec_ctx* ctx, int64_t n, const int32_t* outer, const int32_t* inner, const double* val,
                                      char uplo, int row_major, mispec_cholesky** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && out && outer && n >= 1, "mispec_cholesky_create: bad argument");
        MISPEC_REQUIRE(uplo == 'L' || uplo == 'U' || uplo == 'l' || uplo == 'u', "mispec_cholesky_create: uplo must be 'L' or 'U'");
        MISPEC_REQUIRE(ctx->comm.allgather == nullptr, "mispec_cholesky_create: the B operator cannot be row-sharded");
        const bool lower = (uplo == 'L' || uplo == 'l');
        auto C = std::make_unique<mispec_cholesky>();
        C->ctx = ctx;
        C->n = n;
        if (n > kMaxCholesky)
        {
            // banded B: the partitioned band factorisation with sigma = 0, keeping the triangular halves
            mispec_symshift* S = nullptr;
            if (mispec_symshift_create(ctx, n, outer, inner, val, uplo, row_major, &S) != MISPEC_OK)
                throw Error(MISPEC_EINVAL, mispec_last_error());
            C->band = S;
            // (rounds 2-5: half-bandwidth <= 8 only; round 6: the triangular halves run on the wave-per-chunk solve of the wide
            // bands too — k_chunk_solve_wave modes 0 + u_out and 2 —, tested up to 64)
            MISPEC_REQUIRE(S->half_bandwidth <= kMaxBandwidth,
                           "SparseCholesky: for n > 4096 the matrix must be banded (half-bandwidth <= 64, as given or after a reverse "
                           "Cuthill-McKee ordering); use the regular-inverse mode for other large B");
            S->want_cholesky = true;
            const int rc = mispec_symshift_set_shift(S, 0.0);
            C->info = (rc == MISPEC_OK && S->cholesky_ready) ? 0 : 3;  // CompInfo::NumericalIssue: B is not positive definite
            *out = C.release();
            return;
        }
        // dense symmetric B from the selected triangle (column-major, both triangles filled)
        std::vector<double> B(size_t(n) * n, 0.0);
        for (int64_t o = 0; o < n; o++)
            for (int32_t p = outer[o]; p < outer[o + 1]; p++)
            {
                const int64_t in = inner[p];
                MISPEC_REQUIRE(in >= 0 && in < n, "mispec_cholesky_create: index out of range");
                const int64_t r = row_major ? o : in, c = row_major ? in : o;
                if (lower ? (r >= c) : (r <= c))
                {
                    B[size_t(c) * n + r] += val[p];
                    if (r != c)
                        B[size_t(r) * n + c] += val[p];
                }
            }
        // L (lower, column-major, in place): plain Cholesky, column by column
        std::vector<double>& L = B;
        for (int64_t j = 0; j < n && C->info == 0; j++)
        {
            double d = L[size_t(j) * n + j];
            for (int64_t k = 0; k < j; k++)
                d -= L[size_t(k) * n + j] * L[size_t(k) * n + j];
            if (!(d > 0.0))
            {
                C->info = 3;  // CompInfo::NumericalIssue (SparseCholesky.h:77-79)
                break;
            }
            const double ljj = std::sqrt(d);
            L[size_t(j) * n + j] = ljj;
            for (int64_t i = j + 1; i < n; i++)
            {
                double s = L[size_t(j) * n + i];
                for (int64_t k = 0; k < j; k++)
                    s -= L[size_t(k) * n + i] * L[size_t(k) * n + j];
                L[size_t(j) * n + i] = s / ljj;
            }
        }
        if (C->info == 0)
        {
            // X = L^{-1} by forward substitution on the unit vectors; row-major copies of X and X'
            std::vector<double> X(size_t(n) * n, 0.0), Xt(size_t(n) * n, 0.0), col(static_cast<size_t>(n));
            for (int64_t c = 0; c < n; c++)
            {
                std::fill(col.begin(), col.end(), 0.0);
                col[size_t(c)] = 1.0 / L[size_t(c) * n + c];
                for (int64_t i = c + 1; i < n; i++)
                {
                    double s = 0.0;
                    for (int64_t k = c; k < i; k++)
                        s -= L[size_t(k) * n + i] * col[size_t(k)];
                    col[size_t(i)] = s / L[size_t(i) * n + i];
                }
                for (int64_t i = c; i < n; i++)
                {
                    X[size_t(i) * n + c] = col[size_t(i)];   // row i, column c of L^{-1}
                    Xt[size_t(c) * n + i] = col[size_t(i)];  // row c, column i of L^{-T}
                }
            }
            ctx->make_current();
            C->linv.alloc(X.size());
            C->linvt.alloc(Xt.size());
            MISPEC_HIP(hipMemcpy(C->linv.p, X.data(), X.size() * sizeof(double), hipMemcpyHostToDevice));
            MISPEC_HIP(hipMemcpy(C->linvt.p, Xt.data(), Xt.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        *out = C.release();
    });
}

mispec_cholesky::~mispec_cholesky()
{
    if (band)
        (void) mispec_symshift_destroy(band);
}

extern "C" int mispec_cholesky_destroy(mispec_cholesky* C)
{
    return guarded([&] {
        if (C)
        {
            C->ctx->make_current();
            delete C;
        }
    });
}

extern "C" int64_t mispec_cholesky_rows(const mispec_cholesky* C) { return C ? C->n : 0; }
extern "C" int mispec_cholesky_info(const mispec_cholesky* C) { return C ? C->info : 3; }

namespace {
int solve_host(const mispec_cholesky* C, bool upper, const double* x_host, double* y_host, const char* who)
{
    return guarded([&] {
        MISPEC_REQUIRE(C && x_host && y_host, std::string(who) + ": NULL argument");
        if (C->info != 0)
            throw Error(MISPEC_ELOGIC, std::string(who) + ": the Cholesky factorisation failed (B is not positive definite)");
        C->ctx->make_current();
        if (C->stage_x.n < size_t(C->n))
        {
            C->stage_x.alloc(size_t(C->n) + 2);
            C->stage_y.alloc(size_t(C->n) + 2);
        }
        MISPEC_HIP(hipMemcpyAsync(C->stage_x.p, x_host, size_t(C->n) * sizeof(double), hipMemcpyHostToDevice, C->ctx->stream));
        launch_cholesky_solve(*C, upper, C->stage_x.p, C->stage_y.p);
        MISPEC_HIP(hipMemcpyAsync(y_host, C->stage_y.p, size_t(C->n) * sizeof(double), hipMemcpyDeviceToHost, C->ctx->stream));
        MISPEC_HIP(hipStreamSynchronize(C->ctx->stream));
    });
}
}  // namespace

extern "C" int mispec_cholesky_lower_solve_host(const mispec_cholesky* C, const double* x_host, double* y_host)
{
    return solve_host(C, false, x_host, y_host, "mispec_cholesky_lower_solve_host");
}
extern "C" int mispec_cholesky_upper_solve_host(const mispec_cholesky* C, const double* x_host, double* y_host)
{
    return solve_host(C, true, x_host, y_host, "mispec_cholesky_upper_solve_host");
}
