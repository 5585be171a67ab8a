// Solver-level C entry points: the header-only templates of include/Spectra/ instantiated inside the
// library for bindings that cannot instantiate C++ templates (Python ctypes, cgo, JNI ...).
// Nothing here adds arithmetic: it is Spectra::SymEigsSolver<Spectra::SparseSymMatProd<double>> (or a
// callback operator with the reference's perform_op contract) behind opaque handles.
#include <Spectra/GenEigsComplexShiftSolver.h>
#include <Spectra/GenEigsRealShiftSolver.h>
#include <Spectra/GenEigsSolver.h>
#include <Spectra/LinAlg/DoubleShiftQR.h>
#include <Spectra/LinAlg/UpperHessenbergEigen.h>
#include <Spectra/LinAlg/UpperHessenbergQR.h>
#include <Spectra/LinAlg/UpperHessenbergSchur.h>
#include <Spectra/SymEigsShiftSolver.h>
#include <Spectra/SymEigsSolver.h>
#include <Spectra/SymGEigsShiftSolver.h>
#include <Spectra/SymGEigsSolver.h>

#include <cstring>
#include <memory>
#include <vector>

#include "common.hpp"

using namespace mispec;

namespace {

// A user operator supplied as a C function pointer: the duck-typed OpType concept of the reference
// (SymEigsSolver.h:43-51) expressed for C callers.
class CallbackOp
{
    mispec_ctx* m_ctx;
    mispec_op_fn m_fn;
    void* m_user;
    Spectra::Index m_n;

public:
    using Scalar = double;
    CallbackOp(mispec_ctx* ctx, mispec_op_fn fn, void* user, Spectra::Index n) : m_ctx(ctx), m_fn(fn), m_user(user), m_n(n) {}
    mispec_ctx* mispec_context() const { return m_ctx; }  // where the Krylov basis of this operator lives
    Spectra::Index rows() const { return m_n; }
    Spectra::Index cols() const { return m_n; }
    void perform_op(const double* x_in, double* y_out) const
    {
        if (m_fn(m_user, x_in, y_out) != 0)
            throw std::runtime_error("user perform_op callback reported failure");
    }
};

// y = A2 (A x) on two device matrices (the operator of contrib/PartialSVDSolver.h) for C callers.
class ProductOp
{
    mispec_ctx* m_ctx;
    const mispec_csr *m_first, *m_second;
    mutable std::vector<double> m_cache;

public:
    using Scalar = double;
    ProductOp(mispec_ctx* ctx, const mispec_csr* first, const mispec_csr* second) :
        m_ctx(ctx), m_first(first), m_second(second), m_cache(size_t(mispec_csr_rows(first)))
    {}
    mispec_ctx* mispec_context() const { return m_ctx; }
    const mispec_csr* mispec_product_first() const { return m_first; }
    const mispec_csr* mispec_product_second() const { return m_second; }
    Spectra::Index rows() const { return Spectra::Index(mispec_csr_cols(m_first)); }
    Spectra::Index cols() const { return rows(); }
    void perform_op(const double* x_in, double* y_out) const
    {
        Spectra::internal::check(mispec_spmv_host(m_first, x_in, m_cache.data()));
        Spectra::internal::check(mispec_spmv_host(m_second, m_cache.data(), y_out));
    }
};

// A dense matrix already in HBM (mispec_dense_upload) as an operator for C callers; the C++ classes
// DenseSymMatProd / DenseGenMatProd own theirs.
class DenseHandleOp
{
    mispec_ctx* m_ctx;
    const mispec_dense* m_mat;

public:
    using Scalar = double;
    DenseHandleOp(mispec_ctx* ctx, const mispec_dense* D) : m_ctx(ctx), m_mat(D) {}
    mispec_ctx* mispec_context() const { return m_ctx; }
    const mispec_dense* mispec_dense_matrix() const { return m_mat; }
    Spectra::Index rows() const { return Spectra::Index(mispec_dense_rows(m_mat)); }
    Spectra::Index cols() const { return Spectra::Index(mispec_dense_cols(m_mat)); }
    void perform_op(const double* x_in, double* y_out) const { Spectra::internal::check(mispec_dense_gemv_host(m_mat, x_in, y_out)); }
};

// A user operator on device pointers supplied as a C function pointer.
class DeviceCallbackOp
{
    mispec_ctx* m_ctx;
    mispec_device_op_fn m_fn;
    void* m_user;
    Spectra::Index m_n;

public:
    using Scalar = double;
    DeviceCallbackOp(mispec_ctx* ctx, mispec_device_op_fn fn, void* user, Spectra::Index n) : m_ctx(ctx), m_fn(fn), m_user(user), m_n(n) {}
    mispec_ctx* mispec_context() const { return m_ctx; }
    Spectra::Index rows() const { return m_n; }
    Spectra::Index cols() const { return m_n; }
    void perform_op_device(const double* x_dev, double* y_dev, void* hip_stream) const
    {
        if (m_fn(m_user, x_dev, y_dev, hip_stream) != 0)
            throw std::runtime_error("user device operator callback reported failure");
    }
};

using DevOp = Spectra::SparseSymMatProd<double>;
using DenseSolver = Spectra::SymEigsSolver<DenseHandleOp>;
using DevCbSolver = Spectra::SymEigsSolver<DeviceCallbackOp>;
using DevSolver = Spectra::SymEigsSolver<DevOp>;
using CbSolver = Spectra::SymEigsSolver<CallbackOp>;
using ShiftOp = Spectra::SparseSymShiftSolve<double>;
using ShiftSolver = Spectra::SymEigsShiftSolver<ShiftOp>;
using ProdSolver = Spectra::SymEigsSolver<ProductOp>;
using RegInvBOp = Spectra::SparseRegularInverse<double>;
using GEigsSolver = Spectra::SymGEigsSolver<DevOp, RegInvBOp, Spectra::GEigsMode::RegularInverse>;
using CholBOp = Spectra::SparseCholesky<double>;
using GCholesky = Spectra::SymGEigsSolver<DevOp, CholBOp, Spectra::GEigsMode::Cholesky>;
using PencilOp = Spectra::SymShiftInvert<double>;
using GShiftInvert = Spectra::SymGEigsShiftSolver<PencilOp, DevOp, Spectra::GEigsMode::ShiftInvert>;
using GBuckling = Spectra::SymGEigsShiftSolver<PencilOp, DevOp, Spectra::GEigsMode::Buckling>;
using GCayley = Spectra::SymGEigsShiftSolver<PencilOp, DevOp, Spectra::GEigsMode::Cayley>;

}  // namespace

struct mispec_symeigs
{
    mispec_ctx* ctx = nullptr;
    std::unique_ptr<DevOp> dev_op;
    std::unique_ptr<CallbackOp> cb_op;
    std::unique_ptr<ShiftOp> shift_op;
    std::unique_ptr<ProductOp> prod_op;
    std::unique_ptr<ProdSolver> prod;
    std::unique_ptr<RegInvBOp> b_op;
    std::unique_ptr<GEigsSolver> geigs;
    std::unique_ptr<CholBOp> chol_op;
    std::unique_ptr<GCholesky> g_cholesky;
    std::unique_ptr<PencilOp> pencil_op;
    std::unique_ptr<GShiftInvert> g_shift;
    std::unique_ptr<GBuckling> g_buckling;
    std::unique_ptr<GCayley> g_cayley;
    std::unique_ptr<DenseHandleOp> dense_op;
    std::unique_ptr<DenseSolver> dense;
    std::unique_ptr<DeviceCallbackOp> devcb_op;
    std::unique_ptr<DevCbSolver> devcb;
    std::unique_ptr<DevSolver> dev;
    std::unique_ptr<CbSolver> cb;
    std::unique_ptr<ShiftSolver> shift;
    int64_t nev = 0;

    template <typename F>
    auto visit(F&& f) const
    {
        // the three solver types share HermEigsBase's interface; go through the common base where possible
        if (dev)
            return f(*dev);
        if (shift)
            return f(*shift);
        if (prod)
            return f(*prod);
        if (geigs)
            return f(*geigs);
        if (g_cholesky)
            return f(*g_cholesky);
        if (g_shift)
            return f(*g_shift);
        if (g_buckling)
            return f(*g_buckling);
        if (g_cayley)
            return f(*g_cayley);
        if (dense)
            return f(*dense);
        if (devcb)
            return f(*devcb);
        return f(*cb);
    }
    mispec_fac* fac() const
    {
        return visit([](auto& s) { return s.factorization().handle(); });
    }
};

extern "C" int mispec_symeigs_create(mispec_ctx* ctx, const mispec_csr* A, int64_t nev, int64_t ncv, mispec_symeigs** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && A && out, "mispec_symeigs_create: NULL argument");
        auto s = std::make_unique<mispec_symeigs>();
        s->ctx = ctx;
        s->nev = nev;
        s->dev_op = std::make_unique<DevOp>(ctx, const_cast<mispec_csr*>(A));
        s->dev = std::make_unique<DevSolver>(*s->dev_op, nev, ncv);
        *out = s.release();
    });
}

extern "C" int mispec_symeigs_create_op(mispec_ctx* ctx, mispec_op_fn op, void* op_user, int64_t n, int64_t nev, int64_t ncv,
                                        mispec_symeigs** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && op && out, "mispec_symeigs_create_op: NULL argument");
        auto s = std::make_unique<mispec_symeigs>();
        s->ctx = ctx;
        s->nev = nev;
        s->cb_op = std::make_unique<CallbackOp>(ctx, op, op_user, n);
        s->cb = std::make_unique<CbSolver>(*s->cb_op, nev, ncv);
        *out = s.release();
    });
}

extern "C" int mispec_symeigs_create_dense(mispec_ctx* ctx, const mispec_dense* D, int64_t nev, int64_t ncv, mispec_symeigs** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && D && out, "mispec_symeigs_create_dense: NULL argument");
        auto s = std::make_unique<mispec_symeigs>();
        s->ctx = ctx;
        s->nev = nev;
        s->dense_op = std::make_unique<DenseHandleOp>(ctx, D);
        s->dense = std::make_unique<DenseSolver>(*s->dense_op, nev, ncv);
        *out = s.release();
    });
}

extern "C" int mispec_symeigs_create_device_op(mispec_ctx* ctx, mispec_device_op_fn op, void* op_user, int64_t n, int64_t nev,
                                               int64_t ncv, mispec_symeigs** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && op && out, "mispec_symeigs_create_device_op: NULL argument");
        auto s = std::make_unique<mispec_symeigs>();
        s->ctx = ctx;
        s->nev = nev;
        s->devcb_op = std::make_unique<DeviceCallbackOp>(ctx, op, op_user, n);
        s->devcb = std::make_unique<DevCbSolver>(*s->devcb_op, nev, ncv);
        *out = s.release();
    });
}

extern "C" int mispec_symeigs_create_shift(mispec_ctx* ctx, mispec_symshift* S, int64_t nev, int64_t ncv, double sigma,
                                           mispec_symeigs** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && S && out, "mispec_symeigs_create_shift: NULL argument");
        auto s = std::make_unique<mispec_symeigs>();
        s->ctx = ctx;
        s->nev = nev;
        s->shift_op = std::make_unique<ShiftOp>(ctx, S);
        s->shift = std::make_unique<ShiftSolver>(*s->shift_op, nev, ncv, sigma);
        *out = s.release();
    });
}

extern "C" int mispec_symeigs_create_product(mispec_ctx* ctx, const mispec_csr* A, const mispec_csr* A2, int64_t nev, int64_t ncv,
                                             mispec_symeigs** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && A && A2 && out, "mispec_symeigs_create_product: NULL argument");
        auto s = std::make_unique<mispec_symeigs>();
        s->ctx = ctx;
        s->nev = nev;
        s->prod_op = std::make_unique<ProductOp>(ctx, A, A2);
        s->prod = std::make_unique<ProdSolver>(*s->prod_op, nev, ncv);
        *out = s.release();
    });
}

extern "C" int mispec_symeigs_create_geigs_reginv(mispec_ctx* ctx, const mispec_csr* A, const mispec_reginv* B, int64_t nev, int64_t ncv,
                                                  mispec_symeigs** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && A && B && out, "mispec_symeigs_create_geigs_reginv: NULL argument");
        auto s = std::make_unique<mispec_symeigs>();
        s->ctx = ctx;
        s->nev = nev;
        s->dev_op = std::make_unique<DevOp>(ctx, const_cast<mispec_csr*>(A));
        s->b_op = std::make_unique<RegInvBOp>(ctx, const_cast<mispec_reginv*>(B));
        s->geigs = std::make_unique<GEigsSolver>(*s->dev_op, *s->b_op, nev, ncv);
        *out = s.release();
    });
}

extern "C" int mispec_symeigs_create_geigs_cholesky(mispec_ctx* ctx, const mispec_csr* A, const mispec_cholesky* B, int64_t nev,
                                                    int64_t ncv, mispec_symeigs** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && A && B && out, "mispec_symeigs_create_geigs_cholesky: NULL argument");
        auto s = std::make_unique<mispec_symeigs>();
        s->ctx = ctx;
        s->nev = nev;
        s->dev_op = std::make_unique<DevOp>(ctx, const_cast<mispec_csr*>(A));
        s->chol_op = std::make_unique<CholBOp>(ctx, const_cast<mispec_cholesky*>(B));
        s->g_cholesky = std::make_unique<GCholesky>(*s->dev_op, *s->chol_op, nev, ncv);
        *out = s.release();
    });
}

extern "C" int mispec_symeigs_create_geigs_shift(mispec_ctx* ctx, mispec_symshift* S, const mispec_csr* B, int mode, int64_t nev, int64_t ncv,
                                                 double sigma, mispec_symeigs** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && S && B && out, "mispec_symeigs_create_geigs_shift: NULL argument");
        MISPEC_REQUIRE(mode >= 0 && mode <= 2, "mispec_symeigs_create_geigs_shift: mode must be 0 (shift-invert), 1 (buckling) or 2 (Cayley)");
        auto s = std::make_unique<mispec_symeigs>();
        s->ctx = ctx;
        s->nev = nev;
        s->pencil_op = std::make_unique<PencilOp>(ctx, S);
        s->dev_op = std::make_unique<DevOp>(ctx, const_cast<mispec_csr*>(B));
        if (mode == 0)
            s->g_shift = std::make_unique<GShiftInvert>(*s->pencil_op, *s->dev_op, nev, ncv, sigma);
        else if (mode == 1)
            s->g_buckling = std::make_unique<GBuckling>(*s->pencil_op, *s->dev_op, nev, ncv, sigma);
        else
            s->g_cayley = std::make_unique<GCayley>(*s->pencil_op, *s->dev_op, nev, ncv, sigma);
        *out = s.release();
    });
}

extern "C" int mispec_symeigs_destroy(mispec_symeigs* s)
{
    return guarded([&] { delete s; });
}

extern "C" int mispec_symeigs_init(mispec_symeigs* s, const double* v0_host)
{
    return guarded([&] {
        MISPEC_REQUIRE(s, "mispec_symeigs_init: NULL argument");
        s->visit([&](auto& solver) {
            if (v0_host)
                solver.init(v0_host);
            else
                solver.init();
            return 0;
        });
    });
}

extern "C" int mispec_symeigs_compute(mispec_symeigs* s, int selection, int64_t maxit, double tol, int sorting, int64_t* nconv)
{
    return guarded([&] {
        MISPEC_REQUIRE(s && nconv, "mispec_symeigs_compute: NULL argument");
        MISPEC_REQUIRE(selection >= 0 && selection <= int(Spectra::SortRule::BothEnds) && sorting >= 0 &&
                           sorting <= int(Spectra::SortRule::BothEnds),
                       "mispec_symeigs_compute: unknown SortRule value");
        *nconv = s->visit([&](auto& solver) {
            return int64_t(solver.compute(static_cast<Spectra::SortRule>(selection), Spectra::Index(maxit), tol,
                                          static_cast<Spectra::SortRule>(sorting)));
        });
    });
}

extern "C" int mispec_symeigs_info(const mispec_symeigs* s)
{
    return s ? s->visit([](auto& solver) { return int(solver.info()); }) : int(Spectra::CompInfo::NotComputed);
}
extern "C" int64_t mispec_symeigs_num_iterations(const mispec_symeigs* s)
{
    return s ? s->visit([](auto& solver) { return int64_t(solver.num_iterations()); }) : 0;
}
extern "C" int64_t mispec_symeigs_num_operations(const mispec_symeigs* s)
{
    return s ? s->visit([](auto& solver) { return int64_t(solver.num_operations()); }) : 0;
}

extern "C" int mispec_symeigs_eigenvalues(const mispec_symeigs* s, double* out_host, int64_t* count)
{
    return guarded([&] {
        MISPEC_REQUIRE(s && count, "mispec_symeigs_eigenvalues: NULL argument");
        s->visit([&](auto& solver) {
            const auto ev = solver.eigenvalues();
            *count = ev.size();
            if (out_host)
                std::memcpy(out_host, ev.data(), size_t(ev.size()) * sizeof(double));
            return 0;
        });
    });
}

extern "C" int mispec_symeigs_eigenvectors(mispec_symeigs* s, int64_t nvec, double* out_host, int64_t* ncols)
{
    return guarded([&] {
        MISPEC_REQUIRE(s && ncols, "mispec_symeigs_eigenvectors: NULL argument");
        s->visit([&](auto& solver) {
            if (!out_host)  // keep the result in HBM only
            {
                *ncols = solver.eigenvectors_on_device(Spectra::Index(nvec));
                return 0;
            }
            *ncols = solver.eigenvectors_to(out_host, Spectra::Index(nvec));  // straight into the caller's memory
            return 0;
        });
    });
}

extern "C" int mispec_symeigs_residuals(mispec_symeigs* s, double* resid_host, int64_t* count)
{
    return guarded([&] {
        MISPEC_REQUIRE(s && resid_host && count, "mispec_symeigs_residuals: NULL argument");
        s->visit([&](auto& solver) {
            const auto ev = solver.eigenvalues();
            *count = ev.size();
            if (ev.size() == 0)
                return 0;
            (void) solver.eigenvectors_on_device(Spectra::Index(ev.size()));  // X = V*Y stays in HBM
            const int rc = mispec_fac_residuals(solver.factorization().handle(), ev.data(), int(ev.size()), resid_host);
            if (rc != MISPEC_OK)
                throw Error(rc, mispec_last_error());
            return 0;
        });
    });
}

extern "C" int mispec_symeigs_profile(mispec_symeigs* s, int enable)
{
    return s ? mispec_fac_profile(s->fac(), enable) : MISPEC_EINVAL;
}
extern "C" int mispec_symeigs_get_profile(const mispec_symeigs* s, mispec_profile* out)
{
    return s ? mispec_fac_get_profile(s->fac(), out) : MISPEC_EINVAL;
}
extern "C" int mispec_symeigs_overlap_info(const mispec_symeigs* s, int* first_block, int* block_count, int* total_blocks)
{
    return s ? mispec_fac_overlap_info(s->fac(), first_block, block_count, total_blocks) : MISPEC_EINVAL;
}
extern "C" int mispec_symeigs_set_orth_mode(mispec_symeigs* s, int mode)
{
    return s ? mispec_fac_set_orth_mode(s->fac(), mode) : MISPEC_EINVAL;
}
extern "C" int mispec_symeigs_orth_info(const mispec_symeigs* s, int* mode, int64_t* lagged_steps, int64_t* check_stops,
                                        int64_t* state_stops, double* max_rel_c, double* max_chk)
{
    return s ? mispec_fac_orth_info(s->fac(), mode, lagged_steps, check_stops, state_stops, max_rel_c, max_chk) : MISPEC_EINVAL;
}
extern "C" int mispec_symeigs_onered_steps(const mispec_symeigs* s, int64_t* steps)
{
    return s ? mispec_fac_onered_steps(s->fac(), steps) : MISPEC_EINVAL;
}
extern "C" int mispec_symeigs_turn_info(const mispec_symeigs* s, int64_t* turns, double* host_seconds, int64_t* fallbacks)
{
    return s ? mispec_fac_turn_info(s->fac(), turns, host_seconds, fallbacks) : MISPEC_EINVAL;
}

extern "C" int mispec_symeigs_restart_info(const mispec_symeigs* s, int64_t* fused, int64_t* recorrected)
{
    return s ? mispec_fac_restart_info(s->fac(), fused, recorrected) : MISPEC_EINVAL;
}
extern "C" int mispec_symeigs_exchange_info(const mispec_symeigs* s, int* halo, int64_t* recv_doubles)
{
    return s ? mispec_fac_exchange_info(s->fac(), halo, recv_doubles) : MISPEC_EINVAL;
}

// =================================================================================================
// General solver facade
// =================================================================================================
namespace {
using GenDevOp = Spectra::SparseGenMatProd<double>;
using GenDevSolver = Spectra::GenEigsSolver<GenDevOp>;
using GenCbSolver = Spectra::GenEigsSolver<CallbackOp>;
using GenDenseSolver = Spectra::GenEigsSolver<DenseHandleOp>;
using GenDevCbSolver = Spectra::GenEigsSolver<DeviceCallbackOp>;
using GenShiftOp = Spectra::SparseGenRealShiftSolve<double>;
using GenShiftSolver = Spectra::GenEigsRealShiftSolver<GenShiftOp>;
using GenCShiftOp = Spectra::SparseGenComplexShiftSolve<double>;
using GenCShiftSolver = Spectra::GenEigsComplexShiftSolver<GenCShiftOp>;
}  // namespace

struct mispec_geneigs
{
    std::unique_ptr<GenDevOp> dev_op;
    std::unique_ptr<CallbackOp> cb_op;
    std::unique_ptr<GenShiftOp> shift_op;
    std::unique_ptr<GenDevSolver> dev;
    std::unique_ptr<GenCbSolver> cb;
    std::unique_ptr<GenShiftSolver> shift;
    std::unique_ptr<GenCShiftOp> cshift_op;
    std::unique_ptr<GenCShiftSolver> cshift;
    std::unique_ptr<DenseHandleOp> dense_op;
    std::unique_ptr<GenDenseSolver> dense;
    std::unique_ptr<DeviceCallbackOp> devcb_op;
    std::unique_ptr<GenDevCbSolver> devcb;
    template <typename F>
    auto visit(F&& f) const
    {
        if (dev)
            return f(*dev);
        if (shift)
            return f(*shift);
        if (cshift)
            return f(*cshift);
        if (dense)
            return f(*dense);
        if (devcb)
            return f(*devcb);
        return f(*cb);
    }
    mispec_fac* fac() const
    {
        return visit([](auto& s) { return s.factorization().handle(); });
    }
};

extern "C" int mispec_geneigs_create(mispec_ctx* ctx, const mispec_csr* A, int64_t nev, int64_t ncv, mispec_geneigs** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && A && out, "mispec_geneigs_create: NULL argument");
        auto s = std::make_unique<mispec_geneigs>();
        s->dev_op = std::make_unique<GenDevOp>(ctx, const_cast<mispec_csr*>(A));
        s->dev = std::make_unique<GenDevSolver>(*s->dev_op, nev, ncv);
        *out = s.release();
    });
}
extern "C" int mispec_geneigs_create_op(mispec_ctx* ctx, mispec_op_fn op, void* op_user, int64_t n, int64_t nev, int64_t ncv,
                                        mispec_geneigs** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && op && out, "mispec_geneigs_create_op: NULL argument");
        auto s = std::make_unique<mispec_geneigs>();
        s->cb_op = std::make_unique<CallbackOp>(ctx, op, op_user, n);
        s->cb = std::make_unique<GenCbSolver>(*s->cb_op, nev, ncv);
        *out = s.release();
    });
}
extern "C" int mispec_geneigs_create_dense(mispec_ctx* ctx, const mispec_dense* D, int64_t nev, int64_t ncv, mispec_geneigs** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && D && out, "mispec_geneigs_create_dense: NULL argument");
        auto s = std::make_unique<mispec_geneigs>();
        s->dense_op = std::make_unique<DenseHandleOp>(ctx, D);
        s->dense = std::make_unique<GenDenseSolver>(*s->dense_op, nev, ncv);
        *out = s.release();
    });
}
extern "C" int mispec_geneigs_create_device_op(mispec_ctx* ctx, mispec_device_op_fn op, void* op_user, int64_t n, int64_t nev,
                                               int64_t ncv, mispec_geneigs** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && op && out, "mispec_geneigs_create_device_op: NULL argument");
        auto s = std::make_unique<mispec_geneigs>();
        s->devcb_op = std::make_unique<DeviceCallbackOp>(ctx, op, op_user, n);
        s->devcb = std::make_unique<GenDevCbSolver>(*s->devcb_op, nev, ncv);
        *out = s.release();
    });
}
extern "C" int mispec_geneigs_create_shift(mispec_ctx* ctx, mispec_symshift* S, int64_t nev, int64_t ncv, double sigma, mispec_geneigs** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && S && out, "mispec_geneigs_create_shift: NULL argument");
        auto s = std::make_unique<mispec_geneigs>();
        s->shift_op = std::make_unique<GenShiftOp>(ctx, S);
        s->shift = std::make_unique<GenShiftSolver>(*s->shift_op, nev, ncv, sigma);
        *out = s.release();
    });
}
extern "C" int mispec_geneigs_create_complex_shift(mispec_ctx* ctx, mispec_symshift* S, int64_t nev, int64_t ncv, double sigmar,
                                                   double sigmai, mispec_geneigs** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && S && out, "mispec_geneigs_create_complex_shift: NULL argument");
        auto s = std::make_unique<mispec_geneigs>();
        s->cshift_op = std::make_unique<GenCShiftOp>(ctx, S);
        s->cshift = std::make_unique<GenCShiftSolver>(*s->cshift_op, nev, ncv, sigmar, sigmai);
        *out = s.release();
    });
}
extern "C" int mispec_geneigs_destroy(mispec_geneigs* s)
{
    return guarded([&] { delete s; });
}
extern "C" int mispec_geneigs_init(mispec_geneigs* s, const double* v0_host)
{
    return guarded([&] {
        MISPEC_REQUIRE(s, "mispec_geneigs_init: NULL argument");
        s->visit([&](auto& solver) {
            if (v0_host)
                solver.init(v0_host);
            else
                solver.init();
            return 0;
        });
    });
}
extern "C" int mispec_geneigs_compute(mispec_geneigs* s, int selection, int64_t maxit, double tol, int sorting, int64_t* nconv)
{
    return guarded([&] {
        MISPEC_REQUIRE(s && nconv, "mispec_geneigs_compute: NULL argument");
        MISPEC_REQUIRE(selection >= 0 && selection <= int(Spectra::SortRule::BothEnds) && sorting >= 0 &&
                           sorting <= int(Spectra::SortRule::BothEnds),
                       "mispec_geneigs_compute: unknown SortRule value");
        *nconv = s->visit([&](auto& solver) {
            return int64_t(solver.compute(static_cast<Spectra::SortRule>(selection), Spectra::Index(maxit), tol,
                                          static_cast<Spectra::SortRule>(sorting)));
        });
    });
}
extern "C" int mispec_geneigs_info(const mispec_geneigs* s)
{
    return s ? s->visit([](auto& solver) { return int(solver.info()); }) : int(Spectra::CompInfo::NotComputed);
}
extern "C" int64_t mispec_geneigs_num_iterations(const mispec_geneigs* s)
{
    return s ? s->visit([](auto& solver) { return int64_t(solver.num_iterations()); }) : 0;
}
extern "C" int64_t mispec_geneigs_num_operations(const mispec_geneigs* s)
{
    return s ? s->visit([](auto& solver) { return int64_t(solver.num_operations()); }) : 0;
}
extern "C" int mispec_geneigs_eigenvalues(const mispec_geneigs* s, double* out_host, int64_t* count)
{
    return guarded([&] {
        MISPEC_REQUIRE(s && count, "mispec_geneigs_eigenvalues: NULL argument");
        s->visit([&](auto& solver) {
            const auto ev = solver.eigenvalues();
            *count = ev.size();
            if (out_host)
                std::memcpy(out_host, ev.data(), size_t(ev.size()) * sizeof(std::complex<double>));
            return 0;
        });
    });
}
extern "C" int mispec_geneigs_eigenvectors(mispec_geneigs* s, int64_t nvec, double* out_host, int64_t* ncols)
{
    return guarded([&] {
        MISPEC_REQUIRE(s && ncols && out_host, "mispec_geneigs_eigenvectors: NULL argument");
        s->visit([&](auto& solver) {
            const auto X = solver.eigenvectors(Spectra::Index(nvec));
            *ncols = X.cols();
            if (X.size() > 0)
                std::memcpy(out_host, X.data(), size_t(X.size()) * sizeof(std::complex<double>));
            return 0;
        });
    });
}
extern "C" int mispec_geneigs_residuals(mispec_geneigs* s, double* resid_host, int64_t* count)
{
    return guarded([&] {
        MISPEC_REQUIRE(s && resid_host && count, "mispec_geneigs_residuals: NULL argument");
        s->visit([&](auto& solver) {
            const auto r = solver.residuals();
            *count = r.size();
            if (r.size() > 0)
                std::memcpy(resid_host, r.data(), size_t(r.size()) * sizeof(double));
            return 0;
        });
    });
}
extern "C" int mispec_geneigs_profile(mispec_geneigs* s, int enable)
{
    return s ? mispec_fac_profile(s->fac(), enable) : MISPEC_EINVAL;
}
extern "C" int mispec_geneigs_get_profile(const mispec_geneigs* s, mispec_profile* out)
{
    return s ? mispec_fac_get_profile(s->fac(), out) : MISPEC_EINVAL;
}

// =================================================================================================
// Host-side small kernels of the general restart (no GPU involved)
// =================================================================================================
namespace {
Spectra::DenseMatrix<double> as_matrix(int n, const double* p)
{
    Spectra::DenseMatrix<double> M(n, n);
    std::memcpy(M.data(), p, size_t(n) * n * sizeof(double));
    return M;
}
Spectra::DenseMatrix<double> identity(int n)
{
    Spectra::DenseMatrix<double> M(n, n);
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++)
            M(i, j) = (i == j) ? 1.0 : 0.0;
    return M;
}
}  // namespace

extern "C" int mispec_hess_qr_host(int n, const double* H, double shift, double* Q, double* QtHQ)
{
    return guarded([&] {
        MISPEC_REQUIRE(H && n >= 2, "mispec_hess_qr_host: bad argument");
        Spectra::UpperHessenbergQR<double> qr(as_matrix(n, H), shift);
        if (Q)
        {
            auto q = identity(n);
            qr.apply_YQ(q);
            std::memcpy(Q, q.data(), size_t(n) * n * sizeof(double));
        }
        if (QtHQ)
        {
            Spectra::DenseMatrix<double> d;
            qr.matrix_QtHQ(d);
            std::memcpy(QtHQ, d.data(), size_t(n) * n * sizeof(double));
        }
    });
}
extern "C" int mispec_double_shift_qr_host(int n, const double* H, double s, double t, double* Q, double* QtHQ)
{
    return guarded([&] {
        MISPEC_REQUIRE(H && n >= 3, "mispec_double_shift_qr_host: bad argument");
        Spectra::DoubleShiftQR<double> qr(as_matrix(n, H), s, t);
        if (Q)
        {
            auto q = identity(n);
            qr.apply_YQ(q);
            std::memcpy(Q, q.data(), size_t(n) * n * sizeof(double));
        }
        if (QtHQ)
        {
            Spectra::DenseMatrix<double> d;
            qr.matrix_QtHQ(d);
            std::memcpy(QtHQ, d.data(), size_t(n) * n * sizeof(double));
        }
    });
}
extern "C" int mispec_hess_schur_host(int n, const double* H, double* T, double* U)
{
    return guarded([&] {
        MISPEC_REQUIRE(H && T && U && n >= 1, "mispec_hess_schur_host: bad argument");
        Spectra::UpperHessenbergSchur<double> sc(as_matrix(n, H));
        std::memcpy(T, sc.matrix_T().data(), size_t(n) * n * sizeof(double));
        std::memcpy(U, sc.matrix_U().data(), size_t(n) * n * sizeof(double));
    });
}
extern "C" int mispec_hess_eigen_host(int n, const double* H, double* evals, double* evecs)
{
    return guarded([&] {
        MISPEC_REQUIRE(H && evals && n >= 1, "mispec_hess_eigen_host: bad argument");
        Spectra::UpperHessenbergEigen<double> eg(as_matrix(n, H));
        std::memcpy(evals, eg.eigenvalues().data(), size_t(n) * sizeof(std::complex<double>));
        if (evecs)
        {
            const auto V = eg.eigenvectors();
            std::memcpy(evecs, V.data(), size_t(n) * n * sizeof(std::complex<double>));
        }
    });
}
