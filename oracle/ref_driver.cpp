// =============================================================================
//  TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
//  C entry points around the REFERENCE'S OWN HEADERS (yixuan/spectra v1.2.0,
//  compiled from where they lie under /root/reference/include by
//  oracle/build_ref.sh; nothing of them is copied into this repository), with
//  oracle/eigen_shim standing in for Eigen, which this image does not have.
//  Output: oracle/_ref/libspectra_ref.so (git-ignored).  It is what pins
//  oracle/spectra_oracle.hpp — the restatement — to the reference's control
//  flow: tests/test_ref_pin.py runs both on the same inputs and compares
//  nconv / num_iterations / num_operations / eigenvalues / the factorisation.
//
//  Every class instantiated below is the reference's: Spectra::SymEigsSolver,
//  SparseSymMatProd, SparseGenMatProd, DenseSymMatProd, Lanczos, Arnoldi,
//  TridiagQR, TridiagEigen, Givens, argsort, SimpleRandom, GenEigsSolver,
//  UpperHessenbergQR, DoubleShiftQR, UpperHessenbergEigen.
// =============================================================================
#include <Eigen/Core>
#include <Eigen/SparseCore>

#include <Spectra/GenEigsSolver.h>
#include <Spectra/LinAlg/Arnoldi.h>
#include <Spectra/LinAlg/DoubleShiftQR.h>
#include <Spectra/LinAlg/Givens.h>
#include <Spectra/LinAlg/Lanczos.h>
#include <Spectra/LinAlg/TridiagEigen.h>
#include <Spectra/LinAlg/UpperHessenbergEigen.h>
#include <Spectra/LinAlg/UpperHessenbergQR.h>
#include <Spectra/MatOp/DenseGenMatProd.h>
#include <Spectra/MatOp/DenseSymMatProd.h>
#include <Spectra/MatOp/SparseGenMatProd.h>
#include <Spectra/MatOp/SparseSymMatProd.h>
#include <Spectra/MatOp/internal/ArnoldiOp.h>
#include <Spectra/SymEigsSolver.h>
#include <Spectra/SymEigsShiftSolver.h>
#include <Spectra/GenEigsRealShiftSolver.h>
#include <Spectra/SymGEigsSolver.h>
#include <Spectra/SymGEigsShiftSolver.h>
#include <Spectra/contrib/PartialSVDSolver.h>
#include <Spectra/Util/SelectionRule.h>
#include <Spectra/Util/SimpleRandom.h>

#include <chrono>
#include <complex>
#include <cstring>
#include <exception>
#include <string>

using namespace Spectra;
typedef Eigen::SparseMatrix<double, Eigen::ColMajor, int> SpCsc;
typedef Eigen::SparseMatrix<double, Eigen::RowMajor, int> SpCsr;
typedef Eigen::Map<const SpCsc> MapCsc;
typedef Eigen::Map<const SpCsr> MapCsr;
typedef Eigen::Map<const Eigen::MatrixXd> MapConstMat;
typedef Eigen::Map<Eigen::MatrixXd> MapMat;
typedef Eigen::Map<Eigen::VectorXd> MapVec;

static std::string g_err;
#define REF_TRY try {
#define REF_CATCH                      \
    }                                  \
    catch (const std::exception& e)    \
    {                                  \
        g_err = e.what();              \
        return -1;                     \
    }

// the operator kinds the entry points accept; kind 0: symmetric CSC of which one triangle is read, 1: general CSC,
// 2: general CSR, 3: dense symmetric (lower triangle read), 4: dense general, 5: callback
struct RefOp
{
    int kind;
    long n;
    const int* ptr;
    const int* ind;
    const double* val;
    int lower;
    void (*cb)(const double*, double*);
};
// a user operator in the reference's plugin concept (SymEigsSolver.h:43-51)
class CallbackOp
{
    long m_n;
    void (*m_cb)(const double*, double*);

public:
    using Scalar = double;
    CallbackOp(long n, void (*cb)(const double*, double*)) : m_n(n), m_cb(cb) {}
    Eigen::Index rows() const { return m_n; }
    Eigen::Index cols() const { return m_n; }
    void perform_op(const double* x, double* y) const { m_cb(x, y); }
};

// the same with the shift-solve concept (SymEigsShiftSolver.h:25-32): the callback IS y = (A - sigma I)^{-1} x for the sigma the
// solver is created with (the caller factors A - sigma I); set_shift records what the solver asked for
class CallbackShiftOp
{
    long m_n;
    void (*m_cb)(const double*, double*);
    mutable double m_sigma = 0.0;

public:
    using Scalar = double;
    CallbackShiftOp(long n, void (*cb)(const double*, double*)) : m_n(n), m_cb(cb) {}
    Eigen::Index rows() const { return m_n; }
    Eigen::Index cols() const { return m_n; }
    void set_shift(const double& sigma) { m_sigma = sigma; }
    void perform_op(const double* x, double* y) const { m_cb(x, y); }
    double shift() const { return m_sigma; }
};

// B operator of the generalized drivers: y = B x by the reference's own SparseSymMatProd (lower triangle of a CSC matrix),
// B^{-1} x by a callback (the reference's SparseRegularInverse delegates it to Eigen::ConjugateGradient, third party)
class PencilBOp
{
    SparseSymMatProd<double, Eigen::Lower> m_prod;
    void (*m_solve)(const double*, double*);

public:
    using Scalar = double;
    PencilBOp(const MapCsc& B, void (*solve)(const double*, double*)) : m_prod(B), m_solve(solve) {}
    Eigen::Index rows() const { return m_prod.rows(); }
    Eigen::Index cols() const { return m_prod.cols(); }
    void perform_op(const double* x, double* y) const { m_prod.perform_op(x, y); }
    void solve(const double* x, double* y) const { m_solve(x, y); }
};

template <typename Solver>
static long run_sym(Solver& eigs, const double* v0, int selection, long maxit, double tol, int sorting, long* counters, double* evals, double* evecs, long n)
{
    if (v0)
        eigs.init(v0);
    else
        eigs.init();
    const long nconv = long(eigs.compute(static_cast<SortRule>(selection), maxit, tol, static_cast<SortRule>(sorting)));
    counters[0] = nconv;
    counters[1] = long(eigs.num_iterations());
    counters[2] = long(eigs.num_operations());
    counters[3] = long(static_cast<int>(eigs.info()));
    const Eigen::VectorXd ev = eigs.eigenvalues();
    for (long i = 0; i < long(ev.size()); i++)
        evals[i] = ev[i];
    if (evecs)
    {
        const Eigen::MatrixXd U = eigs.eigenvectors();
        for (long j = 0; j < long(U.cols()); j++)
            for (long i = 0; i < n; i++)
                evecs[j * n + i] = U(i, j);
    }
    return long(ev.size());
}

// GenEigsSolver: eigenvalues / vectors interleaved complex
template <typename Solver>
static long run_gen(Solver& eigs, const double* v0, int selection, long maxit, double tol, int sorting, long* counters, double* evals, double* evecs, long n)
{
    if (v0)
        eigs.init(v0);
    else
        eigs.init();
    const long nconv = long(eigs.compute(static_cast<SortRule>(selection), maxit, tol, static_cast<SortRule>(sorting)));
    counters[0] = nconv;
    counters[1] = long(eigs.num_iterations());
    counters[2] = long(eigs.num_operations());
    counters[3] = long(static_cast<int>(eigs.info()));
    const Eigen::VectorXcd ev = eigs.eigenvalues();
    for (long i = 0; i < long(ev.size()); i++)
    {
        evals[2 * i] = ev[i].real();
        evals[2 * i + 1] = ev[i].imag();
    }
    if (evecs)
    {
        const Eigen::MatrixXcd U = eigs.eigenvectors();
        for (long j = 0; j < long(U.cols()); j++)
            for (long i = 0; i < n; i++)
            {
                evecs[2 * (j * n + i)] = U(i, j).real();
                evecs[2 * (j * n + i) + 1] = U(i, j).imag();
            }
    }
    return long(ev.size());
}
// the factorisation on its own: Lanczos (symmetric = 1) or Arnoldi; init(v0) then factorize_from(1, m).
// out: V (n x m), H (m x m), f (n); scal = {beta, k, nops}
template <typename OpT>
static int run_fac(const OpT& mop, long n, long m, int symmetric, const double* v0, double* V, double* H, double* f, double* scal)
{
    using AOp = ArnoldiOp<OpT, IdentityBOp>;
    IdentityBOp bop;
    Eigen::Index nops = 0;
    Eigen::Map<const Eigen::VectorXd> v0m(v0, n);
    if (symmetric)
    {
        Lanczos<AOp> fac(AOp(mop, bop), m);
        fac.init(v0m, nops);
        fac.factorize_from(1, m, nops);
        std::memcpy(V, fac.matrix_V().data(), sizeof(double) * size_t(n * m));
        std::memcpy(H, fac.matrix_H().data(), sizeof(double) * size_t(m * m));
        std::memcpy(f, fac.vector_f().data(), sizeof(double) * size_t(n));
        scal[0] = fac.f_norm();
        scal[1] = double(fac.subspace_dim());
    }
    else
    {
        Arnoldi<AOp> fac(AOp(mop, bop), m);
        fac.init(v0m, nops);
        fac.factorize_from(1, m, nops);
        std::memcpy(V, fac.matrix_V().data(), sizeof(double) * size_t(n * m));
        std::memcpy(H, fac.matrix_H().data(), sizeof(double) * size_t(m * m));
        std::memcpy(f, fac.vector_f().data(), sizeof(double) * size_t(n));
        scal[0] = fac.f_norm();
        scal[1] = double(fac.subspace_dim());
    }
    scal[2] = double(nops);
    return 0;
}
// Only the ref_* entry points are exported (build_ref.sh compiles with -fvisibility=hidden): the reference's classes live in
// namespace Spectra, and so do the classes of this repository's include/Spectra inside libmispec.so — left visible, the dynamic
// linker would merge same-named template instantiations of the two libraries when both are loaded into one process.
#pragma GCC visibility push(default)
extern "C" {

const char* ref_last_error() { return g_err.c_str(); }
const char* ref_describe()
{
    return "reference headers: yixuan/spectra (SPECTRA_VERSION "
#define REF_STR2(x) #x
#define REF_STR(x) REF_STR2(x)
        REF_STR(SPECTRA_MAJOR_VERSION) "." REF_STR(SPECTRA_MINOR_VERSION) "." REF_STR(SPECTRA_PATCH_VERSION) "), dense algebra: oracle/eigen_shim (not Eigen)";
}

// Util/SimpleRandom.h: the default start vector
void ref_simple_random(unsigned long seed, long n, double* out)
{
    SimpleRandom<double> rng(seed);
    Eigen::VectorXd v = rng.random_vec(n);
    std::memcpy(out, v.data(), sizeof(double) * size_t(n));
}

// LinAlg/Givens.h
void ref_givens(double x, double y, double* r, double* c, double* s) { Givens<double>::compute_rotation(x, y, *r, *c, *s); }

// Util/SelectionRule.h argsort
int ref_argsort(int rule, const double* values, long len, long* out)
{
    REF_TRY
    Eigen::VectorXd v(len);
    for (long i = 0; i < len; i++)
        v[i] = values[i];
    std::vector<Eigen::Index> ind = argsort(static_cast<SortRule>(rule), v);
    for (long i = 0; i < len; i++)
        out[i] = long(ind[size_t(i)]);
    return 0;
    REF_CATCH
}

// LinAlg/UpperHessenbergQR.h TridiagQR: R, Q'HQ and Q (= apply_YQ on the identity)
int ref_tridiag_qr(long n, const double* T, double shift, double* R, double* QtHQ, double* Q)
{
    REF_TRY
    MapConstMat Tm(T, n, n);
    TridiagQR<double> decomp(n);
    decomp.compute(Tm, shift);
    Eigen::MatrixXd Rm = decomp.matrix_R();
    Eigen::MatrixXd Hm;
    decomp.matrix_QtHQ(Hm);
    Eigen::MatrixXd Qm = Eigen::MatrixXd::Identity(n, n);
    decomp.apply_YQ(Qm);
    std::memcpy(R, Rm.data(), sizeof(double) * size_t(n * n));
    std::memcpy(QtHQ, Hm.data(), sizeof(double) * size_t(n * n));
    std::memcpy(Q, Qm.data(), sizeof(double) * size_t(n * n));
    return 0;
    REF_CATCH
}
int ref_hess_qr(long n, const double* Hin, double shift, double* Q, double* QtHQ)
{
    REF_TRY
    MapConstMat Hm(Hin, n, n);
    UpperHessenbergQR<double> decomp(n);
    decomp.compute(Hm, shift);
    Eigen::MatrixXd Qm = Eigen::MatrixXd::Identity(n, n);
    decomp.apply_YQ(Qm);
    Eigen::MatrixXd Out;
    decomp.matrix_QtHQ(Out);
    std::memcpy(Q, Qm.data(), sizeof(double) * size_t(n * n));
    std::memcpy(QtHQ, Out.data(), sizeof(double) * size_t(n * n));
    return 0;
    REF_CATCH
}
int ref_double_shift_qr(long n, const double* Hin, double s, double t, double* Q, double* QtHQ)
{
    REF_TRY
    MapConstMat Hm(Hin, n, n);
    DoubleShiftQR<double> decomp(n);
    decomp.compute(Hm, s, t);
    Eigen::MatrixXd Qm = Eigen::MatrixXd::Identity(n, n);
    decomp.apply_YQ(Qm);
    Eigen::MatrixXd Out;
    decomp.matrix_QtHQ(Out);
    std::memcpy(Q, Qm.data(), sizeof(double) * size_t(n * n));
    std::memcpy(QtHQ, Out.data(), sizeof(double) * size_t(n * n));
    return 0;
    REF_CATCH
}
// LinAlg/TridiagEigen.h
int ref_tridiag_eigen(long n, const double* T, double* evals, double* evecs)
{
    REF_TRY
    MapConstMat Tm(T, n, n);
    TridiagEigen<double> decomp(Tm);
    std::memcpy(evals, decomp.eigenvalues().data(), sizeof(double) * size_t(n));
    std::memcpy(evecs, decomp.eigenvectors().data(), sizeof(double) * size_t(n * n));
    return 0;
    REF_CATCH
}
// LinAlg/UpperHessenbergEigen.h: eigenvalues / eigenvectors interleaved (re, im)
int ref_hess_eigen(long n, const double* Hin, double* evals, double* evecs)
{
    REF_TRY
    MapConstMat Hm(Hin, n, n);
    UpperHessenbergEigen<double> decomp(Hm);
    const Eigen::VectorXcd& ev = decomp.eigenvalues();
    Eigen::MatrixXcd U = decomp.eigenvectors();
    for (long i = 0; i < n; i++)
    {
        evals[2 * i] = ev[i].real();
        evals[2 * i + 1] = ev[i].imag();
    }
    for (long j = 0; j < n; j++)
        for (long i = 0; i < n; i++)
        {
            evecs[2 * (j * n + i)] = U(i, j).real();
            evecs[2 * (j * n + i) + 1] = U(i, j).imag();
        }
    return 0;
    REF_CATCH
}

// one application of the reference's operator
int ref_op_apply(const RefOp* op, const double* x, double* y)
{
    REF_TRY
    const long n = op->n;
    switch (op->kind)
    {
        case 0:
        {
            MapCsc A(n, n, op->ptr[n], op->ptr, op->ind, op->val);
            if (op->lower)
                SparseSymMatProd<double, Eigen::Lower>(A).perform_op(x, y);
            else
                SparseSymMatProd<double, Eigen::Upper>(A).perform_op(x, y);
            break;
        }
        case 1:
        {
            MapCsc A(n, n, op->ptr[n], op->ptr, op->ind, op->val);
            SparseGenMatProd<double>(A).perform_op(x, y);
            break;
        }
        case 2:
        {
            MapCsr A(n, n, op->ptr[n], op->ptr, op->ind, op->val);
            SparseGenMatProd<double, Eigen::RowMajor>(A).perform_op(x, y);
            break;
        }
        case 3:
        {
            MapConstMat A(op->val, n, n);
            DenseSymMatProd<double>(A).perform_op(x, y);
            break;
        }
        case 4:
        {
            MapConstMat A(op->val, n, n);
            DenseGenMatProd<double>(A).perform_op(x, y);
            break;
        }
        default:
            op->cb(x, y);
    }
    return 0;
    REF_CATCH
}

// SymEigsSolver<OpType>(op, nev, ncv).init(v0 or default).compute(...): counters = {nconv, niter, nops, info}
long ref_symeigs(const RefOp* op, long nev, long ncv, const double* v0, int selection, long maxit, double tol, int sorting, long* counters, double* evals, double* evecs)
{
    REF_TRY
    const long n = op->n;
    switch (op->kind)
    {
        case 0:
        {
            MapCsc A(n, n, op->ptr[n], op->ptr, op->ind, op->val);
            if (op->lower)
            {
                SparseSymMatProd<double, Eigen::Lower> mop(A);
                SymEigsSolver<SparseSymMatProd<double, Eigen::Lower>> eigs(mop, nev, ncv);
                return run_sym(eigs, v0, selection, maxit, tol, sorting, counters, evals, evecs, n);
            }
            SparseSymMatProd<double, Eigen::Upper> mop(A);
            SymEigsSolver<SparseSymMatProd<double, Eigen::Upper>> eigs(mop, nev, ncv);
            return run_sym(eigs, v0, selection, maxit, tol, sorting, counters, evals, evecs, n);
        }
        case 2:
        {
            // a symmetric matrix held as full CSR behind the general product (what tests/ feed the GPU path)
            MapCsr A(n, n, op->ptr[n], op->ptr, op->ind, op->val);
            SparseGenMatProd<double, Eigen::RowMajor> mop(A);
            SymEigsSolver<SparseGenMatProd<double, Eigen::RowMajor>> eigs(mop, nev, ncv);
            return run_sym(eigs, v0, selection, maxit, tol, sorting, counters, evals, evecs, n);
        }
        case 3:
        {
            MapConstMat A(op->val, n, n);
            DenseSymMatProd<double> mop(A);
            SymEigsSolver<DenseSymMatProd<double>> eigs(mop, nev, ncv);
            return run_sym(eigs, v0, selection, maxit, tol, sorting, counters, evals, evecs, n);
        }
        case 5:
        {
            CallbackOp mop(n, op->cb);
            SymEigsSolver<CallbackOp> eigs(mop, nev, ncv);
            return run_sym(eigs, v0, selection, maxit, tol, sorting, counters, evals, evecs, n);
        }
        default:
            g_err = "ref_symeigs: operator kind not symmetric";
            return -1;
    }
    REF_CATCH
}

// SymEigsShiftSolver<CallbackShiftOp>(op, nev, ncv, sigma): the eigenvalues come back transformed (1 / nu + sigma) and sorted
// by the reference's own code (SymEigsShiftSolver.h:190-215)
long ref_symeigs_shift(const RefOp* op, long nev, long ncv, double sigma, const double* v0, int selection, long maxit, double tol, int sorting,
                       long* counters, double* evals, double* evecs)
{
    REF_TRY
    if (op->kind != 5)
    {
        g_err = "ref_symeigs_shift: needs a callback operator (the shift solve)";
        return -1;
    }
    CallbackShiftOp mop(op->n, op->cb);
    SymEigsShiftSolver<CallbackShiftOp> eigs(mop, nev, ncv, sigma);
    if (mop.shift() != sigma)
    {
        g_err = "ref_symeigs_shift: the solver did not set the shift";
        return -1;
    }
    return run_sym(eigs, v0, selection, maxit, tol, sorting, counters, evals, evecs, op->n);
    REF_CATCH
}

// SymGEigsSolver<SparseSymMatProd, PencilBOp, GEigsMode::RegularInverse>(A, B, nev, ncv) (SymGEigsSolver.h:224-238): the Krylov
// operator B^{-1} A and the B-inner products of ArnoldiOp<Op, BOp> are the reference's; a / b: lower triangles as CSC;
// bsolve: y = B^{-1} x
long ref_symgeigs_reginv(const RefOp* a, const RefOp* b, void (*bsolve)(const double*, double*), long nev, long ncv, int selection, long maxit,
                         double tol, int sorting, long* counters, double* evals, double* evecs)
{
    REF_TRY
    const long n = a->n;
    MapCsc A(n, n, a->ptr[n], a->ptr, a->ind, a->val), B(n, n, b->ptr[n], b->ptr, b->ind, b->val);
    SparseSymMatProd<double, Eigen::Lower> aop(A);
    PencilBOp bop(B, bsolve);
    SymGEigsSolver<SparseSymMatProd<double, Eigen::Lower>, PencilBOp, GEigsMode::RegularInverse> eigs(aop, bop, nev, ncv);
    return run_sym(eigs, nullptr, selection, maxit, tol, sorting, counters, evals, evecs, n);
    REF_CATCH
}

// SymGEigsShiftSolver<CallbackShiftOp, SparseSymMatProd, mode>(op, Bop, nev, ncv, sigma) (SymGEigsShiftSolver.h:36-207):
// mode 1 shift-invert, 2 buckling, 3 Cayley; inv: y = (A - sigma B)^{-1} x (buckling: (K - sigma KG)^{-1}); b: the matrix
// of the inner product (B; K for buckling), lower triangle as CSC
long ref_symgeigs_shift(const RefOp* inv, const RefOp* b, int mode, long nev, long ncv, double sigma, int selection, long maxit, double tol,
                        int sorting, long* counters, double* evals, double* evecs)
{
    REF_TRY
    const long n = b->n;
    if (inv->kind != 5)
    {
        g_err = "ref_symgeigs_shift: needs a callback operator (the shift solve)";
        return -1;
    }
    MapCsc B(n, n, b->ptr[n], b->ptr, b->ind, b->val);
    using BOp = SparseSymMatProd<double, Eigen::Lower>;
    BOp bop(B);
    CallbackShiftOp op(n, inv->cb);
    if (mode == 1)
    {
        SymGEigsShiftSolver<CallbackShiftOp, BOp, GEigsMode::ShiftInvert> eigs(op, bop, nev, ncv, sigma);
        return run_sym(eigs, nullptr, selection, maxit, tol, sorting, counters, evals, evecs, n);
    }
    if (mode == 2)
    {
        SymGEigsShiftSolver<CallbackShiftOp, BOp, GEigsMode::Buckling> eigs(op, bop, nev, ncv, sigma);
        return run_sym(eigs, nullptr, selection, maxit, tol, sorting, counters, evals, evecs, n);
    }
    if (mode == 3)
    {
        SymGEigsShiftSolver<CallbackShiftOp, BOp, GEigsMode::Cayley> eigs(op, bop, nev, ncv, sigma);
        return run_sym(eigs, nullptr, selection, maxit, tol, sorting, counters, evals, evecs, n);
    }
    g_err = "ref_symgeigs_shift: mode must be 1, 2 or 3";
    return -1;
    REF_CATCH
}

// contrib/PartialSVDSolver.h:112-209 on a sparse m x n matrix (CSC): compute(), singular_values(), and the singular vectors that
// ARE the eigenvectors of the product operator (V for a tall matrix, U otherwise: n_small x ncomp); counters as in run_sym
long ref_partial_svd(long m, long n, const int* colptr, const int* rowind, const double* val, long ncomp, long ncv, long maxit, double tol,
                     long* counters, double* svals, double* vecs)
{
    REF_TRY
    MapCsc A(m, n, colptr[n], colptr, rowind, val);
    const SpCsc Acopy = A;  // PartialSVDSolver takes a Ref<const SparseMatrix>
    PartialSVDSolver<SpCsc> svds(Acopy, ncomp, ncv);
    const long nconv = long(svds.compute(maxit, tol));
    counters[0] = nconv;
    counters[1] = counters[2] = counters[3] = -1;  // the inner solver is private
    const Eigen::VectorXd sv = svds.singular_values();
    for (long i = 0; i < long(sv.size()); i++)
        svals[i] = sv[i];
    if (vecs)
    {
        const long dim = m > n ? n : m;
        const Eigen::MatrixXd X = m > n ? svds.matrix_V(ncomp) : svds.matrix_U(ncomp);
        for (long j = 0; j < long(X.cols()); j++)
            for (long i = 0; i < dim; i++)
                vecs[j * dim + i] = X(i, j);
    }
    return long(sv.size());
    REF_CATCH
}

// GenEigsRealShiftSolver<CallbackShiftOp>(op, nev, ncv, sigma) (GenEigsRealShiftSolver.h:52-58)
long ref_geneigs_real_shift(const RefOp* op, long nev, long ncv, double sigma, const double* v0, int selection, long maxit, double tol, int sorting,
                            long* counters, double* evals, double* evecs)
{
    REF_TRY
    if (op->kind != 5)
    {
        g_err = "ref_geneigs_real_shift: needs a callback operator (the shift solve)";
        return -1;
    }
    CallbackShiftOp mop(op->n, op->cb);
    GenEigsRealShiftSolver<CallbackShiftOp> eigs(mop, nev, ncv, sigma);
    return run_gen(eigs, v0, selection, maxit, tol, sorting, counters, evals, evecs, op->n);
    REF_CATCH
}

long ref_geneigs(const RefOp* op, long nev, long ncv, const double* v0, int selection, long maxit, double tol, int sorting, long* counters, double* evals, double* evecs)
{
    REF_TRY
    const long n = op->n;
    switch (op->kind)
    {
        case 1:
        {
            MapCsc A(n, n, op->ptr[n], op->ptr, op->ind, op->val);
            SparseGenMatProd<double> mop(A);
            GenEigsSolver<SparseGenMatProd<double>> eigs(mop, nev, ncv);
            return run_gen(eigs, v0, selection, maxit, tol, sorting, counters, evals, evecs, n);
        }
        case 2:
        {
            MapCsr A(n, n, op->ptr[n], op->ptr, op->ind, op->val);
            SparseGenMatProd<double, Eigen::RowMajor> mop(A);
            GenEigsSolver<SparseGenMatProd<double, Eigen::RowMajor>> eigs(mop, nev, ncv);
            return run_gen(eigs, v0, selection, maxit, tol, sorting, counters, evals, evecs, n);
        }
        case 4:
        {
            MapConstMat A(op->val, n, n);
            DenseGenMatProd<double> mop(A);
            GenEigsSolver<DenseGenMatProd<double>> eigs(mop, nev, ncv);
            return run_gen(eigs, v0, selection, maxit, tol, sorting, counters, evals, evecs, n);
        }
        case 5:
        {
            CallbackOp mop(n, op->cb);
            GenEigsSolver<CallbackOp> eigs(mop, nev, ncv);
            return run_gen(eigs, v0, selection, maxit, tol, sorting, counters, evals, evecs, n);
        }
        default:
            g_err = "ref_geneigs: operator kind not supported";
            return -1;
    }
    REF_CATCH
}

int ref_factorize(const RefOp* op, long m, int symmetric, const double* v0, double* V, double* H, double* f, double* scal)
{
    REF_TRY
    const long n = op->n;
    switch (op->kind)
    {
        case 0:
        {
            MapCsc A(n, n, op->ptr[n], op->ptr, op->ind, op->val);
            if (op->lower)
                return run_fac(SparseSymMatProd<double, Eigen::Lower>(A), n, m, symmetric, v0, V, H, f, scal);
            return run_fac(SparseSymMatProd<double, Eigen::Upper>(A), n, m, symmetric, v0, V, H, f, scal);
        }
        case 1:
        {
            MapCsc A(n, n, op->ptr[n], op->ptr, op->ind, op->val);
            return run_fac(SparseGenMatProd<double>(A), n, m, symmetric, v0, V, H, f, scal);
        }
        case 2:
        {
            MapCsr A(n, n, op->ptr[n], op->ptr, op->ind, op->val);
            return run_fac(SparseGenMatProd<double, Eigen::RowMajor>(A), n, m, symmetric, v0, V, H, f, scal);
        }
        case 3:
        {
            MapConstMat A(op->val, n, n);
            return run_fac(DenseSymMatProd<double>(A), n, m, symmetric, v0, V, H, f, scal);
        }
        case 4:
        {
            MapConstMat A(op->val, n, n);
            return run_fac(DenseGenMatProd<double>(A), n, m, symmetric, v0, V, H, f, scal);
        }
        default:
            return run_fac(CallbackOp(n, op->cb), n, m, symmetric, v0, V, H, f, scal);
    }
    REF_CATCH
}

// wall time of init + factorize_from(1, ncv) [+ restarts until `maxit` iterations] of the reference's solver on a CSR matrix:
// the "reference" CPU baseline of bench.py when this library has been built
double ref_symeigs_time(const RefOp* op, long nev, long ncv, long maxit, double tol, long* counters)
{
    double dummy[1024];
    const auto t0 = std::chrono::steady_clock::now();
    const long rc = ref_symeigs(op, nev, ncv, nullptr, int(SortRule::LargestMagn), maxit, tol, int(SortRule::LargestAlge), counters, dummy, nullptr);
    const auto t1 = std::chrono::steady_clock::now();
    if (rc < 0)
        return -1.0;
    return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
#pragma GCC visibility pop
