// =============================================================================
//  TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
//  CPU restatement (Eigen-free, plain std::vector, column-major) of the hot
//  path of yixuan/spectra v1.2.0: implicitly-restarted Lanczos (symmetric).
//  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
//  build, link, import or execute anything under oracle/.  The product path
//  (spectra_amd/, include/) never includes or links this file.
//
//  Every function cites the reference file:line whose arithmetic and control
//  flow it follows (paths relative to /root/reference/include/Spectra/).
//
//  PINNING STATUS: pinned to the reference's own code.  The reference cannot be
//  BUILT with its own build system in this image (every header needs Eigen
//  3.4.0, fetched from the network by CMakeLists.txt:25-38, absent here), but
//  its headers compile unmodified, where they lie, against oracle/eigen_shim —
//  a small dense/sparse algebra with Eigen's names whose reductions run left to
//  right, the order used below (oracle/build_ref.sh -> oracle/_ref/
//  libspectra_ref.so, oracle/ref_driver.cpp).  tests/test_ref_pin.py compares
//  this restatement with that library BIT FOR BIT: SimpleRandom, argsort,
//  Givens, TridiagQR, TridiagEigen, UpperHessenbergQR, DoubleShiftQR,
//  UpperHessenbergEigen, the sparse products, Lanczos / Arnoldi factorisations
//  (V, H, f), and whole SymEigsSolver / GenEigsSolver solves (nconv, info,
//  num_iterations, num_operations, eigenvalues, eigenvectors) on the
//  reference's fixtures x all selection rules, Example1/2/4 and graded /
//  clustered spectra; SymEigsShiftSolver / GenEigsRealShiftSolver (the back-
//  transformation and sorting of the Ritz values) on the reference's shift
//  fixtures with the shift solve handed to both sides as the same callback;
//  the generalized drivers with B-inner products (SymGEigsSolver regular
//  inverse, SymGEigsShiftSolver shift-invert / buckling / Cayley) with the
//  inverses as callbacks and B x from the reference's own SparseSymMatProd;
//  contrib/PartialSVDSolver (product operators and driver).  The vectors the library returned are committed
//  (tests/golden/ref_pin_golden.npz, generator alongside) and checked on every
//  run, with or without the library.  What stays outside the pin is real
//  Eigen's vectorised reduction order and its third-party kernels (SparseLU,
//  ConjugateGradient): the pieces marked [Eigen] are restated from Eigen
//  3.4.0's published scalar algorithms, so against a build with real Eigen the
//  oracle is an oracle for RESULTS TO TOLERANCE, not for bits.
// =============================================================================
#pragma once

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <limits>
#include <memory>
#include <vector>

namespace oracle {

using Index = long;

// ----------------------------------------------------------------------------
// Util/TypeTraits.h:66-76  (double specialisation)
// ----------------------------------------------------------------------------
constexpr double kEps = DBL_EPSILON;   // TypeTraits<double>::epsilon()
constexpr double kMin = DBL_MIN;       // TypeTraits<double>::min()
constexpr double kNear0 = kMin * 10.0; // "near_0": Arnoldi.h:50, HermEigsBase.h:185

// ----------------------------------------------------------------------------
// Util/SelectionRule.h:33-58 and Util/CompInfo.h:17-32 (same enumerator order)
// ----------------------------------------------------------------------------
enum class SortRule : int
{
    LargestMagn = 0,
    LargestReal,
    LargestImag,
    LargestAlge,
    SmallestMagn,
    SmallestReal,
    SmallestImag,
    SmallestAlge,
    BothEnds
};

enum class CompInfo : int
{
    Successful = 0,
    NotComputed,
    NotConverging,
    NumericalIssue
};

// ----------------------------------------------------------------------------
// Util/SimpleRandom.h:30-52: x <- 16807 * x mod (2^31 - 1).  The reference
// evaluates the product with a 16-bit split and end-around carries; for
// 1 <= x <= 2^31-2 that is exactly the 64-bit modular product below (the
// modulus is prime, so the folded value never lands on 2^31-1 itself).  The one
// state outside that range a seed can produce, x = 2^31-1 (seed & m_max, :95),
// is a fixed point of the folded arithmetic (checked against the reference's
// code, tests/test_ref_pin.py); x = 0 (seed 2^31) stays 0 either way.
// ----------------------------------------------------------------------------
inline long lcg_next(long x)
{
    if (x == 2147483647L)
        return x;
    return static_cast<long>((16807ULL * static_cast<unsigned long long>(x)) % 2147483647ULL);
}

// Util/SimpleRandom.h:88-123 (+ RandomScalar<double>::run at :56-66)
class SimpleRandom
{
    long state_;

public:
    explicit SimpleRandom(unsigned long seed)
    {
        state_ = seed ? static_cast<long>(seed & 2147483647UL) : 1;  // :92-96
    }
    double random()
    {
        state_ = lcg_next(state_);
        return double(state_) / double(2147483647L) - 0.5;  // :64-65
    }
    void fill(double* v, Index n)
    {
        for (Index i = 0; i < n; i++)
            v[i] = random();
    }
};

// ----------------------------------------------------------------------------
// Minimal dense column-major matrix
// ----------------------------------------------------------------------------
struct Mat
{
    Index rows = 0, cols = 0;
    std::vector<double> a;
    Mat() {}
    Mat(Index r, Index c) : rows(r), cols(c), a(size_t(r) * size_t(c), 0.0) {}
    void resize(Index r, Index c)
    {
        rows = r;
        cols = c;
        a.assign(size_t(r) * size_t(c), 0.0);
    }
    double& operator()(Index i, Index j) { return a[size_t(j) * rows + i]; }
    const double& operator()(Index i, Index j) const { return a[size_t(j) * rows + i]; }
    double* col(Index j) { return a.data() + size_t(j) * rows; }
    const double* col(Index j) const { return a.data() + size_t(j) * rows; }
    void set_identity()
    {
        std::fill(a.begin(), a.end(), 0.0);
        for (Index i = 0; i < std::min(rows, cols); i++)
            (*this)(i, i) = 1.0;
    }
};

// ----------------------------------------------------------------------------
// Matrix operators — the plugin concept of SymEigsSolver.h:43-51
//   rows(), cols(), perform_op(x_in, y_out)
// ----------------------------------------------------------------------------
struct Op
{
    virtual ~Op() {}
    virtual Index rows() const = 0;
    virtual Index cols() const { return rows(); }
    virtual void perform_op(const double* x, double* y) const = 0;
};

// MatOp/SparseSymMatProd.h:83-88 with Flags=ColMajor: y = selfadjointView<Uplo>(A) * x.
// [Eigen] The arithmetic is Eigen 3.4.0's sparse self-adjoint * dense product
// (SparseSelfAdjointView.h, sparse_selfadjoint_time_dense_product): y is zeroed;
// per stored column j the diagonal term is added first, then each strictly
// lower (upper) entry a_ij contributes  y[i] += a_ij * x[j]  and to a scalar
// accumulator  r_j += a_ij * x[i], which is added to y[j] after the column.
// Entries in the other triangle are skipped (test/SymEigs.cpp:27-28 relies on that).
struct SparseSymCsc : Op
{
    Index n;
    bool lower;
    std::vector<int> colptr, rowind;
    std::vector<double> val;
    SparseSymCsc(Index n_, const int* cp, const int* ri, const double* v, bool lower_) :
        n(n_), lower(lower_), colptr(cp, cp + n_ + 1), rowind(ri, ri + cp[n_]), val(v, v + cp[n_])
    {}
    Index rows() const override { return n; }
    void perform_op(const double* x, double* y) const override
    {
        for (Index i = 0; i < n; i++)
            y[i] = 0.0;
        for (Index j = 0; j < n; j++)
        {
            int p = colptr[j];
            const int pe = colptr[j + 1];
            if (lower)
            {
                while (p < pe && rowind[p] < j)
                    p++;
                if (p < pe && rowind[p] == j)
                {
                    y[j] += val[p] * x[j];
                    p++;
                }
                const double xj = x[j];
                double rj = 0.0;
                for (; p < pe; p++)
                {
                    const double a = val[p];
                    rj += a * x[rowind[p]];
                    y[rowind[p]] += a * xj;
                }
                y[j] += rj;
            }
            else
            {
                const double xj = x[j];
                double rj = 0.0;
                for (; p < pe && rowind[p] < j; p++)
                {
                    const double a = val[p];
                    rj += a * x[rowind[p]];
                    y[rowind[p]] += a * xj;
                }
                y[j] += rj;
                if (p < pe && rowind[p] == j)
                    y[j] += val[p] * x[j];
            }
        }
    }
};

// MatOp/SparseGenMatProd.h:82-87 with Flags=RowMajor (CSR): per-row dot product,
// accumulated sequentially in storage order. [Eigen] sparse_time_dense_product, row-major branch.
struct SparseCsr : Op
{
    Index nr, nc;
    std::vector<int> rowptr, colind;
    std::vector<double> val;
    SparseCsr(Index nr_, Index nc_, const int* rp, const int* ci, const double* v) :
        nr(nr_), nc(nc_), rowptr(rp, rp + nr_ + 1), colind(ci, ci + rp[nr_]), val(v, v + rp[nr_])
    {}
    Index rows() const override { return nr; }
    Index cols() const override { return nc; }
    void perform_op(const double* x, double* y) const override
    {
        for (Index i = 0; i < nr; i++)
        {
            double s = 0.0;
            for (int p = rowptr[i]; p < rowptr[i + 1]; p++)
                s += val[p] * x[colind[p]];
            y[i] = s;
        }
    }
};

// MatOp/SparseGenMatProd.h:82-87 with Flags=ColMajor (CSC, the reference default):
// y = 0; for each column j: y += x[j] * A[:, j]. [Eigen] col-major branch.
struct SparseCsc : Op
{
    Index nr, nc;
    std::vector<int> colptr, rowind;
    std::vector<double> val;
    SparseCsc(Index nr_, Index nc_, const int* cp, const int* ri, const double* v) :
        nr(nr_), nc(nc_), colptr(cp, cp + nc_ + 1), rowind(ri, ri + cp[nc_]), val(v, v + cp[nc_])
    {}
    Index rows() const override { return nr; }
    Index cols() const override { return nc; }
    void perform_op(const double* x, double* y) const override
    {
        for (Index i = 0; i < nr; i++)
            y[i] = 0.0;
        for (Index j = 0; j < nc; j++)
        {
            const double xj = x[j];
            for (int p = colptr[j]; p < colptr[j + 1]; p++)
                y[rowind[p]] += val[p] * xj;
        }
    }
};

// MatOp/DenseSymMatProd.h (lower triangle of a dense column-major matrix).  Only
// needed because the reference's own unit tests for the factorisation and the
// regression examples (test/Example1.cpp, Example2.cpp, Example4.cpp) are dense.
struct DenseSym : Op
{
    Index n;
    std::vector<double> a;  // column-major n x n, lower triangle referenced
    DenseSym(Index n_, const double* a_) : n(n_), a(a_, a_ + size_t(n_) * n_) {}
    Index rows() const override { return n; }
    void perform_op(const double* x, double* y) const override
    {
        for (Index i = 0; i < n; i++)
        {
            double s = 0.0;
            for (Index j = 0; j < n; j++)
            {
                const double aij = (i >= j) ? a[size_t(j) * n + i] : a[size_t(i) * n + j];
                s += aij * x[j];
            }
            y[i] = s;
        }
    }
};

// Dense general operator (MatOp/DenseGenMatProd.h), for the Arnoldi unit test only.
struct DenseGen : Op
{
    Index n;
    std::vector<double> a;
    DenseGen(Index n_, const double* a_) : n(n_), a(a_, a_ + size_t(n_) * n_) {}
    Index rows() const override { return n; }
    void perform_op(const double* x, double* y) const override
    {
        for (Index i = 0; i < n; i++)
            y[i] = 0.0;
        for (Index j = 0; j < n; j++)
            for (Index i = 0; i < n; i++)
                y[i] += a[size_t(j) * n + i] * x[j];
    }
};

// A user-supplied callable, e.g. the MyDiagonalTen doc example (SymEigsSolver.h:99-126)
struct CallbackOp : Op
{
    Index n;
    std::function<void(const double*, double*)> fn;
    CallbackOp(Index n_, std::function<void(const double*, double*)> f) : n(n_), fn(std::move(f)) {}
    Index rows() const override { return n; }
    void perform_op(const double* x, double* y) const override { fn(x, y); }
};

// ----------------------------------------------------------------------------
// MatOp/internal/ArnoldiOp.h:113-162 (B = I specialisation).
// [Eigen] x.dot(y), X.adjoint()*y, x.norm(): plain left-to-right sums here
// (Eigen's vectorised reduction order is internal and not reproducible).
// x.norm() is sqrt(sum x^2) without scaling (ArnoldiOp.h:152-155).
// ----------------------------------------------------------------------------
inline double dot(const double* x, const double* y, Index n)
{
    double s = 0.0;
    for (Index i = 0; i < n; i++)
        s += x[i] * y[i];
    return s;
}
inline double norm2(const double* x, Index n) { return std::sqrt(dot(x, x, n)); }
// res[j] = <V[:,j], y>, j < ncol ; V column-major with leading dimension ld
inline void adjoint_product(const double* V, Index ld, Index n, Index ncol, const double* y, double* res)
{
    for (Index j = 0; j < ncol; j++)
        res[j] = dot(V + size_t(j) * ld, y, n);
}
inline double max_abs(const double* x, Index n)
{
    double m = 0.0;
    for (Index i = 0; i < n; i++)
        m = std::max(m, std::fabs(x[i]));
    return m;
}

// ----------------------------------------------------------------------------
// LinAlg/Givens.h:22-86 StableScaling<double>::run (real overload)
// Given a >= b > 0: r = sqrt(a^2+b^2), c = a/r, s = b/r.
// ----------------------------------------------------------------------------
// Generalized problem A x = lambda B x, regular-inverse mode.
//   MatOp/SparseRegularInverse.h:55-127 — B held as a sparse matrix of which the Uplo triangle is used;
//     perform_op: y = selfadjointView<Uplo>(B) x (:122-127);  solve: y = B^{-1} x by Eigen::ConjugateGradient
//     with its defaults (:85-96): lower triangle, diagonal (Jacobi) preconditioner, tolerance = epsilon,
//     at most 2n iterations, start vector 0.
//   MatOp/internal/SymGEigsRegInvOp.h:76-81 — the Krylov operator y = B^{-1} (A x).
// [Eigen] ConjugateGradient is restated from Eigen 3.4.0's published algorithm
// (IterativeLinearSolvers/ConjugateGradient.h, conjugate_gradient()): r = b; stop when |r|^2 (updated by
// recurrence) < max(tol^2 |b|^2, min); p = M^{-1} r; each iteration  t = B p, alpha = (r.z)/(p.t), x += alpha p,
// r -= alpha t, z = M^{-1} r, beta = (r.z)_new/(r.z)_old, p = z + beta p.
// ----------------------------------------------------------------------------
struct RegularInverse
{
    const SparseSymCsc& B;
    const Index n;
    std::vector<double> invdiag;
    mutable Index last_iterations = 0;
    explicit RegularInverse(const SparseSymCsc& B_) : B(B_), n(B_.n), invdiag(B_.n, 1.0)
    {
        for (Index j = 0; j < n; j++)  // DiagonalPreconditioner: 1/diag, 1 where the diagonal entry is zero or absent
            for (int p = B.colptr[j]; p < B.colptr[j + 1]; p++)
                if (B.rowind[p] == j && B.val[p] != 0.0)
                    invdiag[j] = 1.0 / B.val[p];
    }
    // returns false if the iteration limit was hit before the tolerance
    bool solve(const double* rhs, double* x) const
    {
        std::vector<double> r(rhs, rhs + n), p(n), z(n), tmp(n);
        std::fill(x, x + n, 0.0);
        last_iterations = 0;
        const double rhs2 = dot(rhs, rhs, n);
        if (rhs2 == 0.0)
            return true;
        const double tol = kEps;
        const double threshold = std::max(tol * tol * rhs2, std::numeric_limits<double>::min());
        double r2 = dot(r.data(), r.data(), n);
        if (r2 < threshold)
            return true;
        for (Index i = 0; i < n; i++)
            p[i] = invdiag[i] * r[i];
        double abs_new = dot(r.data(), p.data(), n);
        const Index max_iters = 2 * n;
        Index it = 0;
        while (it < max_iters)
        {
            B.perform_op(p.data(), tmp.data());
            const double alpha = abs_new / dot(p.data(), tmp.data(), n);
            for (Index i = 0; i < n; i++)
                x[i] += alpha * p[i];
            for (Index i = 0; i < n; i++)
                r[i] -= alpha * tmp[i];
            r2 = dot(r.data(), r.data(), n);
            if (r2 < threshold)
                break;
            for (Index i = 0; i < n; i++)
                z[i] = invdiag[i] * r[i];
            const double abs_old = abs_new;
            abs_new = dot(r.data(), z.data(), n);
            const double beta = abs_new / abs_old;
            for (Index i = 0; i < n; i++)
                p[i] = z[i] + beta * p[i];
            it++;
        }
        last_iterations = it;
        return r2 < threshold;
    }
};

// SymGEigsRegInvOp.h:76-81
struct RegInvOp : Op
{
    const Op& A;
    const RegularInverse& Binv;
    mutable std::vector<double> cache;
    RegInvOp(const Op& A_, const RegularInverse& Binv_) : A(A_), Binv(Binv_), cache(Binv_.n) {}
    Index rows() const override { return Binv.n; }
    void perform_op(const double* x, double* y) const override
    {
        A.perform_op(x, cache.data());
        if (!Binv.solve(cache.data(), y))
            throw std::runtime_error("SparseRegularInverse: CG solver does not converge");  // SparseRegularInverse.h:113-114
    }
};

// ----------------------------------------------------------------------------
inline void stable_scaling(double a, double b, double& r, double& c, double& s)
{
    const double t = b / a;
    const double cutoff = 0.1 * std::pow(kEps, 0.25);  // :45  (~1.22e-5)
    if (t >= cutoff)
    {
        r = std::hypot(a, b);  // :49-51
        c = a / r;
        s = b / r;
    }
    else
    {
        // Taylor branch :67-76
        const double t2 = t * t;
        c = 1.0 - t2 * (0.5 - t2 * (0.375 - 0.3125 * t2));
        s = t * c;
        r = a + 0.5 * b * t * (1.0 - t2 * (0.25 - 0.125 * t2));
    }
}

// LinAlg/Givens.h:149-206 Givens<double>::compute_rotation
//   c*x - s*y = r,  s*x + c*y = 0,  i.e. c = x/r, s = -y/r
inline void givens_rotation(double x, double y, double& r, double& c, double& s)
{
    const double xsign = (x > 0.0) ? 1.0 : -1.0;
    const double xabs = std::fabs(x);
    if (y == 0.0)
    {
        c = (x == 0.0) ? 1.0 : xsign;  // :175-180
        s = 0.0;
        r = xabs;
        return;
    }
    const double ysign = (y > 0.0) ? 1.0 : -1.0;
    const double yabs = std::fabs(y);
    if (x == 0.0)
    {
        c = 0.0;  // :185-191
        s = -ysign;
        r = yabs;
        return;
    }
    if (xabs >= yabs)
    {
        stable_scaling(xabs, yabs, r, c, s);  // :194-199
        c = xsign * c;
        s = -ysign * s;
    }
    else
    {
        stable_scaling(yabs, xabs, r, s, c);  // :200-205 (roles of c and s swapped)
        c = xsign * c;
        s = -ysign * s;
    }
}

// ----------------------------------------------------------------------------
// LinAlg/UpperHessenbergQR.h:470-693  TridiagQR<double>
// (+ the inherited apply_YQ at :383-417)
// ----------------------------------------------------------------------------
class TridiagQR
{
public:
    Index n = 0;
    double shift = 0.0;
    std::vector<double> rot_cos, rot_sin;
    std::vector<double> T_diag, T_subd;           // saved T (after deflation)
    std::vector<double> R_diag, R_supd, R_supd2;  // R of T - shift*I = QR
    bool computed = false;

    // :515-598  `mat` is a full n x n column-major matrix; only diag and subdiag are read
    void compute(const Mat& mat, double s)
    {
        n = mat.rows;
        if (n != mat.cols)
            throw std::invalid_argument("TridiagQR: matrix must be square");
        shift = s;
        rot_cos.assign(n - 1, 0.0);
        rot_sin.assign(n - 1, 0.0);
        T_diag.resize(n);
        T_subd.resize(n - 1);
        for (Index i = 0; i < n; i++)
            T_diag[i] = mat(i, i);
        for (Index i = 0; i < n - 1; i++)
            T_subd[i] = mat(i + 1, i);

        // Deflation of small sub-diagonal elements :532-539
        for (Index i = 0; i < n - 1; i++)
            if (std::fabs(T_subd[i]) <= kEps * (std::fabs(T_diag[i]) + std::fabs(T_diag[i + 1])))
                T_subd[i] = 0.0;

        R_diag.resize(n);
        R_supd.resize(n - 1);
        R_supd2.assign(n > 2 ? n - 2 : 0, 0.0);
        for (Index i = 0; i < n; i++)
            R_diag[i] = T_diag[i] - shift;
        for (Index i = 0; i < n - 1; i++)
            R_supd[i] = T_subd[i];

        const Index n1 = n - 1, n2 = n - 2;
        for (Index i = 0; i < n1; i++)
        {
            double r, c, sn;
            givens_rotation(R_diag[i], T_subd[i], r, c, sn);  // :557
            rot_cos[i] = c;
            rot_sin[i] = sn;
            R_diag[i] = r;  // :568
            const double Tii1 = R_supd[i];
            const double Ti1i1 = R_diag[i + 1];
            R_supd[i] = c * Tii1 - sn * Ti1i1;      // :574
            R_diag[i + 1] = sn * Tii1 + c * Ti1i1;  // :575
            if (i < n2)
            {
                R_supd2[i] = -sn * R_supd[i + 1];  // :581
                R_supd[i + 1] *= c;                // :582
            }
        }
        computed = true;
    }

    // :604-615
    Mat matrix_R() const
    {
        Mat R(n, n);
        for (Index i = 0; i < n; i++)
            R(i, i) = R_diag[i];
        for (Index i = 0; i < n - 1; i++)
            R(i, i + 1) = R_supd[i];
        for (Index i = 0; i < n - 2; i++)
            R(i, i + 2) = R_supd2[i];
        return R;
    }

    // :627-693  dest <- Q'TQ computed from the saved T (not R*Q + s*I), re-deflated
    void matrix_QtHQ(Mat& dest) const
    {
        if (!computed)
            throw std::logic_error("TridiagQR: need to call compute() first");
        dest.resize(n, n);
        for (Index i = 0; i < n; i++)
            dest(i, i) = T_diag[i];
        for (Index i = 0; i < n - 1; i++)
            dest(i + 1, i) = T_subd[i];

        const Index n1 = n - 1, n2 = n - 2;
        for (Index i = 0; i < n1; i++)
        {
            const double c = rot_cos[i], s = rot_sin[i];
            const double cs = c * s, c2 = c * c, s2 = s * s;
            const double x = dest(i, i), y = dest(i + 1, i), z = dest(i + 1, i + 1);
            const double c2x = c2 * x, s2x = s2 * x, c2z = c2 * z, s2z = s2 * z;
            const double csy2 = 2.0 * c * s * y;
            dest(i, i) = c2x - csy2 + s2z;                  // x' :661
            dest(i + 1, i) = cs * (x - z) + (c2 - s2) * y;  // y' :662
            dest(i + 1, i + 1) = s2x + csy2 + c2z;          // z' :663
            if (i < n2)
            {
                const double ci1 = rot_cos[i + 1], si1 = rot_sin[i + 1];
                const double o = -s * T_subd[i + 1];          // :669
                dest(i + 2, i + 1) *= c;                      // :670
                dest(i + 1, i) = ci1 * dest(i + 1, i) - si1 * o;  // :671
            }
        }
        // Deflation :676-682
        for (Index i = 0; i < n1; i++)
        {
            const double diag = std::fabs(dest(i, i)) + std::fabs(dest(i + 1, i + 1));
            if (std::fabs(dest(i + 1, i)) <= kEps * diag)
                dest(i + 1, i) = 0.0;
        }
        for (Index i = 0; i < n1; i++)
            dest(i, i + 1) = dest(i + 1, i);  // :685
    }

    // :383-417  Y <- Y*Q = Y*G1*G2*...
    void apply_YQ(Mat& Y) const
    {
        if (!computed)
            throw std::logic_error("TridiagQR: need to call compute() first");
        const Index nrow = Y.rows;
        for (Index i = 0; i < n - 1; i++)
        {
            const double c = rot_cos[i], s = rot_sin[i];
            double* Yi = Y.col(i);
            double* Yi1 = Y.col(i + 1);
            for (Index j = 0; j < nrow; j++)
            {
                const double tmp = Yi[j];
                Yi[j] = c * tmp - s * Yi1[j];
                Yi1[j] = s * tmp + c * Yi1[j];
            }
        }
    }
};

// ----------------------------------------------------------------------------
// [Eigen] Eigen 3.4.0 Jacobi.h, JacobiRotation<double>::makeGivens (real branch)
// used by TridiagEigen.h:79-80, and numext::hypot (MathFunctionsImpl.h,
// positive_real_hypot) used at TridiagEigen.h:64.
// ----------------------------------------------------------------------------
inline void eigen_make_givens(double p, double q, double& c, double& s)
{
    if (q == 0.0)
    {
        c = p < 0.0 ? -1.0 : 1.0;
        s = 0.0;
    }
    else if (p == 0.0)
    {
        c = 0.0;
        s = q < 0.0 ? 1.0 : -1.0;
    }
    else if (std::fabs(p) > std::fabs(q))
    {
        const double t = q / p;
        double u = std::sqrt(1.0 + t * t);
        if (p < 0.0)
            u = -u;
        c = 1.0 / u;
        s = -t * c;
    }
    else
    {
        const double t = p / q;
        double u = std::sqrt(1.0 + t * t);
        if (q < 0.0)
            u = -u;
        s = -1.0 / u;
        c = -t * s;
    }
}
inline double eigen_hypot(double x, double y)
{
    x = std::fabs(x);
    y = std::fabs(y);
    if (std::isinf(x) || std::isinf(y))
        return INFINITY;
    if (std::isnan(x) || std::isnan(y))
        return NAN;
    const double p = std::max(x, y);
    if (p == 0.0)
        return 0.0;
    const double qp = std::min(y, x) / p;
    return p * std::sqrt(1.0 + qp * qp);
}

// ----------------------------------------------------------------------------
// LinAlg/TridiagEigen.h:24-230
// ----------------------------------------------------------------------------
class TridiagEigen
{
public:
    Index n = 0;
    std::vector<double> main_diag, sub_diag;
    Mat evecs;
    bool computed = false;

    // :44-108 implicit symmetric QR step with Wilkinson shift (adapted from Eigen)
    static void qr_step(double* diag, double* subdiag, Index start, Index end, double* Q, Index n)
    {
        double td = (diag[end - 1] - diag[end]) * 0.5;
        double e = subdiag[end - 1];
        double mu = diag[end];
        if (td == 0.0)
            mu -= std::fabs(e);
        else if (e != 0.0)
        {
            const double e2 = e * e;
            const double h = eigen_hypot(td, e);
            if (e2 == 0.0)
                mu -= e / ((td + (td > 0.0 ? h : -h)) / e);
            else
                mu -= e2 / (td + (td > 0.0 ? h : -h));
        }
        double x = diag[start] - mu;
        double z = subdiag[start];
        for (Index k = start; k < end && z != 0.0; ++k)
        {
            double c, s;
            eigen_make_givens(x, z, c, s);
            const double sdk = s * diag[k] + c * subdiag[k];
            const double dkp1 = s * subdiag[k] + c * diag[k + 1];
            diag[k] = c * (c * diag[k] - s * subdiag[k]) - s * (c * subdiag[k] - s * diag[k + 1]);
            diag[k + 1] = s * sdk + c * dkp1;
            subdiag[k] = c * sdk - s * dkp1;
            if (k > start)
                subdiag[k - 1] = c * subdiag[k - 1] - s * z;
            x = subdiag[k];
            if (k < end - 1)
            {
                z = -s * subdiag[k + 1];
                subdiag[k + 1] = c * subdiag[k + 1];
            }
            // Q <- Q * G : [Eigen] applyOnTheRight(k, k+1, rot): x' = c x - s y, y' = s x + c y
            if (Q)
            {
                double* qk = Q + size_t(k) * n;
                double* qk1 = Q + size_t(k + 1) * n;
                for (Index i = 0; i < n; i++)
                {
                    const double xi = qk[i], yi = qk1[i];
                    qk[i] = c * xi - s * yi;
                    qk1[i] = s * xi + c * yi;
                }
            }
        }
    }

    // :121-210
    void compute(const Mat& mat)
    {
        n = mat.rows;
        if (n != mat.cols)
            throw std::invalid_argument("TridiagEigen: matrix must be square");
        main_diag.assign(n, 0.0);
        sub_diag.assign(n > 0 ? n - 1 : 0, 0.0);
        evecs.resize(n, n);
        evecs.set_identity();

        double scale = 0.0;
        for (Index i = 0; i < n; i++)
            scale = std::max(scale, std::fabs(mat(i, i)));
        for (Index i = 0; i < n - 1; i++)
            scale = std::max(scale, std::fabs(mat(i + 1, i)));
        if (scale < kNear0)  // :142-150
        {
            computed = true;
            return;
        }
        for (Index i = 0; i < n; i++)
            main_diag[i] = mat(i, i) / scale;
        for (Index i = 0; i < n - 1; i++)
            sub_diag[i] = mat(i + 1, i) / scale;

        double* diag = main_diag.data();
        double* subdiag = sub_diag.data();
        Index end = n - 1, start = 0, iter = 0;
        int info = 0;
        const double considerAsZero = kMin;
        const double precision_inv = 1.0 / kEps;

        while (end > 0)
        {
            for (Index i = start; i < end; i++)
            {
                if (std::fabs(subdiag[i]) <= considerAsZero)
                    subdiag[i] = 0.0;
                else
                {
                    const double scaled = precision_inv * subdiag[i];
                    if (scaled * scaled <= (std::fabs(diag[i]) + std::fabs(diag[i + 1])))
                        subdiag[i] = 0.0;
                }
            }
            while (end > 0 && subdiag[end - 1] == 0.0)
                end--;
            if (end <= 0)
                break;
            iter++;
            if (iter > 30 * n)
            {
                info = 1;
                break;
            }
            start = end - 1;
            while (start > 0 && subdiag[start - 1] != 0.0)
                start--;
            qr_step(diag, subdiag, start, end, evecs.a.data(), n);
        }
        if (info > 0)
            throw std::runtime_error("TridiagEigen: eigen decomposition failed");
        for (Index i = 0; i < n; i++)
            main_diag[i] *= scale;
        computed = true;
    }
};

// ----------------------------------------------------------------------------
// Util/SelectionRule.h:62-287  argsort with the "sorting target" of each rule.
// std::sort on indices, like the reference (same libstdc++ => same tie order).
// ----------------------------------------------------------------------------
inline std::vector<Index> argsort(SortRule rule, const double* values, Index len)
{
    std::function<double(double)> target;
    switch (rule)
    {
        case SortRule::LargestMagn:
            target = [](double v) { return -std::fabs(v); };
            break;
        case SortRule::BothEnds:
        case SortRule::LargestAlge:
            target = [](double v) { return -v; };
            break;
        case SortRule::SmallestMagn:
            target = [](double v) { return std::fabs(v); };
            break;
        case SortRule::SmallestAlge:
            target = [](double v) { return v; };
            break;
        default:
            throw std::invalid_argument("unsupported selection rule");
    }
    std::vector<Index> ind(len);
    for (Index i = 0; i < len; i++)
        ind[i] = i;
    std::sort(ind.begin(), ind.end(),
              [&](Index i, Index j) { return target(values[i]) < target(values[j]); });
    if (rule == SortRule::BothEnds)  // :265-284: largest, smallest, 2nd largest, ...
    {
        std::vector<Index> copy(ind);
        for (Index i = 0; i < len; i++)
            ind[i] = (i % 2 == 0) ? copy[i / 2] : copy[len - 1 - i / 2];
    }
    return ind;
}

// ----------------------------------------------------------------------------
// LinAlg/Arnoldi.h:32-340 + LinAlg/Lanczos.h:28-217, merged for real symmetric A.
// `symmetric == true` follows Lanczos::factorize_from, false follows
// Arnoldi::factorize_from (general A, full H column every step).
// ----------------------------------------------------------------------------
class Factorization
{
public:
    const Op& op;
    const Index n, m;
    Index k = 0;
    Mat V, H;
    std::vector<double> f;
    double beta = 0.0;

    // Generalized problems (SymGEigsSolver.h:224-238, RegularInverse mode): inner products are taken in the B-inner
    // product, MatOp/internal/ArnoldiOp.h:68-101 — B*y is formed first (m_cache), then the plain dot / V' product.
    // bop == nullptr is the IdentityBOp specialisation (:113-162): plain dot, norm and V'y.
    const Op* bop = nullptr;
    mutable std::vector<double> bcache;

    // Test hook, unset by default: a NON-reference factorisation variant that replaces factorize_from_lanczos
    // (oracle/onesweep_variant.hpp restates this repository's opt-in one-sweep variant so that it can be compared with the
    // reference-faithful code below under the same driver).  nullptr = the reference's algorithm.
    void (*lanczos_variant)(Factorization&, Index, Index, Index&, void*) = nullptr;
    // ... and, for the same variant, a hook around compress_V (the variant may leave the last correction of a sweep pending and
    // let it ride on the restart).  nullptr = the reference's compress_V.
    void (*compress_variant)(Factorization&, const Mat&, void*) = nullptr;
    std::shared_ptr<void> variant_user;

    Factorization(const Op& op_, Index m_, const Op* bop_ = nullptr) : op(op_), n(op_.rows()), m(m_), bop(bop_) {}

    double ip(const double* x, const double* y) const  // ArnoldiOp::inner_product
    {
        if (!bop)
            return dot(x, y, n);
        bcache.resize(n);
        bop->perform_op(y, bcache.data());
        return dot(x, bcache.data(), n);
    }
    double nrm(const double* x) const { return std::sqrt(ip(x, x)); }  // ArnoldiOp::norm
    void adj(Index ncol, const double* y, double* res) const          // ArnoldiOp::adjoint_product
    {
        if (!bop)
        {
            adjoint_product(V.a.data(), n, n, ncol, y, res);
            return;
        }
        bcache.resize(n);
        bop->perform_op(y, bcache.data());
        adjoint_product(V.a.data(), n, n, ncol, bcache.data(), res);
    }

    // Arnoldi.h:66-115
    void expand_basis(Index ncol, Index seed, std::vector<double>& fv, double& fnorm, Index& op_counter)
    {
        std::vector<double> v(n), Vf(ncol);
        for (Index iter = 0; iter < 5; iter++)
        {
            SimpleRandom rng(static_cast<unsigned long>(seed + 123 * iter));
            if (iter == 0)
            {
                rng.fill(v.data(), n);
                op.perform_op(v.data(), fv.data());
                op_counter++;
            }
            else
                rng.fill(fv.data(), n);
            adj(ncol, fv.data(), Vf.data());
            for (Index j = 0; j < ncol; j++)  // f -= V * Vf
            {
                const double cj = Vf[j];
                const double* vj = V.col(j);
                for (Index i = 0; i < n; i++)
                    fv[i] -= vj[i] * cj;
            }
            fnorm = nrm(fv.data());
            adj(ncol, fv.data(), Vf.data());
            double ortho_err = max_abs(Vf.data(), ncol);
            int count = 0;
            while (count < 3 && ortho_err >= kEps * fnorm)
            {
                for (Index j = 0; j < ncol; j++)
                {
                    const double cj = Vf[j];
                    const double* vj = V.col(j);
                    for (Index i = 0; i < n; i++)
                        fv[i] -= vj[i] * cj;
                }
                fnorm = nrm(fv.data());
                adj(ncol, fv.data(), Vf.data());
                ortho_err = max_abs(Vf.data(), ncol);
                count++;
            }
            if (ortho_err < kEps * fnorm)
                return;
        }
    }

    // Arnoldi.h:136-195
    void init(const double* v0, Index& op_counter)
    {
        V.resize(n, m);
        H.resize(m, m);
        f.assign(n, 0.0);
        const double v0norm = nrm(v0);
        if (v0norm < kNear0)
            throw std::invalid_argument("initial residual vector cannot be zero");
        double* v = V.col(0);
        op.perform_op(v0, v);
        op_counter++;
        const double vnorm = nrm(v);
        if (vnorm < kNear0)
            for (Index i = 0; i < n; i++)
                v[i] = v0[i] / v0norm;  // :162-165
        else
            for (Index i = 0; i < n; i++)
                v[i] /= vnorm;  // :168
        std::vector<double> w(n);
        op.perform_op(v, w.data());
        op_counter++;
        H(0, 0) = ip(v, w.data());
        for (Index i = 0; i < n; i++)
            f[i] = w[i] - v[i] * H(0, 0);
        if (max_abs(f.data(), n) < kEps * std::fabs(H(0, 0)))  // :183-191
        {
            std::fill(f.begin(), f.end(), 0.0);
            beta = 0.0;
        }
        else
            beta = nrm(f.data());
        k = 1;
    }

    void zero_outside_leading(Index from_k)  // Lanczos.h:85-86 / Arnoldi.h:219-220
    {
        for (Index j = from_k; j < m; j++)
            for (Index i = 0; i < m; i++)
                H(i, j) = 0.0;
        for (Index j = 0; j < from_k; j++)
            for (Index i = from_k; i < m; i++)
                H(i, j) = 0.0;
    }

    void axpy_V(Index ncol, const double* c)  // f -= V[:, :ncol] * c
    {
        for (Index j = 0; j < ncol; j++)
        {
            const double cj = c[j];
            const double* vj = V.col(j);
            for (Index i = 0; i < n; i++)
                f[i] -= vj[i] * cj;
        }
    }

    // Lanczos.h:62-187
    void factorize_from_lanczos(Index from_k, Index to_m, Index& op_counter)
    {
        if (lanczos_variant)  // test hook (see the member): not the reference's algorithm
        {
            lanczos_variant(*this, from_k, to_m, op_counter, variant_user.get());
            return;
        }
        if (to_m <= from_k)
            return;
        if (from_k > k)
            throw std::invalid_argument("Lanczos: from_k (= " + std::to_string(from_k) +
                                        ") is larger than the current subspace dimension (= " + std::to_string(k) + ")");
        const double beta_thresh = kEps * std::sqrt(double(n));
        const double eps_sqrt = std::sqrt(kEps);
        std::vector<double> Vf(to_m), w(n);
        zero_outside_leading(from_k);

        for (Index i = from_k; i <= to_m - 1; i++)
        {
            bool restart = (beta < kNear0);  // :99
            double* v = V.col(i);
            if (!restart)
            {
                for (Index r = 0; r < n; r++)
                    v[r] = f[r] / beta;  // :106
                if (beta < eps_sqrt)
                {
                    const double Viv = ip(V.col(i - 1), v);  // :110
                    restart = (std::fabs(Viv) > eps_sqrt);
                }
            }
            if (restart)
            {
                expand_basis(i, 2 * i, f, beta, op_counter);  // :117-119
                for (Index r = 0; r < n; r++)
                    v[r] = f[r] / beta;
            }
            H(i, i - 1) = restart ? 0.0 : beta;  // :127
            H(i - 1, i) = H(i, i - 1);

            op.perform_op(v, w.data());  // :131
            op_counter++;

            if (!restart)  // :138-139
            {
                const double h = H(i, i - 1);
                const double* vp = V.col(i - 1);
                for (Index r = 0; r < n; r++)
                    w[r] -= h * vp[r];
            }
            H(i, i) = ip(v, w.data());  // :142
            const double hii = H(i, i);
            for (Index r = 0; r < n; r++)
                f[r] = w[r] - hii * v[r];  // :145
            beta = nrm(f.data());      // :146

            const Index i1 = i + 1;
            adj(i1, f.data(), Vf.data());  // :152
            double ortho_err = max_abs(Vf.data(), i1);
            int count = 0;
            while (count < 5 && ortho_err > kEps * beta)  // :156
            {
                if (beta < beta_thresh)  // :163-168
                {
                    std::fill(f.begin(), f.end(), 0.0);
                    beta = 0.0;
                    break;
                }
                axpy_V(i1, Vf.data());        // :171
                H(i - 1, i) += Vf[i - 1];     // :173-175
                H(i, i - 1) = H(i - 1, i);
                H(i, i) += Vf[i];
                beta = nrm(f.data());    // :177
                adj(i1, f.data(), Vf.data());
                ortho_err = max_abs(Vf.data(), i1);
                count++;
            }
        }
        k = to_m;
    }

    // Arnoldi.h:198-295
    void factorize_from_arnoldi(Index from_k, Index to_m, Index& op_counter)
    {
        if (to_m <= from_k)
            return;
        if (from_k > k)
            throw std::invalid_argument("Arnoldi: from_k (= " + std::to_string(from_k) +
                                        ") is larger than the current subspace dimension (= " + std::to_string(k) + ")");
        const double beta_thresh = kEps * std::sqrt(double(n));
        std::vector<double> Vf(to_m), w(n);
        zero_outside_leading(from_k);

        for (Index i = from_k; i <= to_m - 1; i++)
        {
            bool restart = false;
            if (beta < kNear0)  // :228-233
            {
                expand_basis(i, 2 * i, f, beta, op_counter);
                restart = true;
            }
            double* v = V.col(i);
            for (Index r = 0; r < n; r++)
                v[r] = f[r] / beta;             // :236
            H(i, i - 1) = restart ? 0.0 : beta;  // :239
            op.perform_op(v, w.data());          // :242
            op_counter++;

            const Index i1 = i + 1;
            double* h = &H(0, i);
            adjoint_product(V.a.data(), n, n, i1, w.data(), h);  // :251
            for (Index r = 0; r < n; r++)
                f[r] = w[r];
            axpy_V(i1, h);  // f = w - Vs*h :254
            beta = norm2(f.data(), n);

            if (beta > 0.717 * norm2(h, i1))  // :257
                continue;

            adjoint_product(V.a.data(), n, n, i1, f.data(), Vf.data());
            double ortho_err = max_abs(Vf.data(), i1);
            int count = 0;
            while (count < 5 && ortho_err > kEps * beta)
            {
                if (beta < beta_thresh)
                {
                    std::fill(f.begin(), f.end(), 0.0);
                    beta = 0.0;
                    break;
                }
                axpy_V(i1, Vf.data());
                for (Index j = 0; j < i1; j++)
                    h[j] += Vf[j];
                beta = norm2(f.data(), n);
                adjoint_product(V.a.data(), n, n, i1, f.data(), Vf.data());
                ortho_err = max_abs(Vf.data(), i1);
                count++;
            }
        }
        k = to_m;
    }

    // Lanczos.h:198-202
    void compress_H(const TridiagQR& decomp)
    {
        decomp.matrix_QtHQ(H);
        k--;
    }

    void compress_V(const Mat& Q)
    {
        if (compress_variant)  // test hook (see the member): not the reference's algorithm
            compress_variant(*this, Q, variant_user.get());
        else
            compress_V_reference(Q);
    }

    // Arnoldi.h:320-340.  Column i of Q has its first (m - k + i + 1) entries non-zero.
    void compress_V_reference(const Mat& Q)
    {
        Mat Vs(n, k + 1);
        for (Index i = 0; i < k; i++)
        {
            const Index nnz = m - k + i + 1;
            double* out = Vs.col(i);
            for (Index j = 0; j < nnz; j++)
            {
                const double q = Q(j, i);
                const double* vj = V.col(j);
                for (Index r = 0; r < n; r++)
                    out[r] += vj[r] * q;
            }
        }
        {
            double* out = Vs.col(k);
            for (Index j = 0; j < m; j++)
            {
                const double q = Q(j, k);
                const double* vj = V.col(j);
                for (Index r = 0; r < n; r++)
                    out[r] += vj[r] * q;
            }
        }
        for (Index i = 0; i <= k; i++)
            std::copy(Vs.col(i), Vs.col(i) + n, V.col(i));
        const double q = Q(m - 1, k - 1), h = H(k, k - 1);
        const double* vk = V.col(k);
        for (Index r = 0; r < n; r++)
            f[r] = f[r] * q + vk[r] * h;  // :337
        beta = nrm(f.data());
    }
};

// ----------------------------------------------------------------------------
// HermEigsBase.h:44-478 specialised to real symmetric (SymEigsSolver.h:133-160),
// with the SymEigsShiftSolver.h:163-169 eigenvalue back-transform as an option.
// ----------------------------------------------------------------------------
class SymEigs
{
public:
    const Op& op;
    const Index n, nev, ncv;
    Index nmatop = 0, niter = 0;
    Factorization fac;
    std::vector<double> ritz_val, ritz_est;
    Mat ritz_vec;
    std::vector<char> ritz_conv;
    CompInfo info = CompInfo::NotComputed;
    bool shift_invert = false;
    // back-transform of the Ritz values in the generalized shift modes (SymGEigsShiftSolver.h:59-65, :109-116, :160-167):
    // 0 none / shift_invert flag, 2 buckling lambda = sigma nu / (nu - 1), 3 Cayley lambda = sigma (nu + 1) / (nu - 1)
    int transform = 0;
    double sigma = 0.0;

    // bop_: the B operator of a generalized problem in regular-inverse mode (HermEigsBase<ModeMatOp, BOpType>,
    // SymGEigsSolver.h:224-238); nullptr = standard problem
    SymEigs(const Op& op_, Index nev_, Index ncv_, const Op* bop_ = nullptr) :
        op(op_), n(op_.rows()), nev(nev_), ncv(ncv_ > n ? n : ncv_), fac(op_, ncv_ > n ? n : ncv_, bop_)
    {
        // HermEigsBase.h:267-271
        if (nev_ < 1 || nev_ > n - 1)
            throw std::invalid_argument("nev must satisfy 1 <= nev <= n - 1, n is the size of matrix");
        if (ncv_ <= nev_ || ncv_ > n)
            throw std::invalid_argument("ncv must satisfy nev < ncv <= n, n is the size of matrix");
    }

    // :309-328
    void init(const double* init_resid)
    {
        ritz_val.assign(ncv, 0.0);
        ritz_vec.resize(ncv, nev);
        ritz_est.assign(ncv, 0.0);
        ritz_conv.assign(nev, 0);
        nmatop = 0;
        niter = 0;
        fac.init(init_resid, nmatop);
    }
    // :337-342
    void init()
    {
        SimpleRandom rng(0);
        std::vector<double> v0(n);
        rng.fill(v0.data(), n);
        init(v0.data());
    }

    // :205-224
    void retrieve_ritzpair(SortRule selection)
    {
        TridiagEigen decomp;
        decomp.compute(fac.H);
        const std::vector<double>& evals = decomp.main_diag;
        std::vector<Index> ind = argsort(selection, evals.data(), ncv);
        for (Index i = 0; i < ncv; i++)
        {
            ritz_val[i] = evals[ind[i]];
            ritz_est[i] = decomp.evecs(ncv - 1, ind[i]);
        }
        for (Index i = 0; i < nev; i++)
            for (Index r = 0; r < ncv; r++)
                ritz_vec(r, i) = decomp.evecs(r, ind[i]);
    }

    // :158-175
    Index num_converged(double tol)
    {
        const double eps23 = std::pow(kEps, 2.0 / 3.0);
        Index cnt = 0;
        for (Index i = 0; i < nev; i++)
        {
            const double thresh = tol * std::max(eps23, std::fabs(ritz_val[i]));
            const double resid = std::fabs(ritz_est[i]) * fac.beta;
            ritz_conv[i] = (resid < thresh);
            cnt += ritz_conv[i];
        }
        return cnt;
    }

    // :178-202
    Index nev_adjusted(Index nconv)
    {
        Index nev_new = nev;
        for (Index i = nev; i < ncv; i++)
            if (std::fabs(ritz_est[i]) < kNear0)
                nev_new++;
        nev_new += std::min(nconv, (ncv - nev_new) / 2);
        if (nev_new == 1 && ncv >= 6)
            nev_new = ncv / 2;
        else if (nev_new == 1 && ncv > 2)
            nev_new = 2;
        if (nev_new > ncv - 1)
            nev_new = ncv - 1;
        return nev_new;
    }

    // :105-155
    void restart(Index k, SortRule selection)
    {
        if (k >= ncv)
            return;
        TridiagQR decomp;
        Mat Q(ncv, ncv);
        Q.set_identity();
        const Index nshift = ncv - k;
        std::vector<double> shifts(ritz_val.end() - nshift, ritz_val.end());
        std::sort(shifts.begin(), shifts.end(),
                  [](const double& a, const double& b) { return std::fabs(a) > std::fabs(b); });
        for (Index i = 0; i < nshift; i++)
        {
            decomp.compute(fac.H, shifts[i]);
            decomp.apply_YQ(Q);
            fac.compress_H(decomp);
        }
        fac.compress_V(Q);
        fac.factorize_from_lanczos(k, ncv, nmatop);
        retrieve_ritzpair(selection);
    }

    // :229-251 (+ SymEigsShiftSolver.h:163-169 when shift_invert)
    void sort_ritzpair(SortRule sort_rule)
    {
        if (shift_invert)
            for (Index i = 0; i < nev; i++)
                ritz_val[i] = 1.0 / ritz_val[i] + sigma;
        else if (transform == 2)
            for (Index i = 0; i < nev; i++)
                ritz_val[i] = sigma * ritz_val[i] / (ritz_val[i] - 1.0);
        else if (transform == 3)
            for (Index i = 0; i < nev; i++)
                ritz_val[i] = sigma * (ritz_val[i] + 1.0) / (ritz_val[i] - 1.0);
        if (sort_rule != SortRule::LargestAlge && sort_rule != SortRule::LargestMagn &&
            sort_rule != SortRule::SmallestAlge && sort_rule != SortRule::SmallestMagn)
            throw std::invalid_argument("unsupported sorting rule");
        std::vector<Index> ind = argsort(sort_rule, ritz_val.data(), nev);
        std::vector<double> new_val(ncv, 0.0);
        Mat new_vec(ncv, nev);
        std::vector<char> new_conv(nev, 0);
        for (Index i = 0; i < nev; i++)
        {
            new_val[i] = ritz_val[ind[i]];
            for (Index r = 0; r < ncv; r++)
                new_vec(r, i) = ritz_vec(r, ind[i]);
            new_conv[i] = ritz_conv[ind[i]];
        }
        ritz_val.swap(new_val);
        ritz_vec = new_vec;
        ritz_conv.swap(new_conv);
    }

    // :366-390
    Index compute(SortRule selection = SortRule::LargestMagn, Index maxit = 1000, double tol = 1e-10,
                  SortRule sorting = SortRule::LargestAlge)
    {
        fac.factorize_from_lanczos(1, ncv, nmatop);
        retrieve_ritzpair(selection);
        Index i, nconv = 0, nev_adj;
        for (i = 0; i < maxit; i++)
        {
            nconv = num_converged(tol);
            if (nconv >= nev)
                break;
            nev_adj = nev_adjusted(nconv);
            restart(nev_adj, selection);
        }
        sort_ritzpair(sorting);
        niter += (i + 1);
        info = (nconv >= nev) ? CompInfo::Successful : CompInfo::NotConverging;
        return std::min(nev, nconv);
    }

    // :417-436
    std::vector<double> eigenvalues() const
    {
        std::vector<double> res;
        for (Index i = 0; i < nev; i++)
            if (ritz_conv[i])
                res.push_back(ritz_val[i]);
        return res;
    }

    // :447-470  returns n x nvec (column-major); nvec clipped to #converged
    Mat eigenvectors(Index nvec) const
    {
        Index nconv = 0;
        for (Index i = 0; i < nev; i++)
            nconv += ritz_conv[i];
        nvec = std::min(nvec, nconv);
        Mat res(n, nvec);
        if (!nvec)
            return res;
        Index j = 0;
        for (Index i = 0; i < nev && j < nvec; i++)
        {
            if (!ritz_conv[i])
                continue;
            double* out = res.col(j);
            for (Index c = 0; c < ncv; c++)
            {
                const double y = ritz_vec(c, i);
                const double* vc = fac.V.col(c);
                for (Index r = 0; r < n; r++)
                    out[r] += vc[r] * y;
            }
            j++;
        }
        return res;
    }
};

}  // namespace oracle
