// Drop-in check of the header-only API: this file is written the way a Spectra user writes code
// (compare /root/reference/test/SymEigs.cpp:44-97 and the doc example SymEigsSolver.h:99-126) and is built
// with a plain host compiler against include/Spectra + libmispec.so:
//     g++ -std=c++17 -I include tests/cpp/dropin_symeigs.cpp -L spectra_amd -lmispec -Wl,-rpath,$PWD/spectra_amd
// It needs a GPU to run (tests/test_gpu_cpp_dropin.py).  Eigen is not available here, so matrices are handed
// over as Spectra::SparseView and results come back as Spectra::DenseVector / DenseMatrix.
#include <Spectra/DavidsonSymEigsSolver.h>
#include <Spectra/GenEigsComplexShiftSolver.h>
#include <Spectra/GenEigsRealShiftSolver.h>
#include <Spectra/GenEigsSolver.h>
#include <Spectra/MatOp/DenseCholesky.h>
#include <Spectra/MatOp/DenseGenComplexShiftSolve.h>
#include <Spectra/MatOp/DenseGenMatProd.h>
#include <Spectra/MatOp/DenseGenRealShiftSolve.h>
#include <Spectra/MatOp/DenseSymShiftSolve.h>
#include <Spectra/MatOp/DenseSymMatProd.h>
#include <Spectra/MatOp/SparseGenMatProd.h>
#include <Spectra/MatOp/SparseSymMatProd.h>
#include <Spectra/MatOp/SparseSymShiftSolve.h>
#include <Spectra/SymEigsShiftSolver.h>
#include <Spectra/SymEigsSolver.h>
#include <Spectra/SymGEigsShiftSolver.h>
#include <Spectra/SymGEigsSolver.h>
#include <Spectra/contrib/PartialSVDSolver.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <random>
#include <vector>

using namespace Spectra;

struct Csc
{
    int n;
    std::vector<int> colptr, rowind;
    std::vector<double> val;
    SparseView<double> view() const
    {
        SparseView<double> v;
        v.rows = v.cols = n;
        v.outer = colptr.data();
        v.inner = rowind.data();
        v.values = val.data();
        v.row_major = false;
        return v;
    }
};

// The reproducible fixture of test/SymEigs.cpp:25-42 (libstdc++ RNG), stored column-major like Eigen's default.
static Csc gen_sparse_data(int n, double prob)
{
    std::vector<std::vector<std::pair<int, double>>> cols(n);
    std::default_random_engine gen;
    gen.seed(0);
    std::uniform_real_distribution<double> distr(0.0, 1.0);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            if (distr(gen) < prob)
                cols[j].push_back({i, distr(gen) - 0.5});
    Csc A;
    A.n = n;
    A.colptr.push_back(0);
    for (int j = 0; j < n; j++)
    {
        for (auto& e : cols[j])
        {
            A.rowind.push_back(e.first);
            A.val.push_back(e.second);
        }
        A.colptr.push_back((int) A.rowind.size());
    }
    return A;
}

// ||A U - U D||_inf with A = selfadjointView<Lower> of the fixture
static double residual(const Csc& A, const DenseVector<double>& evals, const DenseMatrix<double>& U)
{
    double err = 0.0;
    std::vector<double> y(A.n);
    for (Index c = 0; c < U.cols(); c++)
    {
        std::fill(y.begin(), y.end(), 0.0);
        for (int j = 0; j < A.n; j++)
            for (int p = A.colptr[j]; p < A.colptr[j + 1]; p++)
            {
                const int i = A.rowind[p];
                if (i < j)
                    continue;
                y[i] += A.val[p] * U(j, c);
                if (i != j)
                    y[j] += A.val[p] * U(i, c);
            }
        for (int i = 0; i < A.n; i++)
            err = std::max(err, std::fabs(y[i] - evals[c] * U(i, c)));
    }
    return err;
}

static int failures = 0;
#define REQUIRE(cond)                                                      \
    do                                                                     \
    {                                                                      \
        if (!(cond))                                                       \
        {                                                                  \
            std::printf("REQUIRE failed at line %d: %s\n", __LINE__, #cond); \
            failures++;                                                    \
        }                                                                  \
    } while (0)

static void run_test_sets(const Csc& A, int k, int m)
{
    SparseSymMatProd<double> op(A.view());
    REQUIRE(op.rows() == A.n && op.cols() == A.n);
    const SortRule rules[] = {SortRule::LargestMagn, SortRule::LargestAlge, SortRule::SmallestMagn, SortRule::SmallestAlge,
                              SortRule::BothEnds};
    for (SortRule rule : rules)
    {
        SymEigsSolver<SparseSymMatProd<double>> eigs(op, k, m);
        eigs.init();
        const int nconv = (int) eigs.compute(rule);
        REQUIRE(eigs.info() == CompInfo::Successful);
        const auto evals = eigs.eigenvalues();
        const auto evecs = eigs.eigenvectors();
        const double err = residual(A, evals, evecs);
        std::printf("n=%d rule=%d nconv=%d niter=%d nops=%d ||AU-UD||_inf=%.3e\n", A.n, (int) rule, nconv,
                    (int) eigs.num_iterations(), (int) eigs.num_operations(), err);
        REQUIRE(nconv == k);
        REQUIRE(err < 1e-9);  // test/SymEigs.cpp:64
        // the opt-in one-sweep orthogonalisation through the header API: same eigenvalues, the reference's bar on the residual
        SymEigsSolver<SparseSymMatProd<double>> one(op, k, m);
        one.set_onesweep_orthogonalization(true);
        one.init();
        const int nconv1 = (int) one.compute(rule);
        REQUIRE(one.info() == CompInfo::Successful && nconv1 == k);
        const auto evals1 = one.eigenvalues();
        double dmax = 0.0;
        for (int i = 0; i < k; i++)
            dmax = std::max(dmax, std::fabs(evals1[i] - evals[i]));
        REQUIRE(dmax < 1e-9);
        REQUIRE(residual(A, evals1, one.eigenvectors()) < 1e-9);
    }
}

// A StorageIndex other than int (the reference's operators take any, MatOp/SparseSymMatProd.h:30): the same fixture with 64-bit
// index arrays through SparseSymMatProd / SparseGenMatProd / SparseSymShiftSolve — narrowed to the device's int32 at ingest; the
// solve must be the int one's bit for bit.
static void run_wide_storage_index(const Csc& A, int k, int m)
{
    const std::vector<long long> colptr(A.colptr.begin(), A.colptr.end()), rowind(A.rowind.begin(), A.rowind.end());
    SparseView<double, long long> v;
    v.rows = v.cols = A.n;
    v.outer = colptr.data();
    v.inner = rowind.data();
    v.values = A.val.data();
    v.row_major = false;
    SparseSymMatProd<double, Lower, ColMajor, long long> op64(v);
    SparseSymMatProd<double> op32(A.view());
    SymEigsSolver<SparseSymMatProd<double, Lower, ColMajor, long long>> e64(op64, k, m);
    SymEigsSolver<SparseSymMatProd<double>> e32(op32, k, m);
    e64.init();
    e32.init();
    REQUIRE(e64.compute(SortRule::LargestAlge) == k && e32.compute(SortRule::LargestAlge) == k);
    const auto a = e64.eigenvalues(), b = e32.eigenvalues();
    for (int i = 0; i < k; i++)
        REQUIRE(a[i] == b[i]);
    REQUIRE(e64.num_operations() == e32.num_operations());
    SparseGenMatProd<double, ColMajor, long long> g64(v);
    REQUIRE(g64.rows() == A.n && g64.cols() == A.n);
    std::vector<double> x(A.n, 1.0), y64(A.n), y32(A.n);
    g64.perform_op(x.data(), y64.data());
    SparseGenMatProd<double> g32(A.view());
    g32.perform_op(x.data(), y32.data());
    for (int i = 0; i < A.n; i++)
        REQUIRE(y64[i] == y32[i]);
    // an index beyond int32 is refused like a bad argument
    std::vector<long long> bad = rowind;
    if (!bad.empty())
        bad[0] = 3000000000LL;
    v.inner = bad.data();
    bool threw = false;
    try
    {
        SparseSymMatProd<double, Lower, ColMajor, long long> nope(v);
    }
    catch (const std::invalid_argument&)
    {
        threw = true;
    }
    REQUIRE(threw);
    std::printf("StorageIndex = long long: n=%d identical to the int solve\n", A.n);
}

// test/GenEigs.cpp:38-108 on the same fixture read as a general matrix (complex results)
static void run_gen_sets(const Csc& A, int k, int m)
{
    SparseGenMatProd<double> op(A.view());
    const SortRule rules[] = {SortRule::LargestMagn, SortRule::LargestReal, SortRule::LargestImag, SortRule::SmallestReal};
    for (SortRule rule : rules)
    {
        GenEigsSolver<SparseGenMatProd<double>> eigs(op, k, m);
        eigs.init();
        const int nconv = (int) eigs.compute(rule, 300);
        REQUIRE(eigs.info() == CompInfo::Successful);
        const auto evals = eigs.eigenvalues();
        const auto evecs = eigs.eigenvectors();
        double err = 0.0;
        std::vector<std::complex<double>> y(A.n);
        for (Index c = 0; c < evecs.cols(); c++)
        {
            std::fill(y.begin(), y.end(), std::complex<double>(0, 0));
            for (int j = 0; j < A.n; j++)
                for (int p = A.colptr[j]; p < A.colptr[j + 1]; p++)
                    y[A.rowind[p]] += A.val[p] * evecs(j, c);
            for (int i = 0; i < A.n; i++)
                err = std::max(err, std::abs(y[i] - evals[c] * evecs(i, c)));
        }
        std::printf("gen n=%d rule=%d nconv=%d nops=%d ||AU-UD||_inf=%.3e\n", A.n, (int) rule, nconv, (int) eigs.num_operations(), err);
        REQUIRE(nconv == k);
        REQUIRE(err < 1e-9);  // test/GenEigs.cpp:70
    }
}

// test/SymEigsShift.cpp: eigenvalues closest to sigma through (A - sigma I)^{-1}
static void run_shift(const Csc& A, int k, int m, double sigma)
{
    SparseSymShiftSolve<double> op(A.view());
    SymEigsShiftSolver<SparseSymShiftSolve<double>> eigs(op, k, m, sigma);
    eigs.init();
    const int nconv = (int) eigs.compute(SortRule::LargestMagn, 500);
    REQUIRE(eigs.info() == CompInfo::Successful);
    const auto evals = eigs.eigenvalues();
    const auto evecs = eigs.eigenvectors();
    const double err = residual(A, evals, evecs);
    std::printf("shift n=%d sigma=%g nconv=%d nops=%d ||AU-UD||_inf=%.3e\n", A.n, sigma, nconv, (int) eigs.num_operations(), err);
    REQUIRE(nconv == k);
    REQUIRE(err < 1e-9);  // test/SymEigsShift.cpp:76
}

// The documentation example: a user-supplied operator class (SymEigsSolver.h:99-126)
class MyDiagonalTen
{
public:
    using Scalar = double;
    int rows() const { return 10; }
    int cols() const { return 10; }
    void perform_op(const double* x_in, double* y_out) const
    {
        for (int i = 0; i < rows(); i++)
            y_out[i] = x_in[i] * (i + 1);
    }
};

// test/SymEigs.cpp:100-131 on a dense matrix (README.md:150-180 is this program): only the lower triangle of the
// column-major input is read.  The fixture is the sparse one scattered into a dense array, so `residual` applies.
static void run_dense(const Csc& A, int k, int m)
{
    DenseMatrix<double> M(A.n, A.n);
    for (int j = 0; j < A.n; j++)
    {
        for (int i = 0; i < A.n; i++)
            M(i, j) = 0.0;
        for (int p = A.colptr[j]; p < A.colptr[j + 1]; p++)
            M(A.rowind[p], j) += A.val[p];
    }
    DenseSymMatProd<double> op{DenseView<double>(M)};
    REQUIRE(op.rows() == A.n && op.cols() == A.n);
    REQUIRE(op(A.n - 1, 0) == M(A.n - 1, 0) && op(0, A.n - 1) == M(A.n - 1, 0));  // the mirrored lower triangle
    SymEigsSolver<DenseSymMatProd<double>> eigs(op, k, m);
    eigs.init();
    const int nconv = (int) eigs.compute(SortRule::LargestAlge);
    REQUIRE(eigs.info() == CompInfo::Successful);
    const double err = residual(A, eigs.eigenvalues(), eigs.eigenvectors());
    std::printf("dense n=%d nconv=%d nops=%d ||AU-UD||_inf=%.3e\n", A.n, nconv, (int) eigs.num_operations(), err);
    REQUIRE(nconv == k);
    REQUIRE(err < 1e-9);

    // general dense operator: the same matrix, all of it (test/GenEigs.cpp:110-146 shape)
    DenseGenMatProd<double> gop{DenseView<double>(M)};
    GenEigsSolver<DenseGenMatProd<double>> geigs(gop, k, m + 10);
    geigs.init();
    const int gconv = (int) geigs.compute(SortRule::LargestMagn);
    REQUIRE(geigs.info() == CompInfo::Successful);
    const auto ev = geigs.eigenvalues();
    const auto U = geigs.eigenvectors();
    double gerr = 0.0;
    for (Index c = 0; c < U.cols(); c++)
        for (int i = 0; i < A.n; i++)
        {
            std::complex<double> y = 0.0;
            for (int j = 0; j < A.n; j++)
                y += M(i, j) * U(j, c);
            gerr = std::max(gerr, std::abs(y - ev[c] * U(i, c)));
        }
    std::printf("dense-gen n=%d nconv=%d nops=%d ||AU-UD||_inf=%.3e\n", A.n, gconv, (int) geigs.num_operations(), gerr);
    REQUIRE(gconv >= k - 1);
    REQUIRE(gerr < 1e-9);
}

// The dense forms of the shift-and-invert and Cholesky operators (MatOp/DenseSymShiftSolve.h, DenseGenRealShiftSolve.h,
// DenseCholesky.h) on the sparse fixtures scattered into dense arrays — test/SymEigsShift.cpp:112-158,
// test/GenEigsRealShift.cpp:110-146 and test/SymGEigsCholesky.cpp:120-170 use dense matrices of these shapes.
static Csc gram_plus_ridge(const Csc& A);
static double pencil_residual(const Csc& P, const Csc& Q, const DenseVector<double>& evals, const DenseMatrix<double>& U);

static DenseMatrix<double> scatter(const Csc& A)
{
    DenseMatrix<double> M(A.n, A.n);
    for (int j = 0; j < A.n; j++)
    {
        for (int i = 0; i < A.n; i++)
            M(i, j) = 0.0;
        for (int p = A.colptr[j]; p < A.colptr[j + 1]; p++)
            M(A.rowind[p], j) += A.val[p];
    }
    return M;
}

static void run_dense_shift_and_cholesky(int n, double prob, int k, int m)
{
    const Csc A = gen_sparse_data(n, prob);
    const DenseMatrix<double> M = scatter(A);
    {
        DenseSymShiftSolve<double> op{DenseView<double>(M)};
        SymEigsShiftSolver<DenseSymShiftSolve<double>> eigs(op, k, m, 10.0);
        eigs.init();
        const int nconv = (int) eigs.compute(SortRule::LargestMagn, 500);
        REQUIRE(eigs.info() == CompInfo::Successful);
        const double err = residual(A, eigs.eigenvalues(), eigs.eigenvectors());
        std::printf("dense-shift n=%d nconv=%d nops=%d ||AU-UD||_inf=%.3e\n", n, nconv, (int) eigs.num_operations(), err);
        REQUIRE(nconv == k);
        REQUIRE(err < 1e-9);
    }
    {
        DenseGenRealShiftSolve<double> op{DenseView<double>(M)};
        GenEigsRealShiftSolver<DenseGenRealShiftSolve<double>> eigs(op, k, m + 10, 10.0);
        eigs.init();
        const int nconv = (int) eigs.compute(SortRule::LargestMagn, 500);
        REQUIRE(eigs.info() == CompInfo::Successful);
        const auto evals = eigs.eigenvalues();
        const auto U = eigs.eigenvectors();
        double err = 0.0;
        for (Index c = 0; c < U.cols(); c++)
            for (int i = 0; i < n; i++)
            {
                std::complex<double> y = 0.0;
                for (int j = 0; j < n; j++)
                    y += M(i, j) * U(j, c);
                err = std::fmax(err, std::abs(y - evals[c] * U(i, c)));
            }
        std::printf("dense-gen-realshift n=%d nconv=%d nops=%d ||AU-UD||_inf=%.3e\n", n, nconv, (int) eigs.num_operations(), err);
        REQUIRE(nconv >= k - 1);
        REQUIRE(err < 1e-8);
    }
    {
        // dense A and dense B = A'A + 0.1 I in Cholesky mode: the operator falls back to the host-pointer path of
        // SymGEigsCholeskyOp (three staged products per step), the Krylov basis stays in HBM
        const Csc B = gram_plus_ridge(A);
        const DenseMatrix<double> MB = scatter(B);
        DenseSymMatProd<double> op{DenseView<double>(M)};
        DenseCholesky<double> Bop{DenseView<double>(MB)};
        REQUIRE(Bop.info() == CompInfo::Successful);
        SymGEigsSolver<DenseSymMatProd<double>, DenseCholesky<double>, GEigsMode::Cholesky> eigs(op, Bop, k, m);
        eigs.init();
        const int nconv = (int) eigs.compute(SortRule::LargestAlge, 100);
        REQUIRE(eigs.info() == CompInfo::Successful);
        const double err = pencil_residual(A, B, eigs.eigenvalues(), eigs.eigenvectors());
        std::printf("dense-geigs-cholesky n=%d nconv=%d nops=%d ||AU-BUD||_inf=%.3e\n", n, nconv, (int) eigs.num_operations(), err);
        REQUIRE(nconv == k);
        REQUIRE(err < 1e-9);
    }
}

// test/GenEigsComplexShift.cpp:163-173: sparse 100 x 100, k = 10, m = 30, sigma = 20 + 10i; ||AU - UD||_inf < 1e-8
static void run_gen_complex_shift(int n, double prob, int k, int m, double sigmar, double sigmai)
{
    const Csc A = gen_sparse_data(n, prob);
    SparseGenComplexShiftSolve<double> op(A.view());
    const SortRule rules[] = {SortRule::LargestMagn, SortRule::LargestReal};
    for (SortRule rule : rules)
    {
        GenEigsComplexShiftSolver<SparseGenComplexShiftSolve<double>> eigs(op, k, m, sigmar, sigmai);
        eigs.init();
        const int nconv = (int) eigs.compute(rule, 500);
        REQUIRE(eigs.info() == CompInfo::Successful);
        const auto evals = eigs.eigenvalues();
        const auto U = eigs.eigenvectors();
        double err = 0.0;
        for (Index c = 0; c < U.cols(); c++)
        {
            std::vector<std::complex<double>> au(n, std::complex<double>(0.0, 0.0));
            for (int j = 0; j < n; j++)
                for (int p = A.colptr[j]; p < A.colptr[j + 1]; p++)
                    au[A.rowind[p]] += A.val[p] * U(j, c);
            for (int i = 0; i < n; i++)
                err = std::fmax(err, std::abs(au[i] - evals[c] * U(i, c)));
        }
        std::printf("gen-complexshift n=%d rule=%d nconv=%d nops=%d ||AU-UD||_inf=%.3e\n", n, (int) rule, nconv, (int) eigs.num_operations(), err);
        REQUIRE(nconv > 0);
        REQUIRE(err < 1e-8);
    }
}

// A user operator that works on DEVICE pointers: perform_op_device(x_dev, y_dev, stream) is the reference's
// perform_op contract with the vectors left in HBM.  Here it forwards to the library's SpMV on a device matrix.
class MyDeviceOperator
{
    SparseSymMatProd<double> m_mat;

public:
    using Scalar = double;
    mutable int calls = 0;
    explicit MyDeviceOperator(const Csc& A) : m_mat(A.view()) {}
    Index rows() const { return m_mat.rows(); }
    Index cols() const { return m_mat.cols(); }
    mispec_ctx* mispec_context() const { return m_mat.mispec_context(); }
    void perform_op_device(const double* x_dev, double* y_dev, void* /*hip_stream*/) const
    {
        calls++;
        internal::check(mispec_spmv(m_mat.mispec_matrix(), x_dev, y_dev));  // enqueued on the context's stream
    }
};

static void run_device_op(const Csc& A, int k, int m)
{
    MyDeviceOperator op(A);
    SymEigsSolver<MyDeviceOperator> eigs(op, k, m);
    eigs.init();
    const int nconv = (int) eigs.compute(SortRule::LargestMagn);
    REQUIRE(eigs.info() == CompInfo::Successful);
    const double err = residual(A, eigs.eigenvalues(), eigs.eigenvectors());
    std::printf("device-op n=%d nconv=%d nops=%d ||AU-UD||_inf=%.3e\n", A.n, nconv, (int) eigs.num_operations(), err);
    REQUIRE(nconv == k);
    REQUIRE(err < 1e-9);
    REQUIRE(op.calls == (int) eigs.num_operations());
}

// test/DavidsonSymEigs.cpp:46-123: the sparse fixture (diagonal i + 1, off-diagonal 0.1 (u - 0.5) with probability 0.5),
// nev = 10, largest and smallest eigenvalues, ||AU - UD||_inf < 1e-10
static Csc gen_davidson_sparse(int n)
{
    std::default_random_engine gen;
    gen.seed(0);
    std::uniform_real_distribution<double> distr(0.0, 1.0);
    std::vector<std::vector<std::pair<int, double>>> cols(n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
        {
            bool have = false;
            double v = 0.0;
            if (distr(gen) < 0.5)
            {
                v = 0.1 * (distr(gen) - 0.5);
                have = true;
            }
            if (i == j)
            {
                v = i + 1;
                have = true;
            }
            if (have)
                cols[j].push_back(std::make_pair(i, v));
        }
    Csc A;
    A.n = n;
    A.colptr.push_back(0);
    for (int j = 0; j < n; j++)
    {
        for (const auto& e : cols[j])
        {
            A.rowind.push_back(e.first);
            A.val.push_back(e.second);
        }
        A.colptr.push_back((int) A.rowind.size());
    }
    return A;
}

static void run_davidson(int n, int k)
{
    const Csc A = gen_davidson_sparse(n);
    SparseSymMatProd<double> op(A.view());
    const SortRule rules[] = {SortRule::LargestAlge, SortRule::SmallestAlge};
    for (SortRule rule : rules)
    {
        DavidsonSymEigsSolver<SparseSymMatProd<double>> eigs(op, k);
        const int nconv = (int) eigs.compute(rule);
        REQUIRE(eigs.info() == CompInfo::Successful);
        const double err = residual(A, eigs.eigenvalues(), eigs.eigenvectors());
        std::printf("davidson n=%d rule=%d nconv=%d niter=%d ||AU-UD||_inf=%.3e\n", n, (int) rule, nconv, (int) eigs.num_iterations(), err);
        REQUIRE(nconv == k);
        REQUIRE(err < 1e-10);  // test/DavidsonSymEigs.cpp:89
    }
}

// test/SymGEigsRegInv.cpp:35-106: A = sprand(n, prob) (lower triangle used), B = A'A + 0.1 I, regular-inverse mode;
// ||A U - B U D||_inf <= 1e-9 with the symmetric A the solver sees.
// B = A'A + 0.1 I as a full CSC matrix (assembled densely: n <= 1000)
static Csc gram_plus_ridge(const Csc& A)
{
    const int n = A.n;
    std::vector<double> Bd((size_t) n * n, 0.0);
    for (int j = 0; j < n; j++)
        for (int p = A.colptr[j]; p < A.colptr[j + 1]; p++)
            for (int jj = 0; jj < n; jj++)
                for (int q = A.colptr[jj]; q < A.colptr[jj + 1]; q++)
                    if (A.rowind[q] == A.rowind[p])
                        Bd[(size_t) jj * n + j] += A.val[p] * A.val[q];
    for (int i = 0; i < n; i++)
        Bd[(size_t) i * n + i] += 0.1;
    Csc B;
    B.n = n;
    B.colptr.push_back(0);
    for (int j = 0; j < n; j++)
    {
        for (int i = 0; i < n; i++)
            if (Bd[(size_t) j * n + i] != 0.0)
            {
                B.rowind.push_back(i);
                B.val.push_back(Bd[(size_t) j * n + i]);
            }
        B.colptr.push_back((int) B.rowind.size());
    }
    return B;
}

// max |selfadjointView<Lower>(P) U - selfadjointView<Lower>(Q) U D|
static double pencil_residual(const Csc& P, const Csc& Q, const DenseVector<double>& evals, const DenseMatrix<double>& U)
{
    const int n = P.n;
    double err = 0.0;
    for (Index c = 0; c < U.cols(); c++)
    {
        std::vector<double> pu(n, 0.0), qu(n, 0.0);
        const Csc* mats[2] = {&P, &Q};
        std::vector<double>* outs[2] = {&pu, &qu};
        for (int w = 0; w < 2; w++)
            for (int j = 0; j < n; j++)
                for (int p = mats[w]->colptr[j]; p < mats[w]->colptr[j + 1]; p++)
                {
                    const int i = mats[w]->rowind[p];
                    if (i < j)
                        continue;
                    (*outs[w])[i] += mats[w]->val[p] * U(j, c);
                    if (i != j)
                        (*outs[w])[j] += mats[w]->val[p] * U(i, c);
                }
        for (int i = 0; i < n; i++)
            err = std::fmax(err, std::fabs(pu[i] - evals[c] * qu[i]));
    }
    return err;
}

// test/SymGEigsShift.cpp sparse-sparse cases (:121-141 shift-invert, :214-234 buckling, :307-327 Cayley), sigma = 1.2345
template <GEigsMode Mode>
static void run_geigs_shift(const char* name)
{
    const int n = 100, k = 10, m = 20;
    const double sigma = 1.2345;
    const Csc A = gen_sparse_data(n, 0.1);
    const Csc B = gram_plus_ridge(A);
    const bool buckling = (Mode == GEigsMode::Buckling);
    const Csc& first = buckling ? B : A;   // buckling: the pencil is (K, KG) = (B, A) and the inner product is K
    const Csc& second = buckling ? A : B;
    using OpType = SymShiftInvert<double>;
    using BOpType = SparseSymMatProd<double>;
    OpType op(first.view(), second.view());
    BOpType Bop((buckling ? first : second).view());
    const SortRule rules[] = {SortRule::LargestMagn, SortRule::LargestAlge, SortRule::SmallestAlge, SortRule::BothEnds};
    for (SortRule rule : rules)
    {
        SymGEigsShiftSolver<OpType, BOpType, Mode> eigs(op, Bop, k, m, sigma);
        eigs.init();
        const int nconv = (int) eigs.compute(rule, 100);
        REQUIRE(eigs.info() == CompInfo::Successful);
        REQUIRE(nconv == k);
        const double err = pencil_residual(first, second, eigs.eigenvalues(), eigs.eigenvectors());
        std::printf("geigs-%s rule=%d nconv=%d nops=%d ||AU-BUD||_inf=%.3e\n", name, (int) rule, nconv, (int) eigs.num_operations(), err);
        REQUIRE(err < 1e-9);  // test/SymGEigsShift.cpp:91
    }
}

// test/GenEigsRealShift.cpp:46-105 on the sparse fixture: eigenvalues of a general matrix closest to sigma
static void run_gen_real_shift(int n, double prob, int k, int m, double sigma)
{
    const Csc A = gen_sparse_data(n, prob);
    SparseGenRealShiftSolve<double> op(A.view());
    const SortRule rules[] = {SortRule::LargestMagn, SortRule::LargestReal, SortRule::LargestImag, SortRule::SmallestReal};
    for (SortRule rule : rules)
    {
        GenEigsRealShiftSolver<SparseGenRealShiftSolve<double>> eigs(op, k, m, sigma);
        eigs.init();
        const int nconv = (int) eigs.compute(rule, 500);
        REQUIRE(eigs.info() == CompInfo::Successful);
        const auto evals = eigs.eigenvalues();
        const auto U = eigs.eigenvectors();
        double err = 0.0;
        for (Index c = 0; c < U.cols(); c++)
        {
            std::vector<std::complex<double>> au(n, std::complex<double>(0.0, 0.0));
            for (int j = 0; j < n; j++)
                for (int p = A.colptr[j]; p < A.colptr[j + 1]; p++)
                    au[A.rowind[p]] += A.val[p] * U(j, c);
            for (int i = 0; i < n; i++)
                err = std::fmax(err, std::abs(au[i] - evals[c] * U(i, c)));
        }
        std::printf("gen-realshift n=%d rule=%d nconv=%d nops=%d ||AU-UD||_inf=%.3e\n", n, (int) rule, nconv, (int) eigs.num_operations(), err);
        REQUIRE(nconv >= k - 1);
        REQUIRE(err < 1e-8);  // test/GenEigsRealShift.cpp:69
    }
}

// test/SymGEigsCholesky.cpp:44-133 on the sparse fixtures: Cholesky mode, eigenvectors back-transformed by L^{-T}
static void run_geigs_cholesky(int n, double prob, int k, int m)
{
    const Csc A = gen_sparse_data(n, prob);
    const Csc B = gram_plus_ridge(A);
    using OpType = SparseSymMatProd<double>;
    using BOpType = SparseCholesky<double>;
    OpType op(A.view());
    BOpType Bop(B.view());
    REQUIRE(Bop.info() == CompInfo::Successful);
    const SortRule rules[] = {SortRule::LargestMagn, SortRule::LargestAlge, SortRule::SmallestAlge, SortRule::BothEnds};
    for (SortRule rule : rules)
    {
        SymGEigsSolver<OpType, BOpType, GEigsMode::Cholesky> eigs(op, Bop, k, m);
        eigs.init();
        const int nconv = (int) eigs.compute(rule, 100);
        REQUIRE(eigs.info() == CompInfo::Successful);
        REQUIRE(nconv == k);
        const double err = pencil_residual(A, B, eigs.eigenvalues(), eigs.eigenvectors());
        std::printf("geigs-cholesky n=%d rule=%d nconv=%d nops=%d ||AU-BUD||_inf=%.3e\n", n, (int) rule, nconv,
                    (int) eigs.num_operations(), err);
        REQUIRE(err < 1e-9);
    }
}

static void run_geigs(int n, double prob, int k, int m)
{
    const Csc A = gen_sparse_data(n, prob);
    const Csc B = gram_plus_ridge(A);
    using OpType = SparseSymMatProd<double>;
    using BOpType = SparseRegularInverse<double>;
    OpType op(A.view());
    BOpType Bop(B.view());
    const SortRule rules[] = {SortRule::LargestMagn, SortRule::LargestAlge, SortRule::SmallestAlge, SortRule::BothEnds};
    for (SortRule rule : rules)
    {
        SymGEigsSolver<OpType, BOpType, GEigsMode::RegularInverse> eigs(op, Bop, k, m);
        eigs.init();
        const int nconv = (int) eigs.compute(rule, 100);
        REQUIRE(eigs.info() == CompInfo::Successful);
        REQUIRE(nconv == k);
        const auto evals = eigs.eigenvalues();
        const auto U = eigs.eigenvectors();
        double err = 0.0;
        for (int c = 0; c < nconv; c++)
        {
            std::vector<double> au(n, 0.0), bu(n, 0.0);
            for (int j = 0; j < n; j++)
            {
                for (int p = A.colptr[j]; p < A.colptr[j + 1]; p++)  // selfadjointView<Lower>(A)
                {
                    const int i = A.rowind[p];
                    if (i < j)
                        continue;
                    au[i] += A.val[p] * U(j, c);
                    if (i != j)
                        au[j] += A.val[p] * U(i, c);
                }
                for (int p = B.colptr[j]; p < B.colptr[j + 1]; p++)
                    bu[B.rowind[p]] += B.val[p] * U(j, c);
            }
            for (int i = 0; i < n; i++)
                err = std::fmax(err, std::fabs(au[i] - evals[c] * bu[i]));
        }
        std::printf("geigs n=%d rule=%d nconv=%d niter=%d nops=%d ||AU-BUD||_inf=%.3e\n", n, (int) rule, nconv,
                    (int) eigs.num_iterations(), (int) eigs.num_operations(), err);
        REQUIRE(err < 1e-9);  // test/SymGEigsRegInv.cpp:82
    }
}

// test/SVD.cpp:17-67: partial SVD of the rectangular sparse fixture; without Eigen's JacobiSVD the check is the
// singular-triplet residual  ||A v - s u||, ||A' u - s v|| <= 1e-9  and the descending order of s.
static void run_svd(int m, int n, int k, int ncv)
{
    std::vector<std::vector<std::pair<int, double>>> cols(n);
    std::default_random_engine gen;
    gen.seed(0);
    std::uniform_real_distribution<double> distr(0.0, 1.0);
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++)
            if (distr(gen) < 0.1)
                cols[j].push_back({i, distr(gen) - 0.5});
    std::vector<int> colptr{0}, rowind;
    std::vector<double> val;
    for (int j = 0; j < n; j++)
    {
        for (auto& e : cols[j])
        {
            rowind.push_back(e.first);
            val.push_back(e.second);
        }
        colptr.push_back((int) rowind.size());
    }
    SparseView<double> A;
    A.rows = m;
    A.cols = n;
    A.outer = colptr.data();
    A.inner = rowind.data();
    A.values = val.data();
    A.row_major = false;

    PartialSVDSolver<SparseView<double>> svds(A, k, ncv);
    const int nconv = (int) svds.compute();
    REQUIRE(nconv == k);
    const auto s = svds.singular_values();
    const auto U = svds.matrix_U(k);
    const auto V = svds.matrix_V(k);
    REQUIRE(U.rows() == m && U.cols() == k && V.rows() == n && V.cols() == k);
    double err = 0.0;
    for (int c = 0; c < k; c++)
    {
        std::vector<double> av(m, 0.0), atu(n, 0.0);
        for (int j = 0; j < n; j++)
            for (int p = colptr[j]; p < colptr[j + 1]; p++)
            {
                av[rowind[p]] += val[p] * V(j, c);
                atu[j] += val[p] * U(rowind[p], c);
            }
        for (int i = 0; i < m; i++)
            err = std::fmax(err, std::fabs(av[i] - s[c] * U(i, c)));
        for (int j = 0; j < n; j++)
            err = std::fmax(err, std::fabs(atu[j] - s[c] * V(j, c)));
        if (c > 0)
            REQUIRE(s[c] <= s[c - 1]);
    }
    std::printf("svd %dx%d k=%d nconv=%d s0=%.12f triplet residual=%.3e\n", m, n, k, nconv, s[0], err);
    REQUIRE(err < 1e-9);
}

int main()
{
    try
    {
        MyDiagonalTen op;
        SymEigsSolver<MyDiagonalTen> eigs(op, 3, 6);
        eigs.init();
        eigs.compute(SortRule::LargestAlge);
        REQUIRE(eigs.info() == CompInfo::Successful);
        const auto ev = eigs.eigenvalues();
        REQUIRE(ev.size() == 3);
        REQUIRE(std::fabs(ev[0] - 10.0) < 1e-10 && std::fabs(ev[1] - 9.0) < 1e-10 && std::fabs(ev[2] - 8.0) < 1e-10);
        std::printf("diag(1..10): %.12f %.12f %.12f\n", ev[0], ev[1], ev[2]);

        run_test_sets(gen_sparse_data(10, 0.5), 3, 6);      // test/SymEigs.cpp:133-143
        run_test_sets(gen_sparse_data(100, 0.1), 10, 20);   // :145-155
        run_test_sets(gen_sparse_data(1000, 0.01), 20, 50); // :157-167

        run_wide_storage_index(gen_sparse_data(100, 0.1), 10, 20);  // StorageIndex = long long (round 6)
        run_gen_sets(gen_sparse_data(100, 0.1), 10, 30);    // test/GenEigs.cpp:154-163
        run_gen_sets(gen_sparse_data(1000, 0.01), 20, 50);  // :165-174
        run_shift(gen_sparse_data(100, 0.1), 10, 20, 10.0);     // test/SymEigsShift.cpp:160-171
        run_shift(gen_sparse_data(1000, 0.01), 20, 50, 100.0);  // :173-185

        run_geigs(10, 0.5, 3, 6);      // test/SymGEigsRegInv.cpp:109-119
        run_geigs(100, 0.1, 10, 20);   // :121-131
        run_gen_real_shift(100, 0.1, 10, 30, 10.0);  // test/GenEigsRealShift.cpp:158-168
        run_geigs_cholesky(10, 0.5, 3, 6);     // test/SymGEigsCholesky.cpp:172-183
        run_geigs_cholesky(100, 0.1, 10, 20);  // :185-196
        run_geigs_shift<GEigsMode::ShiftInvert>("shiftinvert");
        run_geigs_shift<GEigsMode::Buckling>("buckling");
        run_geigs_shift<GEigsMode::Cayley>("cayley");
        run_svd(1000, 100, 5, 10);  // test/SVD.cpp:105-114 (tall sparse)
        run_svd(100, 1000, 5, 10);  // :116-125 (wide sparse)
        run_dense(gen_sparse_data(100, 0.1), 10, 20);        // test/SymEigs.cpp:111-120 shape, dense operators
        run_device_op(gen_sparse_data(1000, 0.01), 20, 50);  // user operator on device pointers
        run_davidson(1000, 10);                              // test/DavidsonSymEigs.cpp:116-122
        run_dense_shift_and_cholesky(100, 0.1, 10, 20);      // dense shift-and-invert / Cholesky operators
        run_gen_complex_shift(100, 0.1, 10, 30, 20.0, 10.0); // test/GenEigsComplexShift.cpp:163-173

        // constructor argument checks throw std::invalid_argument like the reference (HermEigsBase.h:267-271)
        bool threw = false;
        try
        {
            SymEigsSolver<MyDiagonalTen> bad(op, 3, 3);
        }
        catch (const std::invalid_argument&)
        {
            threw = true;
        }
        REQUIRE(threw);
    }
    catch (const std::exception& e)
    {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
    std::printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
    return failures ? 1 : 0;
}
