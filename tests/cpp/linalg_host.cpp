// Host-side LinAlg classes of the header API, checked the way the reference's own unit tests check theirs
// (/root/reference/test/QR.cpp:20-98, test/Eigen.cpp, test/Schur.cpp): Q orthogonal, R upper triangular,
// H - sI = QR, Q'HQ, every apply_* against the explicit product, eigen / Schur residuals — all to 1e-12.
// Plain C++11, no GPU, no library: g++ -std=c++11 -I include tests/cpp/linalg_host.cpp
#include <Spectra/LinAlg/BKLDLT.h>
#include <Spectra/LinAlg/DoubleShiftQR.h>
#include <Spectra/LinAlg/Givens.h>
#include <Spectra/LinAlg/Orthogonalization.h>
#include <Spectra/LinAlg/TridiagEigen.h>
#include <Spectra/LinAlg/UpperHessenbergEigen.h>
#include <Spectra/LinAlg/UpperHessenbergQR.h>
#include <Spectra/LinAlg/UpperHessenbergSchur.h>
#include <Spectra/Util/SelectionRule.h>
#include <Spectra/Util/Version.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <random>
#include <vector>

using namespace Spectra;
using Matrix = DenseMatrix<double>;
using Vector = DenseVector<double>;

static int failures = 0;
#define REQUIRE(cond)                                                        \
    do                                                                       \
    {                                                                        \
        if (!(cond))                                                         \
        {                                                                    \
            std::printf("REQUIRE failed at line %d: %s\n", __LINE__, #cond); \
            failures++;                                                      \
        }                                                                    \
    } while (0)

static std::mt19937 gen(123);
static double rnd() { return std::uniform_real_distribution<double>(-1.0, 1.0)(gen); }

static Matrix random_matrix(Index r, Index c)
{
    Matrix M(r, c);
    for (Index j = 0; j < c; j++)
        for (Index i = 0; i < r; i++)
            M(i, j) = rnd();
    return M;
}
static Matrix identity(Index n)
{
    Matrix I(n, n);
    for (Index j = 0; j < n; j++)
        for (Index i = 0; i < n; i++)
            I(i, j) = (i == j) ? 1.0 : 0.0;
    return I;
}
static Matrix mul(const Matrix& A, const Matrix& B, bool ta = false, bool tb = false)
{
    const Index m = ta ? A.cols() : A.rows(), k = ta ? A.rows() : A.cols(), n = tb ? B.rows() : B.cols();
    Matrix C(m, n);
    for (Index j = 0; j < n; j++)
        for (Index i = 0; i < m; i++)
        {
            double acc = 0.0;
            for (Index l = 0; l < k; l++)
                acc += (ta ? A(l, i) : A(i, l)) * (tb ? B(j, l) : B(l, j));
            C(i, j) = acc;
        }
    return C;
}
static double max_diff(const Matrix& A, const Matrix& B)
{
    double e = 0.0;
    for (Index j = 0; j < A.cols(); j++)
        for (Index i = 0; i < A.rows(); i++)
            e = std::fmax(e, std::fabs(A(i, j) - B(i, j)));
    return e;
}

// test/QR.cpp:20-98
template <typename Solver>
static void run_qr(const Matrix& H, double shift)
{
    const Index n = H.rows();
    const double tol = 1e-12;
    Solver decomp(H, shift);
    Matrix Hs = H;
    for (Index i = 0; i < n; i++)
        Hs(i, i) -= shift;
    const Matrix I = identity(n);
    Matrix Q = I;
    decomp.apply_QY(Q);
    REQUIRE(max_diff(mul(Q, Q, true, false), I) <= tol);
    REQUIRE(max_diff(mul(Q, Q, false, true), I) <= tol);
    const Matrix R = decomp.matrix_R();
    double lower = 0.0;
    for (Index j = 0; j < n; j++)
        for (Index i = j + 1; i < n; i++)
            lower = std::fmax(lower, std::fabs(R(i, j)));
    REQUIRE(lower <= tol);
    REQUIRE(max_diff(Hs, mul(Q, R)) <= tol);
    Matrix QtHQ;
    decomp.matrix_QtHQ(QtHQ);
    REQUIRE(max_diff(QtHQ, mul(mul(Q, H, true, false), Q)) <= tol);
    const Matrix Y = random_matrix(n, n);
    Matrix T = Y;
    decomp.apply_QY(T);
    REQUIRE(max_diff(T, mul(Q, Y)) <= tol);
    T = Y;
    decomp.apply_YQ(T);
    REQUIRE(max_diff(T, mul(Y, Q)) <= tol);
    T = Y;
    decomp.apply_QtY(T);
    REQUIRE(max_diff(T, mul(Q, Y, true, false)) <= tol);
    T = Y;
    decomp.apply_YQt(T);
    REQUIRE(max_diff(T, mul(Y, Q, false, true)) <= tol);
    Vector y(n), qy(n), qty(n);
    for (Index i = 0; i < n; i++)
        y[i] = rnd();
    qy = y;
    decomp.apply_QY(qy);
    qty = y;
    decomp.apply_QtY(qty);
    double e1 = 0.0, e2 = 0.0;
    for (Index i = 0; i < n; i++)
    {
        double a = 0.0, b = 0.0;
        for (Index l = 0; l < n; l++)
        {
            a += Q(i, l) * y[l];
            b += Q(l, i) * y[l];
        }
        e1 = std::fmax(e1, std::fabs(qy[i] - a));
        e2 = std::fmax(e2, std::fabs(qty[i] - b));
    }
    REQUIRE(e1 <= tol && e2 <= tol);
}

// ---- complex scalars (test/Givens.cpp:103-152, test/QR.cpp:176-189, test/Eigen.cpp:112-152) --------------------------------
using cd = std::complex<double>;
using CMatrix = DenseMatrix<cd>;
using CVector = DenseVector<cd>;

// op: 0 = as is, 1 = adjoint
static CMatrix cmul(const CMatrix& A, const CMatrix& B, int opa = 0, int opb = 0)
{
    const Index m = opa ? A.cols() : A.rows(), k = opa ? A.rows() : A.cols(), n = opb ? B.rows() : B.cols();
    CMatrix C(m, n);
    for (Index j = 0; j < n; j++)
        for (Index i = 0; i < m; i++)
        {
            cd acc(0.0);
            for (Index l = 0; l < k; l++)
                acc += (opa ? std::conj(A(l, i)) : A(i, l)) * (opb ? std::conj(B(j, l)) : B(l, j));
            C(i, j) = acc;
        }
    return C;
}
static double cmax_diff(const CMatrix& A, const CMatrix& B)
{
    double e = 0.0;
    for (Index j = 0; j < A.cols(); j++)
        for (Index i = 0; i < A.rows(); i++)
            e = std::fmax(e, std::abs(A(i, j) - B(i, j)));
    return e;
}
static CMatrix cidentity(Index n)
{
    CMatrix I(n, n);
    for (Index j = 0; j < n; j++)
        for (Index i = 0; i < n; i++)
            I(i, j) = (i == j) ? cd(1.0) : cd(0.0);
    return I;
}
static CMatrix crandom(Index r, Index c)
{
    CMatrix M(r, c);
    for (Index j = 0; j < c; j++)
        for (Index i = 0; i < r; i++)
            M(i, j) = cd(rnd(), rnd());
    return M;
}

static void run_complex_givens()
{
    // G = [c s; -conj(s) c]:  c x - s y = r,  conj(s) x + c y = 0, c real; zero and tiny-ratio operands included
    double r_err = 0.0, z_err = 0.0, u_err = 0.0;
    for (int it = 0; it < 20000; it++)
    {
        cd x(100.0 * rnd(), 100.0 * rnd()), y(100.0 * rnd(), 100.0 * rnd());
        if (it % 10 == 1) x = cd(0.0);
        if (it % 10 == 2) y = cd(0.0);
        if (it % 10 == 3) x = cd(x.real(), 0.0);
        if (it % 10 == 4) y = cd(0.0, y.imag());
        if (it % 10 == 5) y *= 1e-9;   // Taylor branch of the scaling
        if (it % 10 == 6) x *= 1e-9;
        if (it == 7) x = y = cd(0.0);
        cd r, s;
        double c;
        Givens<cd>::compute_rotation(x, y, r, c, s);
        const double scale = std::fmax(1.0, std::abs(x) + std::abs(y));
        r_err = std::fmax(r_err, std::abs(c * x - s * y - r) / scale);
        z_err = std::fmax(z_err, std::abs(std::conj(s) * x + c * y) / scale);
        u_err = std::fmax(u_err, std::fabs(c * c + std::norm(s) - 1.0));
    }
    REQUIRE(r_err <= 1e-14 && z_err <= 1e-14 && u_err <= 1e-14);
}

static void run_complex_qr(const CMatrix& H, cd shift)
{
    const Index n = H.rows();
    const double tol = 1e-12;
    UpperHessenbergQR<cd> decomp(H, shift);
    CMatrix Hs = H;
    for (Index i = 0; i < n; i++)
        Hs(i, i) -= shift;
    const CMatrix I = cidentity(n);
    CMatrix Q = I;
    decomp.apply_QY(Q);
    REQUIRE(cmax_diff(cmul(Q, Q, 1, 0), I) <= tol);
    REQUIRE(cmax_diff(cmul(Q, Q, 0, 1), I) <= tol);
    const CMatrix R = decomp.matrix_R();
    double low = 0.0;
    for (Index j = 0; j < n; j++)
        for (Index i = j + 1; i < n; i++)
            low = std::fmax(low, std::abs(R(i, j)));
    REQUIRE(low <= tol);
    REQUIRE(cmax_diff(Hs, cmul(Q, R)) <= tol);
    CMatrix QtHQ;
    decomp.matrix_QtHQ(QtHQ);
    REQUIRE(cmax_diff(QtHQ, cmul(cmul(Q, H, 1, 0), Q)) <= tol);
    const CMatrix Y = crandom(n, n);
    CMatrix W = Y;
    decomp.apply_QY(W);
    REQUIRE(cmax_diff(W, cmul(Q, Y)) <= tol);
    W = Y;
    decomp.apply_YQ(W);
    REQUIRE(cmax_diff(W, cmul(Y, Q)) <= tol);
    W = Y;
    decomp.apply_QtY(W);
    REQUIRE(cmax_diff(W, cmul(Q, Y, 1, 0)) <= tol);
    W = Y;
    decomp.apply_YQt(W);
    REQUIRE(cmax_diff(W, cmul(Y, Q, 0, 1)) <= tol);
    CVector y(n), qy(n), qty(n);
    for (Index i = 0; i < n; i++)
        qy[i] = qty[i] = y[i] = cd(rnd(), rnd());
    decomp.apply_QY(qy);
    decomp.apply_QtY(qty);
    double e1 = 0.0, e2 = 0.0;
    for (Index i = 0; i < n; i++)
    {
        cd a(0.0), b(0.0);
        for (Index l = 0; l < n; l++)
        {
            a += Q(i, l) * y[l];
            b += std::conj(Q(l, i)) * y[l];
        }
        e1 = std::fmax(e1, std::abs(qy[i] - a));
        e2 = std::fmax(e2, std::abs(qty[i] - b));
    }
    REQUIRE(e1 <= tol && e2 <= tol);
}

static void run_complex_eigen(const CMatrix& H)
{
    const Index n = H.rows();
    UpperHessenbergEigen<cd> eig(H);
    const CVector& ev = eig.eigenvalues();
    const CMatrix& X = eig.eigenvectors();
    double err = 0.0, unit = 0.0;
    for (Index j = 0; j < n; j++)
    {
        double nrm2 = 0.0;
        for (Index i = 0; i < n; i++)
        {
            cd acc(0.0);
            for (Index l = 0; l < n; l++)
                acc += H(i, l) * X(l, j);
            err = std::fmax(err, std::abs(acc - ev[j] * X(i, j)));
            nrm2 += std::norm(X(i, j));
        }
        unit = std::fmax(unit, std::fabs(nrm2 - 1.0));
        if (j > 0)
            REQUIRE(std::abs(ev[j - 1]) <= std::abs(ev[j]));  // increasing modulus (UpperHessenbergEigen.h:384-399)
    }
    REQUIRE(err <= 1e-12 && unit <= 1e-13);
}

int main()
{
    const Index n = 100;
    // upper Hessenberg and symmetric tridiagonal test matrices (test/QR.cpp:100-135)
    Matrix H = random_matrix(n, n);
    for (Index j = 0; j < n; j++)
        for (Index i = j + 2; i < n; i++)
            H(i, j) = 0.0;
    Matrix T(n, n);
    for (Index j = 0; j < n; j++)
        for (Index i = 0; i < n; i++)
            T(i, j) = 0.0;
    for (Index i = 0; i < n; i++)
        T(i, i) = rnd();
    for (Index i = 0; i + 1 < n; i++)
        T(i + 1, i) = T(i, i + 1) = rnd();
    run_qr<UpperHessenbergQR<double>>(H, 0.0);
    run_qr<UpperHessenbergQR<double>>(H, 1.2345);
    run_qr<UpperHessenbergQR<double>>(T, 0.6789);
    run_qr<TridiagQR<double>>(T, 0.0);
    run_qr<TridiagQR<double>>(T, 1.2345);

    // TridiagEigen: T X = X D, X orthogonal (test/Eigen.cpp:60-90)
    {
        TridiagEigen<double> eig(T);
        const Vector ev = eig.eigenvalues();
        const Matrix X = eig.eigenvectors();
        Matrix XD = X;
        for (Index j = 0; j < n; j++)
            for (Index i = 0; i < n; i++)
                XD(i, j) *= ev[j];
        REQUIRE(max_diff(mul(T, X), XD) <= 1e-12);
        REQUIRE(max_diff(mul(X, X, true, false), identity(n)) <= 1e-12);
    }
    // The eigen-decomposition restricted to ONE row of the eigenvector matrix (mispec_fac_ritz_values: the convergence test of an
    // iteration needs the last row only): `lanes` names the rows the rotations are applied to, so {m - 1, 1} must give the
    // eigenvalues and the last row of the full run bit for bit — at the solver's sizes and on a degenerate matrix.
    for (int m : {1, 2, 3, 20, 40, 64, 100})
        for (int kind = 0; kind < 2; kind++)
        {
            std::vector<double> d0(static_cast<std::size_t>(m)), e0(static_cast<std::size_t>(m), 0.0);
            for (int i = 0; i < m; i++)
            {
                d0[static_cast<std::size_t>(i)] = kind ? double(i % 3) : rnd();
                e0[static_cast<std::size_t>(i)] = (i + 1 < m) ? (kind ? ((i % 4) ? 1e-3 : 0.0) : rnd()) : 0.0;
            }
            std::vector<double> d1 = d0, e1 = e0, d2 = d0, e2 = e0;
            std::vector<double> Q1(static_cast<std::size_t>(m) * m, 0.0), Q2(static_cast<std::size_t>(m) * m, 0.0);
            for (int i = 0; i < m; i++)
                Q1[static_cast<std::size_t>(i) * m + i] = Q2[static_cast<std::size_t>(i) * m + i] = 1.0;
            const int rc1 = mispec::small::tridiag_eigen(m, d1.data(), e1.data(), Q1.data(), m, mispec::small::Lanes{0, 1});
            const int rc2 = mispec::small::tridiag_eigen(m, d2.data(), e2.data(), Q2.data(), m, mispec::small::Lanes{m - 1, 1});
            REQUIRE(rc1 == 0 && rc2 == 0);
            bool same = true;
            for (int j = 0; j < m; j++)
                same = same && d1[static_cast<std::size_t>(j)] == d2[static_cast<std::size_t>(j)] &&
                       Q1[static_cast<std::size_t>(j) * m + (m - 1)] == Q2[static_cast<std::size_t>(j) * m + (m - 1)];
            REQUIRE(same);
        }
    // DoubleShiftQR: Q orthogonal, Q'HQ Hessenberg and similar (DoubleShiftQR.h:400-467)
    {
        const double s = 0.3, t = 0.7;
        DoubleShiftQR<double> ds(H, s, t);
        Matrix Q = identity(n);
        ds.apply_YQ(Q);
        REQUIRE(max_diff(mul(Q, Q, true, false), identity(n)) <= 1e-12);
        Matrix QtHQ;
        ds.matrix_QtHQ(QtHQ);
        REQUIRE(max_diff(QtHQ, mul(mul(Q, H, true, false), Q)) <= 1e-11);
        Vector y(n), qty(n);
        for (Index i = 0; i < n; i++)
            y[i] = rnd();
        qty = y;
        ds.apply_QtY(qty);
        double e = 0.0;
        for (Index i = 0; i < n; i++)
        {
            double b = 0.0;
            for (Index l = 0; l < n; l++)
                b += Q(l, i) * y[l];
            e = std::fmax(e, std::fabs(qty[i] - b));
        }
        REQUIRE(e <= 1e-12);
    }
    // UpperHessenbergSchur: H = U T U', U orthogonal, T quasi upper triangular (test/Schur.cpp)
    {
        UpperHessenbergSchur<double> schur(H);
        const Matrix& Ts = schur.matrix_T();
        const Matrix& U = schur.matrix_U();
        REQUIRE(max_diff(mul(U, U, true, false), identity(n)) <= 1e-12);
        REQUIRE(max_diff(mul(mul(U, Ts), U, false, true), H) <= 1e-11);
        double low = 0.0;
        for (Index j = 0; j < n; j++)
            for (Index i = j + 2; i < n; i++)
                low = std::fmax(low, std::fabs(Ts(i, j)));
        REQUIRE(low == 0.0);
        Matrix Tt, Ut;
        UpperHessenbergSchur<double> again(H);
        again.swap_T(Tt);
        again.swap_U(Ut);
        REQUIRE(max_diff(Tt, Ts) == 0.0 && max_diff(Ut, U) == 0.0);
    }
    // UpperHessenbergEigen: H x = lambda x for complex pairs (test/Eigen.cpp:28-58)
    {
        UpperHessenbergEigen<double> eig(H);
        const auto ev = eig.eigenvalues();
        const auto X = eig.eigenvectors();
        double err = 0.0;
        for (Index j = 0; j < n; j++)
            for (Index i = 0; i < n; i++)
            {
                std::complex<double> acc(0.0, 0.0);
                for (Index l = 0; l < n; l++)
                    acc += H(i, l) * X(l, j);
                err = std::fmax(err, std::abs(acc - ev[j] * X(i, j)));
            }
        REQUIRE(err <= 1e-10);
    }
    {
        // LinAlg/BKLDLT.h — test/BKLDLT.cpp:14-74: (A - s I) x = b from either triangle, identical solutions, residual <= 1e-9
        const Index sizes[] = {1, 2, 3, 10, 100, 300};
        for (Index bn : sizes)
        {
            Matrix S = random_matrix(bn, bn);
            for (Index j = 0; j < bn; j++)
                for (Index i = 0; i < j; i++)
                {
                    const double sym = S(i, j) + S(j, i);
                    S(i, j) = sym;
                    S(j, i) = sym;
                }
            for (Index i = 0; i < bn; i++)
                S(i, i) *= 2.0;  // A = M + M'
            // the two triangles differ on purpose outside the one that is read
            Matrix L = S, U = S;
            for (Index j = 0; j < bn; j++)
                for (Index i = 0; i < j; i++)
                {
                    L(i, j) = 1e3;
                    U(j, i) = -1e3;
                }
            Vector b(bn);
            for (Index i = 0; i < bn; i++)
                b[i] = std::sin(1.0 + double(i));
            const double shift = 1.0;
            BKLDLT<double> dl(L, Lower, shift), du(U, Upper, shift);
            REQUIRE(dl.info() == CompInfo::Successful);
            REQUIRE(du.info() == CompInfo::Successful);
            const Vector xl = dl.solve(b), xu = du.solve(b);
            double diff = 0.0, resid = 0.0;
            for (Index i = 0; i < bn; i++)
            {
                diff = std::max(diff, std::fabs(xl[i] - xu[i]));
                double r = -shift * xl[i] - b[i];
                for (Index j = 0; j < bn; j++)
                    r += S(i, j) * xl[j];
                resid = std::max(resid, std::fabs(r));
            }
            REQUIRE(diff == 0.0);
            REQUIRE(resid < 1e-9);
        }
        // an indefinite matrix that needs 2x2 pivots (zero diagonal), and a singular one
        Matrix Z(4, 4);
        const double zv[16] = {0, 1, 2, 3, 1, 0, 4, 5, 2, 4, 0, 6, 3, 5, 6, 0};
        for (Index j = 0; j < 4; j++)
            for (Index i = 0; i < 4; i++)
                Z(i, j) = zv[j * 4 + i];
        BKLDLT<double> dz(Z);
        REQUIRE(dz.info() == CompInfo::Successful);
        Vector bz(4);
        for (Index i = 0; i < 4; i++)
            bz[i] = double(i) - 1.5;
        const Vector xz = dz.solve(bz);
        for (Index i = 0; i < 4; i++)
        {
            double r = -bz[i];
            for (Index j = 0; j < 4; j++)
                r += Z(i, j) * xz[j];
            REQUIRE(std::fabs(r) < 1e-12);
        }
        Matrix O4(4, 4);
        for (Index j = 0; j < 4; j++)
            for (Index i = 0; i < 4; i++)
                O4(i, j) = 0.0;
        BKLDLT<double> dsing(O4);
        REQUIRE(dsing.info() == CompInfo::NumericalIssue);
        bool threw = false;
        try
        {
            Vector t(4);
            dsing.solve_inplace(t);
        }
        catch (const std::logic_error&)
        {
            threw = true;
        }
        REQUIRE(threw);
    }
    {
        // LinAlg/Orthogonalization.h — test/Orthogonalization.cpp:13-99: basis' basis = I to 1e-12, complete (20 x 20) and
        // partial (15 orthonormal columns kept, 5 random ones appended)
        auto overlap_error = [](const Matrix& B) {
            double err = 0.0;
            for (Index a = 0; a < B.cols(); a++)
                for (Index b = 0; b < B.cols(); b++)
                {
                    double s = 0.0;
                    for (Index i = 0; i < B.rows(); i++)
                        s += B(i, a) * B(i, b);
                    err = std::max(err, std::fabs(s - (a == b ? 1.0 : 0.0)));
                }
            return err;
        };
        const Index on = 20, sub = 5, start = on - sub;
        typedef void (*OrthFn)(Matrix&, Index);
        const OrthFn fns[] = {&MGS_orthogonalisation<Matrix>, &GS_orthogonalisation<Matrix>, &twice_is_enough_orthogonalisation<Matrix>,
                              &JensWehner_orthogonalisation<Matrix>};
        for (OrthFn fn : fns)
        {
            Matrix full = random_matrix(on, on);
            fn(full, 0);
            REQUIRE(overlap_error(full) < 1e-12);
            Matrix part = random_matrix(on, on);
            Matrix head(on, start);
            for (Index j = 0; j < start; j++)
                for (Index i = 0; i < on; i++)
                    head(i, j) = part(i, j);
            QR_orthogonalisation(head);
            REQUIRE(overlap_error(head) < 1e-12);
            for (Index j = 0; j < start; j++)
                for (Index i = 0; i < on; i++)
                    part(i, j) = head(i, j);
            const Matrix before = part;
            fn(part, start);
            REQUIRE(overlap_error(part) < 1e-12);
            for (Index j = 0; j < start; j++)  // the leading block is left untouched
                for (Index i = 0; i < on; i++)
                    REQUIRE(part(i, j) == before(i, j));
        }
        Matrix q = random_matrix(on, on);
        QR_orthogonalisation(q);
        REQUIRE(overlap_error(q) < 1e-12);
        Matrix tall = random_matrix(50, 7);  // thin Q of a tall matrix spans the same space
        const Matrix tall0 = tall;
        QR_orthogonalisation(tall);
        REQUIRE(overlap_error(tall) < 1e-12);
        for (Index j = 0; j < 7; j++)  // tall0(:, j) lies in span(Q): || (I - QQ') a_j || ~ 0
        {
            std::vector<double> res(50);
            for (Index i = 0; i < 50; i++)
                res[i] = tall0(i, j);
            for (Index c = 0; c < 7; c++)
            {
                double s = 0.0;
                for (Index i = 0; i < 50; i++)
                    s += tall(i, c) * tall0(i, j);
                for (Index i = 0; i < 50; i++)
                    res[i] -= s * tall(i, c);
            }
            double nr = 0.0;
            for (double x : res)
                nr = std::max(nr, std::fabs(x));
            REQUIRE(nr < 1e-12);
        }
    }
    {
        // Util/SelectionRule.h: the compile-time sorter and argsort agree; BothEnds interleaves (reference :265-284)
        static_assert(SPECTRA_VERSION == 10200, "headers follow Spectra 1.2.0");
        const double v[6] = {0.5, -3.0, 2.0, -0.1, 4.0, 1.0};
        const std::vector<std::ptrdiff_t> lm = SortEigenvalue<double, SortRule::LargestMagn>(v, 6).index();
        REQUIRE(lm == argsort(SortRule::LargestMagn, v, 6));
        REQUIRE(lm[0] == 4 && lm[1] == 1 && lm[5] == 3);
        const std::vector<std::ptrdiff_t> be = argsort(SortRule::BothEnds, v, 6);
        REQUIRE(be[0] == 4 && be[1] == 1 && be[2] == 2 && be[3] == 3 && be[4] == 5 && be[5] == 0);
        std::vector<double> vec(v, v + 6);
        REQUIRE(argsort(SortRule::SmallestAlge, vec) == (SortEigenvalue<double, SortRule::SmallestAlge>(v, 6).index()));
    }
    {
        // the complex instantiations of the host-side classes
        run_complex_givens();
        CMatrix Hc = crandom(n, n);
        for (Index j = 0; j < n; j++)
            for (Index i = j + 2; i < n; i++)
                Hc(i, j) = cd(0.0);
        run_complex_qr(Hc, cd(1.2345, -5.4321));
        run_complex_qr(Hc, cd(0.0));
        run_complex_eigen(Hc);
        CMatrix Hz = Hc;  // a zero sub-diagonal entry and a repeated diagonal pair
        Hz(40, 39) = cd(0.0);
        Hz(11, 11) = Hz(10, 10);
        Hz(11, 10) = cd(0.0);
        run_complex_qr(Hz, cd(-0.5, 0.25));
        run_complex_eigen(Hz);
        CMatrix one(1, 1);
        one(0, 0) = cd(2.0, -1.0);
        run_complex_qr(one, cd(0.5));
        run_complex_eigen(one);
    }
    std::printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
    return failures ? 1 : 0;
}
