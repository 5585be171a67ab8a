// ncv x ncv dense algorithms of the general (non-symmetric) restart, on raw column-major arrays
// (SURVEY.md §8a rows a18/a19).  Host code: these run once per restart on a matrix of a few dozen rows
// (≈1e5 flops) while every length-n operation stays on the GPU; the accumulated Q is then shipped to the
// device (ncv^2 doubles) for the V*Q kernel.
//
// Reference arithmetic followed (yixuan/spectra v1.2.0, include/Spectra/LinAlg/):
//   hess_shifted_qr        UpperHessenbergQR.h:136-195 (compute), :219-255 (matrix_QtHQ), :383-417 (apply_YQ)
//   double_shift_qr        DoubleShiftQR.h:51-231 (reflectors, update_block), :358-425 (compute), :455-467 (apply_YQ)
//   hess_real_schur        UpperHessenbergSchur.h:57-170, :287-340, :354-421
//   hess_eigen             UpperHessenbergEigen.h:53-218 (back-substitution), :231-285, :296-327
// Eigen-internal pieces (makeHouseholder, makeGivens with r, applyOnTheLeft/Right, normalize) are restated
// from Eigen 3.4.0's published source (Eigen is not available in this build environment).
#ifndef MISPEC_SPECTRA_SMALL_DENSE_GEN_H
#define MISPEC_SPECTRA_SMALL_DENSE_GEN_H

#include <algorithm>
#include <cmath>
#include <complex>
#include <stdexcept>
#include <vector>

#include "SmallDense.h"

namespace mispec {
namespace small {

// Column-major view helper
struct MatRef
{
    double* p;
    int ld;
    double& operator()(int i, int j) const { return p[(long) j * ld + i]; }
    double* col(int j) const { return p + (long) j * ld; }
};

// ---------------------------------------------------------------------------------------------------
// One real shift on an upper Hessenberg H: H - sI = QR, H <- RQ + sI, Qacc <- Qacc * Q.
// work: 2n doubles.  Entries below the sub-diagonal of H are treated as (and set to) zero.
// ---------------------------------------------------------------------------------------------------
inline void hess_shifted_qr(int n, double* Hp, int ldh, double shift, double* Qp, int ldq, int qrows, double* work)
{
    MatRef H{Hp, ldh}, Q{Qp, ldq};
    double* rc = work;
    double* rs = work + n;
    for (int i = 0; i < n; i++)
        H(i, i) -= shift;
    for (int i = 0; i < n - 1; i++)
    {
        for (int r = i + 2; r < n; r++)
            H(r, i) = 0.0;
        double r, c, s;
        givens_rotation(H(i, i), H(i + 1, i), r, c, s);
        rc[i] = c;
        rs[i] = s;
        H(i, i) = r;
        H(i + 1, i) = 0.0;
        for (int j = i + 1; j < n; j++)  // rows i, i+1 <- G' * rows
        {
            const double t = H(i, j);
            H(i, j) = c * t - s * H(i + 1, j);
            H(i + 1, j) = s * t + c * H(i + 1, j);
        }
    }
    for (int i = 0; i < n - 1; i++)  // R*Q and Qacc*Q: columns i, i+1 <- columns * G
    {
        const double c = rc[i], s = rs[i];
        double* a = H.col(i);
        double* b = H.col(i + 1);
        for (int j = 0; j < i + 2; j++)
        {
            const double t = a[j];
            a[j] = c * t - s * b[j];
            b[j] = s * t + c * b[j];
        }
        double* qa = Q.col(i);
        double* qb = Q.col(i + 1);
        for (int j = 0; j < qrows; j++)
        {
            const double t = qa[j];
            qa[j] = c * t - s * qb[j];
            qb[j] = s * t + c * qb[j];
        }
    }
    for (int i = 0; i < n; i++)
        H(i, i) += shift;
}

// ---------------------------------------------------------------------------------------------------
// Francis double shift: (H^2 - s H + t I) = QR implicitly; H <- Q'HQ, Qacc <- Qacc * Q.
// ---------------------------------------------------------------------------------------------------
class DoubleShiftStep
{
    int n;
    MatRef H;
    double s, t;
    std::vector<double> u;             // 3 x n reflector vectors
    std::vector<unsigned char> nrows;  // rows each reflector touches: 3 general, 2 Givens-like, 1 identity

    static double norm3(double a, double b, double c)
    {
        a = std::fabs(a);
        b = std::fabs(b);
        c = std::fabs(c);
        if (a < b)
            std::swap(a, b);
        if (a < c)
            std::swap(a, c);
        if (a < kNear0)
            return 0.0;
        const double r2 = b / a, r3 = c / a;
        const double cutoff = 0.1 * 1.220703125e-4;
        const double r = r2 * r2 + r3 * r3;
        return a * ((r2 >= cutoff || r3 >= cutoff) ? std::sqrt(1.0 + r) : (1.0 + r * (0.5 - 0.125 * r)));
    }
    // (x1,x2,x3) /= |x|, given |x1| largest and non-zero
    static void unit3(double& x1, double& x2, double& x3)
    {
        const double sgn = (x1 > 0.0) ? 1.0 : -1.0;
        x1 = std::fabs(x1);
        const double r2 = x2 / x1, r3 = x3 / x1;
        const double cutoff = 0.1 * 1.220703125e-4;
        double r = r2 * r2 + r3 * r3;
        r = (std::fabs(r2) >= cutoff || std::fabs(r3) >= cutoff) ? 1.0 / std::sqrt(1.0 + r) : (1.0 - r * (0.5 - 0.375 * r));
        x1 = sgn * r;
        x2 = r2 * r;
        x3 = r3 * r;
    }
    void reflector(double x1, double x2, double x3, int ind)
    {
        double* v = &u[3 * (size_t) ind];
        const double a2 = std::fabs(x2), a3 = std::fabs(x3);
        if (a2 < kNear0 && a3 < kNear0)
        {
            nrows[ind] = 1;
            return;
        }
        nrows[ind] = (a3 < kNear0) ? 2 : 3;
        const double nrm = (a3 < kNear0) ? eigen_hypot(x1, x2) : norm3(x1, x2, x3);
        const double rho = (x1 <= 0.0) ? 1.0 : -1.0;
        const double y1 = x1 - rho * nrm, a1 = std::fabs(y1);
        v[0] = y1;
        v[1] = x2;
        v[2] = x3;
        if (a1 >= a2 && a1 >= a3)
            unit3(v[0], v[1], v[2]);
        else if (a2 >= a1 && a2 >= a3)
            unit3(v[1], v[0], v[2]);
        else
            unit3(v[2], v[0], v[1]);
    }
    // rows r0.. (nrow of them) of columns [c0, c0+ncol): X <- X - 2 u (u'X)
    void left(int r0, int c0, int nrow, int ncol, int ind)
    {
        const int nr = nrows[ind];
        if (nr == 1)
            return;
        const double* v = &u[3 * (size_t) ind];
        const bool two = (nr == 2 || nrow == 2);
        for (int j = 0; j < ncol; j++)
        {
            double* x = &H(r0, c0 + j);
            const double d = 2.0 * v[0] * x[0] + 2.0 * v[1] * x[1] + (two ? 0.0 : 2.0 * v[2] * x[2]);
            x[0] -= d * v[0];
            x[1] -= d * v[1];
            if (!two)
                x[2] -= d * v[2];
        }
    }
    // rows [0, nrow) of columns c0.. (ncol of them) of M: X <- X - 2 (X u) u'
    void right(MatRef M, int c0, int nrow, int ncol, int ind) const
    {
        const int nr = nrows[ind];
        if (nr == 1)
            return;
        const double* v = &u[3 * (size_t) ind];
        const bool two = (nr == 2 || ncol == 2);
        double* x0 = M.col(c0);
        double* x1 = M.col(c0 + 1);
        double* x2 = two ? nullptr : M.col(c0 + 2);
        for (int i = 0; i < nrow; i++)
        {
            const double d = 2.0 * v[0] * x0[i] + 2.0 * v[1] * x1[i] + (two ? 0.0 : 2.0 * v[2] * x2[i]);
            x0[i] -= d * v[0];
            x1[i] -= d * v[1];
            if (!two)
                x2[i] -= d * v[2];
        }
    }
    void chase(int il, int iu)  // one unreduced diagonal block [il, iu]
    {
        const int bs = iu - il + 1;
        if (bs == 1)
        {
            nrows[il] = 1;
            return;
        }
        const double x00 = H(il, il), x01 = H(il, il + 1), x10 = H(il + 1, il), x11 = H(il + 1, il + 1);
        const double m00 = x00 * (x00 - s) + x01 * x10 + t;
        const double m10 = x10 * (x00 + x11 - s);
        if (bs == 2)
        {
            reflector(m00, m10, 0.0, il);
            left(il, il, 2, n - il, il);
            right(H, il, il + 2, 2, il);
            nrows[il + 1] = 1;
            return;
        }
        const double m20 = H(il + 2, il + 1) * H(il + 1, il);
        reflector(m00, m10, m20, il);
        left(il, il, 3, n - il, il);
        right(H, il, il + std::min(bs, 4), 3, il);
        for (int i = 1; i < bs - 2; i++)
        {
            const double* x = &H(il + i, il + i - 1);
            reflector(x[0], x[1], x[2], il + i);
            left(il + i, il + i - 1, 3, n - il - i + 1, il + i);
            right(H, il + i, il + std::min(bs, i + 4), 3, il + i);
        }
        reflector(H(iu - 1, iu - 2), H(iu, iu - 2), 0.0, iu - 1);
        left(iu - 1, iu - 2, 2, n - iu + 2, iu - 1);
        right(H, iu - 1, il + bs, 2, iu - 1);
        nrows[iu] = 1;
    }

public:
    // H is overwritten with Q'HQ; Qacc (qrows x n) with Qacc * Q.
    DoubleShiftStep(int n_, double* Hp, int ldh, double s_, double t_, double* Qp, int ldq, int qrows) :
        n(n_), H{Hp, ldh}, s(s_), t(t_), u(3 * (size_t) n_, 0.0), nrows((size_t) n_, 0)
    {
        const double eps_abs = kNear0 * (double(n) / kEps);
        auto deflate = [&](bool collect, std::vector<int>* cuts) {
            for (int i = 0; i < n - 1; i++)
            {
                const double h = std::fabs(H(i + 1, i));
                const double d = std::fabs(H(i, i)) + std::fabs(H(i + 1, i + 1));
                if (h <= eps_abs || h <= kEps * d)
                {
                    H(i + 1, i) = 0.0;
                    if (collect)
                        cuts->push_back(i + 1);
                }
                if (collect)
                    for (int r = i + 2; r < n; r++)
                        H(r, i) = 0.0;
            }
        };
        std::vector<int> cuts{0};
        deflate(true, &cuts);
        cuts.push_back(n);
        for (size_t b = 0; b + 1 < cuts.size(); b++)
            chase(cuts[b], cuts[b + 1] - 1);
        deflate(false, nullptr);
        // Qacc <- Qacc * P0 * P1 * ...
        MatRef Q{Qp, ldq};
        for (int i = 0; i < n - 2; i++)
            right(Q, i, qrows, 3, i);
        right(Q, n - 2, qrows, 2, n - 2);
    }
};

// ---------------------------------------------------------------------------------------------------
// Real Schur form of an upper Hessenberg matrix: H = U T U', T quasi-upper-triangular.
// T (n x n) in: H, out: T.  U out.  Returns false if the iteration limit (40 n) is hit.
// ---------------------------------------------------------------------------------------------------
inline void householder3(const double v[3], double ess[2], double& tau, double& beta)
{
    const double tail = v[1] * v[1] + v[2] * v[2];
    if (tail <= kMinPos)
    {
        tau = 0.0;
        beta = v[0];
        ess[0] = ess[1] = 0.0;
        return;
    }
    beta = std::sqrt(v[0] * v[0] + tail);
    if (v[0] >= 0.0)
        beta = -beta;
    ess[0] = v[1] / (v[0] - beta);
    ess[1] = v[2] / (v[0] - beta);
    tau = (beta - v[0]) / beta;
}
inline void make_givens_r(double p, double q, double& c, double& s, double& r)
{
    eigen_make_givens(p, q, c, s);
    if (q == 0.0)
        r = std::fabs(p);
    else if (p == 0.0)
        r = std::fabs(q);
    else if (std::fabs(p) > std::fabs(q))
    {
        const double t = q / p;
        double w = std::sqrt(1.0 + t * t);
        r = p * (p < 0.0 ? -w : w);
    }
    else
    {
        const double t = p / q;
        double w = std::sqrt(1.0 + t * t);
        r = q * (q < 0.0 ? -w : w);
    }
}

inline bool hess_real_schur(int n, double* Tp, int ldt, double* Up, int ldu)
{
    MatRef T{Tp, ldt}, U{Up, ldu};
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++)
            U(i, j) = (i == j) ? 1.0 : 0.0;
    auto rot_rows = [&](int p, int q, int c0, double c, double s) {
        for (int j = c0; j < n; j++)
        {
            const double x = T(p, j), y = T(q, j);
            T(p, j) = c * x - s * y;
            T(q, j) = s * x + c * y;
        }
    };
    auto rot_cols = [&](MatRef M, int p, int q, int nrow, double c, double s) {
        double* a = M.col(p);
        double* b = M.col(q);
        for (int i = 0; i < nrow; i++)
        {
            const double x = a[i], y = b[i];
            a[i] = c * x - s * y;
            b[i] = s * x + c * y;
        }
    };

    double norm = 0.0;
    for (int j = 0; j < n; j++)
        for (int i = 0; i < std::min(n, j + 2); i++)
            norm += std::fabs(T(i, j));
    if (norm == 0.0)
        return true;
    const double near0 = std::max(norm * kEps * kEps, kMinPos);
    const int max_iter = 40 * n;
    int iu = n - 1, iter = 0, total = 0;
    double exshift = 0.0;

    while (iu >= 0)
    {
        int il = iu;  // look for a negligible sub-diagonal entry
        while (il > 0)
        {
            const double sc = std::max((std::fabs(T(il - 1, il - 1)) + std::fabs(T(il, il))) * kEps, near0);
            if (std::fabs(T(il, il - 1)) <= sc)
                break;
            il--;
        }
        if (il == iu)  // one real eigenvalue split off
        {
            T(iu, iu) += exshift;
            if (iu > 0)
                T(iu, iu - 1) = 0.0;
            iu--;
            iter = 0;
        }
        else if (il == iu - 1)  // a 2 x 2 block split off
        {
            const double p = 0.5 * (T(iu - 1, iu - 1) - T(iu, iu));
            const double q = p * p + T(iu, iu - 1) * T(iu - 1, iu);
            T(iu, iu) += exshift;
            T(iu - 1, iu - 1) += exshift;
            if (q >= 0.0)  // two real eigenvalues: triangularise the block
            {
                const double z = std::sqrt(std::fabs(q));
                double c, s;
                eigen_make_givens((p >= 0.0) ? (p + z) : (p - z), T(iu, iu - 1), c, s);
                rot_rows(iu - 1, iu, iu - 1, c, s);
                rot_cols(T, iu - 1, iu, iu + 1, c, s);
                T(iu, iu - 1) = 0.0;
                rot_cols(U, iu - 1, iu, n, c, s);
            }
            if (iu > 1)
                T(iu - 1, iu - 2) = 0.0;
            iu -= 2;
            iter = 0;
        }
        else
        {
            // shift (with the two exceptional shifts at iterations 10 and 30)
            double sh[3] = {T(iu, iu), T(iu - 1, iu - 1), T(iu, iu - 1) * T(iu - 1, iu)};
            if (iter == 10)
            {
                exshift += sh[0];
                for (int i = 0; i <= iu; ++i)
                    T(i, i) -= sh[0];
                const double w = std::fabs(T(iu, iu - 1)) + std::fabs(T(iu - 1, iu - 2));
                sh[0] = sh[1] = 0.75 * w;
                sh[2] = -0.4375 * w * w;
            }
            if (iter == 30)
            {
                double w = (sh[1] - sh[0]) / 2.0;
                w = w * w + sh[2];
                if (w > 0.0)
                {
                    w = std::sqrt(w);
                    if (sh[1] < sh[0])
                        w = -w;
                    w = w + (sh[1] - sh[0]) / 2.0;
                    w = sh[0] - sh[2] / w;
                    exshift += w;
                    for (int i = 0; i <= iu; ++i)
                        T(i, i) -= w;
                    sh[0] = sh[1] = sh[2] = 0.964;
                }
            }
            iter++;
            if (++total > max_iter)
                return false;

            // where the Francis step starts, and its first Householder vector
            int im;
            double v[3] = {0.0, 0.0, 0.0};
            for (im = iu - 2; im >= il; --im)
            {
                const double tmm = T(im, im);
                const double r = sh[0] - tmm, ss = sh[1] - tmm;
                v[0] = (r * ss - sh[2]) / T(im + 1, im) + T(im, im + 1);
                v[1] = T(im + 1, im + 1) - tmm - r - ss;
                v[2] = T(im + 2, im + 1);
                if (im == il)
                    break;
                const double lhs = T(im, im - 1) * (std::fabs(v[1]) + std::fabs(v[2]));
                const double rhs = v[0] * (std::fabs(T(im - 1, im - 1)) + std::fabs(tmm) + std::fabs(T(im + 1, im + 1)));
                if (std::fabs(lhs) < kEps * rhs)
                    break;
            }
            // bulge chase
            for (int k = im; k <= iu - 2; ++k)
            {
                const bool first = (k == im);
                double w[3];
                if (first)
                {
                    w[0] = v[0];
                    w[1] = v[1];
                    w[2] = v[2];
                }
                else
                {
                    w[0] = T(k, k - 1);
                    w[1] = T(k + 1, k - 1);
                    w[2] = T(k + 2, k - 1);
                }
                double tau, beta, ess[2];
                householder3(w, ess, tau, beta);
                if (std::fabs(beta) > near0)
                {
                    if (first && k > il)
                        T(k, k - 1) = -T(k, k - 1);
                    else if (!first)
                        T(k, k - 1) = beta;
                    const double e1 = ess[0], e2 = ess[1];
                    for (int j = k; j < n; j++)
                    {
                        double* x = &T(k, j);
                        const double d = tau * (x[0] + e1 * x[1] + e2 * x[2]);
                        x[0] -= d;
                        x[1] -= d * e1;
                        x[2] -= d * e2;
                    }
                    const int nrT = std::min(iu, k + 3) + 1;
                    for (int pass = 0; pass < 2; pass++)
                    {
                        MatRef M = pass ? U : T;
                        const int rows = pass ? n : nrT;
                        double *x0 = M.col(k), *x1 = M.col(k + 1), *x2 = M.col(k + 2);
                        for (int i = 0; i < rows; i++)
                        {
                            const double d = tau * (x0[i] + e1 * x1[i] + e2 * x2[i]);
                            x0[i] -= d;
                            x1[i] -= d * e1;
                            x2[i] -= d * e2;
                        }
                    }
                }
            }
            double c, s, beta;
            make_givens_r(T(iu - 1, iu - 2), T(iu, iu - 2), c, s, beta);
            if (std::fabs(beta) > near0)
            {
                T(iu - 1, iu - 2) = beta;
                rot_rows(iu - 1, iu, iu - 1, c, s);
                rot_cols(T, iu - 1, iu, iu + 1, c, s);
                rot_cols(U, iu - 1, iu, n, c, s);
            }
            for (int i = im + 2; i <= iu; ++i)
            {
                T(i, i - 2) = 0.0;
                if (i > im + 2)
                    T(i, i - 3) = 0.0;
            }
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------
// Eigen-decomposition of a real upper Hessenberg matrix: eigenvalues (complex), eigenvectors
// (complex n x n column-major, unit 2-norm columns).  Throws std::runtime_error if the Schur
// iteration fails (like UpperHessenbergSchur.h:422).
// ---------------------------------------------------------------------------------------------------
inline void hess_eigen(int n, const double* Hin, int ldh, std::complex<double>* evals, std::complex<double>* evecs)
{
    using cd = std::complex<double>;
    std::vector<double> Tbuf((size_t) n * n), Ubuf((size_t) n * n);
    double scale = 0.0;
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++)
            scale = std::max(scale, std::fabs(Hin[(long) j * ldh + i]));
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++)
            Tbuf[(size_t) j * n + i] = Hin[(long) j * ldh + i] / scale;
    if (!hess_real_schur(n, Tbuf.data(), n, Ubuf.data(), n))
        throw std::runtime_error("UpperHessenbergSchur: Schur decomposition failed");
    MatRef T{Tbuf.data(), n}, U{Ubuf.data(), n};

    for (int i = 0; i < n;)
    {
        if (i == n - 1 || T(i + 1, i) == 0.0)
        {
            evals[i] = cd(T(i, i), 0.0);
            ++i;
        }
        else
        {
            const double p = 0.5 * (T(i, i) - T(i + 1, i + 1));
            double t0 = T(i + 1, i), t1 = T(i, i + 1);
            const double mx = std::max(std::fabs(p), std::max(std::fabs(t0), std::fabs(t1)));
            t0 /= mx;
            t1 /= mx;
            const double p0 = p / mx;
            const double z = mx * std::sqrt(std::fabs(p0 * p0 + t0 * t1));
            evals[i] = cd(T(i + 1, i + 1) + p, z);
            evals[i + 1] = cd(T(i + 1, i + 1) + p, -z);
            i += 2;
        }
    }

    // back-substitution on T for the eigenvectors of the quasi-triangular matrix
    double norm = 0.0;
    for (int j = 0; j < n; ++j)
        for (int c = std::max(j - 1, 0); c < n; c++)
            norm += std::fabs(T(j, c));
    if (norm != 0.0)
    {
        auto dotseg = [&](int i, int col, int l, int hi) {
            double r = 0.0;
            for (int k = l; k <= hi; k++)
                r += T(i, k) * T(k, col);
            return r;
        };
        for (int nn = n - 1; nn >= 0; nn--)
        {
            const double p = evals[nn].real(), q = evals[nn].imag();
            if (q == 0.0)
            {
                double lastr = 0.0, lastw = 0.0;
                int l = nn;
                T(nn, nn) = 1.0;
                for (int i = nn - 1; i >= 0; i--)
                {
                    const double w = T(i, i) - p;
                    const double r = dotseg(i, nn, l, nn);
                    if (evals[i].imag() < 0.0)
                    {
                        lastw = w;
                        lastr = r;
                        continue;
                    }
                    l = i;
                    if (evals[i].imag() == 0.0)
                        T(i, nn) = (w != 0.0) ? -r / w : -r / (kEps * norm);
                    else
                    {
                        const double x = T(i, i + 1), y = T(i + 1, i);
                        const double dr = evals[i].real() - p;
                        const double den = dr * dr + evals[i].imag() * evals[i].imag();
                        const double tt = (x * lastr - lastw * r) / den;
                        T(i, nn) = tt;
                        T(i + 1, nn) = (std::fabs(x) > std::fabs(lastw)) ? (-r - w * tt) / x : (-lastr - y * tt) / lastw;
                    }
                    const double big = std::fabs(T(i, nn));
                    if ((kEps * big) * big > 1.0)
                        for (int k = i; k < n; k++)
                            T(k, nn) /= big;
                }
            }
            else if (q < 0.0 && nn > 0)
            {
                double lastra = 0.0, lastsa = 0.0, lastw = 0.0;
                int l = nn - 1;
                if (std::fabs(T(nn, nn - 1)) > std::fabs(T(nn - 1, nn)))
                {
                    T(nn - 1, nn - 1) = q / T(nn, nn - 1);
                    T(nn - 1, nn) = -(T(nn, nn) - p) / T(nn, nn - 1);
                }
                else
                {
                    const cd cc = cd(0.0, -T(nn - 1, nn)) / cd(T(nn - 1, nn - 1) - p, q);
                    T(nn - 1, nn - 1) = cc.real();
                    T(nn - 1, nn) = cc.imag();
                }
                T(nn, nn - 1) = 0.0;
                T(nn, nn) = 1.0;
                for (int i = nn - 2; i >= 0; i--)
                {
                    const double ra = dotseg(i, nn - 1, l, nn);
                    const double sa = dotseg(i, nn, l, nn);
                    const double w = T(i, i) - p;
                    if (evals[i].imag() < 0.0)
                    {
                        lastw = w;
                        lastra = ra;
                        lastsa = sa;
                        continue;
                    }
                    l = i;
                    if (evals[i].imag() == 0.0)
                    {
                        const cd cc = cd(-ra, -sa) / cd(w, q);
                        T(i, nn - 1) = cc.real();
                        T(i, nn) = cc.imag();
                    }
                    else
                    {
                        const double x = T(i, i + 1), y = T(i + 1, i);
                        const double dr = evals[i].real() - p;
                        double vr = dr * dr + evals[i].imag() * evals[i].imag() - q * q;
                        const double vi = dr * 2.0 * q;
                        if (vr == 0.0 && vi == 0.0)
                            vr = kEps * norm * (std::fabs(w) + std::fabs(q) + std::fabs(x) + std::fabs(y) + std::fabs(lastw));
                        cd cc = cd(x * lastra - lastw * ra + q * sa, x * lastsa - lastw * sa - q * ra) / cd(vr, vi);
                        T(i, nn - 1) = cc.real();
                        T(i, nn) = cc.imag();
                        if (std::fabs(x) > (std::fabs(lastw) + std::fabs(q)))
                        {
                            T(i + 1, nn - 1) = (-ra - w * T(i, nn - 1) + q * T(i, nn)) / x;
                            T(i + 1, nn) = (-sa - w * T(i, nn) - q * T(i, nn - 1)) / x;
                        }
                        else
                        {
                            cc = cd(-lastra - y * T(i, nn - 1), -lastsa - y * T(i, nn)) / cd(lastw, q);
                            T(i + 1, nn - 1) = cc.real();
                            T(i + 1, nn) = cc.imag();
                        }
                    }
                    const double big = std::max(std::fabs(T(i, nn - 1)), std::fabs(T(i, nn)));
                    if ((kEps * big) * big > 1.0)
                        for (int k = i; k < n; k++)
                        {
                            T(k, nn - 1) /= big;
                            T(k, nn) /= big;
                        }
                }
                nn--;
            }
        }
        // back to the basis of H: column j of U <- U[:, :j+1] * T[:j+1, j]
        std::vector<double> tmp((size_t) n);
        for (int j = n - 1; j >= 0; j--)
        {
            std::fill(tmp.begin(), tmp.end(), 0.0);
            for (int c = 0; c <= j; c++)
            {
                const double tc = T(c, j);
                const double* uc = U.col(c);
                for (int r = 0; r < n; r++)
                    tmp[(size_t) r] += uc[r] * tc;
            }
            std::copy(tmp.begin(), tmp.end(), U.col(j));
        }
    }

    for (int i = 0; i < n; i++)
        evals[i] *= scale;
    if (!evecs)
        return;
    auto unit = [&](int j) {
        double z = 0.0;
        for (int i = 0; i < n; i++)
            z += std::norm(evecs[(size_t) j * n + i]);
        if (z > 0.0)
        {
            const double sc = std::sqrt(z);
            for (int i = 0; i < n; i++)
                evecs[(size_t) j * n + i] /= sc;
        }
    };
    for (int j = 0; j < n; ++j)
    {
        if (evals[j].imag() == 0.0 || j + 1 == n)
        {
            for (int i = 0; i < n; i++)
                evecs[(size_t) j * n + i] = cd(U(i, j), 0.0);
            unit(j);
        }
        else
        {
            for (int i = 0; i < n; ++i)
            {
                evecs[(size_t) j * n + i] = cd(U(i, j), U(i, j + 1));
                evecs[(size_t) (j + 1) * n + i] = cd(U(i, j), -U(i, j + 1));
            }
            unit(j);
            unit(j + 1);
            ++j;
        }
    }
}

}  // namespace small
}  // namespace mispec

#endif
