// =============================================================================
//  TEST INFRASTRUCTURE — NOT PRODUCT CODE.  (See spectra_oracle.hpp for scope.)
//
//  CPU restatement of the general (non-symmetric, real) path of yixuan/spectra
//  v1.2.0: implicitly-restarted Arnoldi with real / double shifts.
//  Citations are relative to /root/reference/include/Spectra/.
//  Pieces that live inside Eigen 3.4.0 (makeHouseholder, makeGivens with r,
//  apply{OnTheLeft,OnTheRight}, normalize(), std::complex division by
//  libstdc++) are restated from their published algorithms and marked [Eigen].
// =============================================================================
#pragma once

#include <complex>

#include "spectra_oracle.hpp"

namespace oracle {

using Complex = std::complex<double>;

// ----------------------------------------------------------------------------
// LinAlg/UpperHessenbergQR.h:45-460  UpperHessenbergQR<double> (real)
// ----------------------------------------------------------------------------
class UpperHessenbergQR
{
public:
    Index n = 0;
    double shift = 0.0;
    Mat R;
    std::vector<double> rot_cos, rot_sin;
    bool computed = false;

    // :136-195
    void compute(const Mat& mat, double s)
    {
        n = mat.rows;
        if (n != mat.cols)
            throw std::invalid_argument("UpperHessenbergQR: matrix must be square");
        shift = s;
        R = mat;
        rot_cos.assign(n - 1, 0.0);
        rot_sin.assign(n - 1, 0.0);
        for (Index i = 0; i < n; i++)
            R(i, i) -= shift;
        for (Index i = 0; i < n - 1; i++)
        {
            for (Index r = i + 2; r < n; r++)  // :155 zero below the sub-diagonal
                R(r, i) = 0.0;
            const double xi = R(i, i), xj = R(i + 1, i);
            double r, c, sn;
            givens_rotation(xi, xj, r, c, sn);
            rot_cos[i] = c;
            rot_sin[i] = sn;
            R(i, i) = r;
            R(i + 1, i) = 0.0;
            for (Index j = i + 1; j < n; j++)  // :175-180
            {
                const double tmp = R(i, j);
                R(i, j) = c * tmp - sn * R(i + 1, j);
                R(i + 1, j) = sn * tmp + c * R(i + 1, j);
            }
        }
        computed = true;
    }

    // :219-255  dest = R*Q + s*I
    void matrix_QtHQ(Mat& dest) const
    {
        if (!computed)
            throw std::logic_error("UpperHessenbergQR: need to call compute() first");
        dest = R;
        for (Index i = 0; i < n - 1; i++)
        {
            const double c = rot_cos[i], s = rot_sin[i];
            double* Yi = dest.col(i);
            double* Yi1 = dest.col(i + 1);
            for (Index j = 0; j < i + 2; j++)
            {
                const double tmp = Yi[j];
                Yi[j] = c * tmp - s * Yi1[j];
                Yi1[j] = s * tmp + c * Yi1[j];
            }
        }
        for (Index i = 0; i < n; i++)
            dest(i, i) += shift;
    }

    // :383-417
    void apply_YQ(Mat& Y) const
    {
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
// LinAlg/DoubleShiftQR.h:20-440
// ----------------------------------------------------------------------------
class DoubleShiftQR
{
public:
    Index n = 0;
    Mat H;
    double shift_s = 0.0, shift_t = 0.0;
    std::vector<double> ref_u;       // 3 x n, column-major
    std::vector<unsigned char> ref_nr;
    bool computed = false;

    double* u_col(Index ind) { return ref_u.data() + 3 * ind; }
    const double* u_col(Index ind) const { return ref_u.data() + 3 * ind; }

    // :51-80
    static double stable_norm3(double x1, double x2, double x3)
    {
        x1 = std::fabs(x1);
        x2 = std::fabs(x2);
        x3 = std::fabs(x3);
        if (x1 < x2)
            std::swap(x1, x2);
        if (x1 < x3)
            std::swap(x1, x3);
        if (x1 < kNear0)
            return 0.0;
        const double r2 = x2 / x1, r3 = x3 / x1;
        const double cutoff = 0.1 * std::pow(kEps, 0.25);
        double r = r2 * r2 + r3 * r3;
        r = (r2 >= cutoff || r3 >= cutoff) ? std::sqrt(1.0 + r) : (1.0 + r * (0.5 - 0.125 * r));
        return x1 * r;
    }

    // :84-105
    static void stable_scaling(double& x1, double& x2, double& x3)
    {
        const double x1sign = (x1 > 0.0) ? 1.0 : -1.0;
        x1 = std::fabs(x1);
        const double r2 = x2 / x1, r3 = x3 / x1;
        const double cutoff = 0.1 * std::pow(kEps, 0.25);
        double r = r2 * r2 + r3 * r3;
        r = (std::fabs(r2) >= cutoff || std::fabs(r3) >= cutoff) ? 1.0 / std::sqrt(1.0 + r) : (1.0 - r * (0.5 - 0.375 * r));
        x1 = x1sign * r;
        x2 = r2 * r;
        x3 = r3 * r;
    }

    // :107-146
    void compute_reflector(double x1, double x2, double x3, Index ind)
    {
        double* u = u_col(ind);
        const double x2m = std::fabs(x2), x3m = std::fabs(x3);
        if (x2m < kNear0 && x3m < kNear0)
        {
            ref_nr[ind] = 1;
            return;
        }
        ref_nr[ind] = (x3m < kNear0) ? 2 : 3;
        const double x_norm = (x3m < kNear0) ? eigen_hypot(x1, x2) : stable_norm3(x1, x2, x3);
        const double rho = double(x1 <= 0.0) - double(x1 > 0.0);
        const double x1_new = x1 - rho * x_norm, x1m = std::fabs(x1_new);
        u[0] = x1_new;
        u[1] = x2;
        u[2] = x3;
        if (x1m >= x2m && x1m >= x3m)
            stable_scaling(u[0], u[1], u[2]);
        else if (x2m >= x1m && x2m >= x3m)
            stable_scaling(u[1], u[0], u[2]);
        else
            stable_scaling(u[2], u[0], u[1]);
    }

    // :236-271  X = block of H starting at (r0, c0), nrow x ncol ; PX = X - 2 u (u'X)
    void apply_PX(Index r0, Index c0, Index nrow, Index ncol, Index u_ind)
    {
        const Index nr = ref_nr[u_ind];
        if (nr == 1)
            return;
        const double* u = u_col(u_ind);
        const double u0 = u[0], u1 = u[1];
        const double u0_2 = 2.0 * u0, u1_2 = 2.0 * u1;
        if (nr == 2 || nrow == 2)
        {
            for (Index j = 0; j < ncol; j++)
            {
                double* x = &H(r0, c0 + j);
                const double tmp = u0_2 * x[0] + u1_2 * x[1];
                x[0] -= tmp * u0;
                x[1] -= tmp * u1;
            }
        }
        else
        {
            const double u2 = u[2], u2_2 = 2.0 * u2;
            for (Index j = 0; j < ncol; j++)
            {
                double* x = &H(r0, c0 + j);
                const double tmp = u0_2 * x[0] + u1_2 * x[1] + u2_2 * x[2];
                x[0] -= tmp * u0;
                x[1] -= tmp * u1;
                x[2] -= tmp * u2;
            }
        }
    }

    // :295-333  X = block of M starting at (r0, c0), nrow x ncol ; XP = X - 2 (X u) u'
    void apply_XP(Mat& M, Index r0, Index c0, Index nrow, Index ncol, Index u_ind) const
    {
        const Index nr = ref_nr[u_ind];
        if (nr == 1)
            return;
        const double* u = u_col(u_ind);
        const double u0 = u[0], u1 = u[1];
        const double u0_2 = 2.0 * u0, u1_2 = 2.0 * u1;
        double* X0 = &M(r0, c0);
        double* X1 = &M(r0, c0 + 1);
        if (nr == 2 || ncol == 2)
        {
            for (Index i = 0; i < nrow; i++)
            {
                const double tmp = u0_2 * X0[i] + u1_2 * X1[i];
                X0[i] -= tmp * u0;
                X1[i] -= tmp * u1;
            }
        }
        else
        {
            double* X2 = &M(r0, c0 + 2);
            const double u2 = u[2], u2_2 = 2.0 * u2;
            for (Index i = 0; i < nrow; i++)
            {
                const double tmp = u0_2 * X0[i] + u1_2 * X1[i] + u2_2 * X2[i];
                X0[i] -= tmp * u0;
                X1[i] -= tmp * u1;
                X2[i] -= tmp * u2;
            }
        }
    }

    // :154-231
    void update_block(Index il, Index iu)
    {
        const Index bsize = iu - il + 1;
        if (bsize == 1)
        {
            ref_nr[il] = 1;
            return;
        }
        const double x00 = H(il, il), x01 = H(il, il + 1), x10 = H(il + 1, il), x11 = H(il + 1, il + 1);
        const double m00 = x00 * (x00 - shift_s) + x01 * x10 + shift_t;
        const double m10 = x10 * (x00 + x11 - shift_s);
        if (bsize == 2)
        {
            compute_reflector(m00, m10, 0.0, il);
            apply_PX(il, il, 2, n - il, il);
            apply_XP(H, 0, il, il + 2, 2, il);
            ref_nr[il + 1] = 1;
            return;
        }
        const double m20 = H(il + 2, il + 1) * H(il + 1, il);
        compute_reflector(m00, m10, m20, il);
        apply_PX(il, il, 3, n - il, il);
        apply_XP(H, 0, il, il + std::min<Index>(bsize, 4), 3, il);
        for (Index i = 1; i < bsize - 2; i++)
        {
            const double* x = &H(il + i, il + i - 1);
            compute_reflector(x[0], x[1], x[2], il + i);
            apply_PX(il + i, il + i - 1, 3, n - il - i + 1, il + i);
            apply_XP(H, 0, il + i, il + std::min<Index>(bsize, i + 4), 3, il + i);
        }
        compute_reflector(H(iu - 1, iu - 2), H(iu, iu - 2), 0.0, iu - 1);
        apply_PX(iu - 1, iu - 2, 2, n - iu + 2, iu - 1);
        apply_XP(H, 0, iu - 1, il + bsize, 2, iu - 1);
        ref_nr[iu] = 1;
    }

    // :358-425
    void compute(const Mat& mat, double s, double t)
    {
        n = mat.rows;
        if (n != mat.cols)
            throw std::invalid_argument("DoubleShiftQR: matrix must be square");
        H = mat;
        shift_s = s;
        shift_t = t;
        ref_u.assign(size_t(3) * n, 0.0);
        ref_nr.assign(n, 0);

        const double eps_abs = kNear0 * (double(n) / kEps);
        const double eps_rel = kEps;
        std::vector<Index> zero_ind;
        zero_ind.push_back(0);
        for (Index i = 0; i < n - 1; i++)
        {
            const double h = std::fabs(H(i + 1, i));
            const double diag = std::fabs(H(i, i)) + std::fabs(H(i + 1, i + 1));
            if (h <= eps_abs || h <= eps_rel * diag)
            {
                H(i + 1, i) = 0.0;
                zero_ind.push_back(i + 1);
            }
            for (Index r = i + 2; r < n; r++)
                H(r, i) = 0.0;
        }
        zero_ind.push_back(n);
        for (size_t b = 0; b + 1 < zero_ind.size(); b++)
            update_block(zero_ind[b], zero_ind[b + 1] - 1);
        for (Index i = 0; i < n - 1; i++)
        {
            const double h = std::fabs(H(i + 1, i));
            const double diag = std::fabs(H(i, i)) + std::fabs(H(i + 1, i + 1));
            if (h <= eps_abs || h <= eps_rel * diag)
                H(i + 1, i) = 0.0;
        }
        computed = true;
    }

    void matrix_QtHQ(Mat& dest) const  // :427-433
    {
        if (!computed)
            throw std::logic_error("DoubleShiftQR: need to call compute() first");
        dest = H;
    }

    // :455-467  Y <- Y P0 P1 ...
    void apply_YQ(Mat& Y) const
    {
        const Index nrow = Y.rows;
        const Index n2 = n - 2;
        for (Index i = 0; i < n2; i++)
            apply_XP(Y, 0, i, nrow, 3, i);
        apply_XP(Y, 0, n2, nrow, 2, n2);
    }
};

// ----------------------------------------------------------------------------
// [Eigen] Householder.h MatrixBase::makeHouseholder for a real 3-vector, and
// Jacobi.h makeGivens(p, q, &r) (real).
// ----------------------------------------------------------------------------
inline void eigen_make_householder3(const double v[3], double ess[2], double& tau, double& beta)
{
    const double tailSqNorm = v[1] * v[1] + v[2] * v[2];
    const double c0 = v[0];
    if (tailSqNorm <= DBL_MIN)
    {
        tau = 0.0;
        beta = c0;
        ess[0] = ess[1] = 0.0;
    }
    else
    {
        beta = std::sqrt(c0 * c0 + tailSqNorm);
        if (c0 >= 0.0)
            beta = -beta;
        ess[0] = v[1] / (c0 - beta);
        ess[1] = v[2] / (c0 - beta);
        tau = (beta - c0) / beta;
    }
}
inline void eigen_make_givens_r(double p, double q, double& c, double& s, double& r)
{
    if (q == 0.0)
    {
        c = p < 0.0 ? -1.0 : 1.0;
        s = 0.0;
        r = std::fabs(p);
    }
    else if (p == 0.0)
    {
        c = 0.0;
        s = q < 0.0 ? 1.0 : -1.0;
        r = std::fabs(q);
    }
    else if (std::fabs(p) > std::fabs(q))
    {
        const double t = q / p;
        double u = std::sqrt(1.0 + t * t);
        if (p < 0.0)
            u = -u;
        c = 1.0 / u;
        s = -t * c;
        r = p * u;
    }
    else
    {
        const double t = p / q;
        double u = std::sqrt(1.0 + t * t);
        if (q < 0.0)
            u = -u;
        s = -1.0 / u;
        c = -t * s;
        r = q * u;
    }
}

// ----------------------------------------------------------------------------
// LinAlg/UpperHessenbergSchur.h:24-456 (adapted by the reference from Eigen's RealSchur)
// ----------------------------------------------------------------------------
class UpperHessenbergSchur
{
public:
    Index n = 0;
    Mat T, U;
    bool computed = false;

    // rows p, q of M over columns [c0, ncols): x' = c x - s y ; y' = s x + c y   ([Eigen] applyOnTheLeft(p,q,rot.adjoint()))
    static void rot_rows(Mat& M, Index p, Index q, Index c0, double c, double s)
    {
        for (Index j = c0; j < M.cols; j++)
        {
            const double x = M(p, j), y = M(q, j);
            M(p, j) = c * x - s * y;
            M(q, j) = s * x + c * y;
        }
    }
    // columns p, q of M over rows [0, nrows): same formula ([Eigen] applyOnTheRight(p,q,rot))
    static void rot_cols(Mat& M, Index p, Index q, Index nrows, double c, double s)
    {
        double* xp = M.col(p);
        double* yq = M.col(q);
        for (Index i = 0; i < nrows; i++)
        {
            const double x = xp[i], y = yq[i];
            xp[i] = c * x - s * y;
            yq[i] = s * x + c * y;
        }
    }

    // :57-72
    Index find_small_subdiag(Index iu, double near_0) const
    {
        Index res = iu;
        while (res > 0)
        {
            double s = std::fabs(T(res - 1, res - 1)) + std::fabs(T(res, res));
            s = std::max(s * kEps, near_0);
            if (std::fabs(T(res, res - 1)) <= s)
                break;
            res--;
        }
        return res;
    }

    // :75-101
    void split_off_two_rows(Index iu, double ex_shift)
    {
        const double p = 0.5 * (T(iu - 1, iu - 1) - T(iu, iu));
        const double q = p * p + T(iu, iu - 1) * T(iu - 1, iu);
        T(iu, iu) += ex_shift;
        T(iu - 1, iu - 1) += ex_shift;
        if (q >= 0.0)
        {
            const double z = std::sqrt(std::fabs(q));
            double c, s;
            eigen_make_givens((p >= 0.0) ? (p + z) : (p - z), T(iu, iu - 1), c, s);
            rot_rows(T, iu - 1, iu, iu - 1, c, s);   // rightCols(n - iu + 1)
            rot_cols(T, iu - 1, iu, iu + 1, c, s);   // topRows(iu + 1)
            T(iu, iu - 1) = 0.0;
            rot_cols(U, iu - 1, iu, n, c, s);
        }
        if (iu > 1)
            T(iu - 1, iu - 2) = 0.0;
    }

    // :104-145
    void compute_shift(Index iu, Index iter, double& ex_shift, double si[3])
    {
        si[0] = T(iu, iu);
        si[1] = T(iu - 1, iu - 1);
        si[2] = T(iu, iu - 1) * T(iu - 1, iu);
        if (iter == 10)
        {
            ex_shift += si[0];
            for (Index i = 0; i <= iu; ++i)
                T(i, i) -= si[0];
            const double s = std::fabs(T(iu, iu - 1)) + std::fabs(T(iu - 1, iu - 2));
            si[0] = 0.75 * s;
            si[1] = 0.75 * s;
            si[2] = -0.4375 * s * s;
        }
        if (iter == 30)
        {
            double s = (si[1] - si[0]) / 2.0;
            s = s * s + si[2];
            if (s > 0.0)
            {
                s = std::sqrt(s);
                if (si[1] < si[0])
                    s = -s;
                s = s + (si[1] - si[0]) / 2.0;
                s = si[0] - si[2] / s;
                ex_shift += s;
                for (Index i = 0; i <= iu; ++i)
                    T(i, i) -= s;
                si[0] = si[1] = si[2] = 0.964;
            }
        }
    }

    // :148-170
    void init_francis_qr_step(Index il, Index iu, const double si[3], Index& im, double v[3]) const
    {
        for (im = iu - 2; im >= il; --im)
        {
            const double Tmm = T(im, im);
            const double r = si[0] - Tmm;
            const double s = si[1] - Tmm;
            v[0] = (r * s - si[2]) / T(im + 1, im) + T(im, im + 1);
            v[1] = T(im + 1, im + 1) - Tmm - r - s;
            v[2] = T(im + 2, im + 1);
            if (im == il)
                break;
            const double lhs = T(im, im - 1) * (std::fabs(v[1]) + std::fabs(v[2]));
            const double rhs = v[0] * (std::fabs(T(im - 1, im - 1)) + std::fabs(Tmm) + std::fabs(T(im + 1, im + 1)));
            if (std::fabs(lhs) < kEps * rhs)
                break;
        }
    }

    // :287-340
    void perform_francis_qr_step(Index il, Index im, Index iu, const double first_v[3], double near_0)
    {
        for (Index k = im; k <= iu - 2; ++k)
        {
            const bool first_iter = (k == im);
            double v[3];
            if (first_iter)
            {
                v[0] = first_v[0];
                v[1] = first_v[1];
                v[2] = first_v[2];
            }
            else
            {
                v[0] = T(k, k - 1);
                v[1] = T(k + 1, k - 1);
                v[2] = T(k + 2, k - 1);
            }
            double tau, beta, ess[2];
            eigen_make_householder3(v, ess, tau, beta);
            if (std::fabs(beta) > near_0)
            {
                if (first_iter && k > il)
                    T(k, k - 1) = -T(k, k - 1);
                else if (!first_iter)
                    T(k, k - 1) = beta;
                const double v1 = ess[0], v2 = ess[1];
                // apply_householder_left :173-185 on T(k:k+2, k:n-1)
                for (Index j = k; j < n; j++)
                {
                    double* x = &T(k, j);
                    const double tvx = tau * (x[0] + v1 * x[1] + v2 * x[2]);
                    x[0] -= tvx;
                    x[1] -= tvx * v1;
                    x[2] -= tvx * v2;
                }
                // apply_householder_right :189-203 on T(0:min(iu,k+3), k:k+2) and U(:, k:k+2)
                const Index nr = std::min(iu, k + 3) + 1;
                for (int pass = 0; pass < 2; pass++)
                {
                    Mat& M = pass == 0 ? T : U;
                    const Index rows = pass == 0 ? nr : n;
                    double *x0 = M.col(k), *x1 = M.col(k + 1), *x2 = M.col(k + 2);
                    for (Index i = 0; i < rows; i++)
                    {
                        const double txv = tau * (x0[i] + v1 * x1[i] + v2 * x2[i]);
                        x0[i] -= txv;
                        x1[i] -= txv * v1;
                        x2[i] -= txv * v2;
                    }
                }
            }
        }
        double c, s, beta;
        eigen_make_givens_r(T(iu - 1, iu - 2), T(iu, iu - 2), c, s, beta);
        if (std::fabs(beta) > near_0)
        {
            T(iu - 1, iu - 2) = beta;
            rot_rows(T, iu - 1, iu, iu - 1, c, s);
            rot_cols(T, iu - 1, iu, iu + 1, c, s);
            rot_cols(U, iu - 1, iu, n, c, s);
        }
        for (Index i = im + 2; i <= iu; ++i)  // :332-339
        {
            T(i, i - 2) = 0.0;
            if (i > im + 2)
                T(i, i - 3) = 0.0;
        }
    }

    // :354-421
    void compute(const Mat& mat)
    {
        n = mat.rows;
        if (n != mat.cols)
            throw std::invalid_argument("UpperHessenbergSchur: matrix must be square");
        T = mat;
        U.resize(n, n);
        U.set_identity();
        const Index max_iter = n * 40;
        Index iu = n - 1, iter = 0, total_iter = 0;
        double ex_shift = 0.0;
        double norm = 0.0;  // :46-53 L1 norm of the Hessenberg part
        for (Index j = 0; j < n; j++)
            for (Index i = 0; i < std::min(n, j + 2); i++)
                norm += std::fabs(T(i, j));
        const double near_0 = std::max(norm * kEps * kEps, kMin);
        if (norm != 0.0)
        {
            while (iu >= 0)
            {
                const Index il = find_small_subdiag(iu, near_0);
                if (il == iu)
                {
                    T(iu, iu) += ex_shift;
                    if (iu > 0)
                        T(iu, iu - 1) = 0.0;
                    iu--;
                    iter = 0;
                }
                else if (il == iu - 1)
                {
                    split_off_two_rows(iu, ex_shift);
                    iu -= 2;
                    iter = 0;
                }
                else
                {
                    double first_v[3] = {0.0, 0.0, 0.0}, si[3];
                    compute_shift(iu, iter, ex_shift, si);
                    iter++;
                    total_iter++;
                    if (total_iter > max_iter)
                        break;
                    Index im;
                    init_francis_qr_step(il, iu, si, im, first_v);
                    perform_francis_qr_step(il, im, iu, first_v, near_0);
                }
            }
        }
        if (total_iter > max_iter)
            throw std::runtime_error("UpperHessenbergSchur: Schur decomposition failed");
        computed = true;
    }
};

// ----------------------------------------------------------------------------
// LinAlg/UpperHessenbergEigen.h:28-320 (real)
// ----------------------------------------------------------------------------
class UpperHessenbergEigen
{
public:
    Index n = 0;
    Mat matT, eivec;
    std::vector<Complex> eivalues;
    bool computed = false;

    // :53-218
    void compute_eigenvectors()
    {
        const Index size = eivec.cols;
        double norm = 0.0;
        for (Index j = 0; j < size; ++j)
            for (Index c = std::max<Index>(j - 1, 0); c < size; c++)
                norm += std::fabs(matT(j, c));
        if (norm == 0.0)
            return;

        auto rowdot = [&](Index i, Index col, Index l, Index nn) {  // matT.row(i).segment(l, nn-l+1) . matT.col(col).segment(l, nn-l+1)
            double r = 0.0;
            for (Index k = l; k <= nn; k++)
                r += matT(i, k) * matT(k, col);
            return r;
        };

        for (Index nn = size - 1; nn >= 0; nn--)
        {
            const double p = eivalues[nn].real();
            const double q = eivalues[nn].imag();
            if (q == 0.0)
            {
                double lastr = 0.0, lastw = 0.0;
                Index l = nn;
                matT(nn, nn) = 1.0;
                for (Index i = nn - 1; i >= 0; i--)
                {
                    const double w = matT(i, i) - p;
                    const double r = rowdot(i, nn, l, nn);
                    if (eivalues[i].imag() < 0.0)
                    {
                        lastw = w;
                        lastr = r;
                    }
                    else
                    {
                        l = i;
                        if (eivalues[i].imag() == 0.0)
                        {
                            if (w != 0.0)
                                matT(i, nn) = -r / w;
                            else
                                matT(i, nn) = -r / (kEps * norm);
                        }
                        else
                        {
                            const double x = matT(i, i + 1);
                            const double y = matT(i + 1, i);
                            const double denom = (eivalues[i].real() - p) * (eivalues[i].real() - p) +
                                eivalues[i].imag() * eivalues[i].imag();
                            const double t = (x * lastr - lastw * r) / denom;
                            matT(i, nn) = t;
                            if (std::fabs(x) > std::fabs(lastw))
                                matT(i + 1, nn) = (-r - w * t) / x;
                            else
                                matT(i + 1, nn) = (-lastr - y * t) / lastw;
                        }
                        const double t = std::fabs(matT(i, nn));
                        if ((kEps * t) * t > 1.0)
                            for (Index k = i; k < size; k++)  // col(nn).tail(size - i) /= t
                                matT(k, nn) /= t;
                    }
                }
            }
            else if (q < 0.0 && nn > 0)
            {
                double lastra = 0.0, lastsa = 0.0, lastw = 0.0;
                Index l = nn - 1;
                if (std::fabs(matT(nn, nn - 1)) > std::fabs(matT(nn - 1, nn)))
                {
                    matT(nn - 1, nn - 1) = q / matT(nn, nn - 1);
                    matT(nn - 1, nn) = -(matT(nn, nn) - p) / matT(nn, nn - 1);
                }
                else
                {
                    const Complex cc = Complex(0.0, -matT(nn - 1, nn)) / Complex(matT(nn - 1, nn - 1) - p, q);
                    matT(nn - 1, nn - 1) = cc.real();
                    matT(nn - 1, nn) = cc.imag();
                }
                matT(nn, nn - 1) = 0.0;
                matT(nn, nn) = 1.0;
                for (Index i = nn - 2; i >= 0; i--)
                {
                    const double ra = rowdot(i, nn - 1, l, nn);
                    const double sa = rowdot(i, nn, l, nn);
                    const double w = matT(i, i) - p;
                    if (eivalues[i].imag() < 0.0)
                    {
                        lastw = w;
                        lastra = ra;
                        lastsa = sa;
                    }
                    else
                    {
                        l = i;
                        if (eivalues[i].imag() == 0.0)
                        {
                            const Complex cc = Complex(-ra, -sa) / Complex(w, q);
                            matT(i, nn - 1) = cc.real();
                            matT(i, nn) = cc.imag();
                        }
                        else
                        {
                            const double x = matT(i, i + 1);
                            const double y = matT(i + 1, i);
                            double vr = (eivalues[i].real() - p) * (eivalues[i].real() - p) +
                                eivalues[i].imag() * eivalues[i].imag() - q * q;
                            const double vi = (eivalues[i].real() - p) * 2.0 * q;
                            if (vr == 0.0 && vi == 0.0)
                                vr = kEps * norm * (std::fabs(w) + std::fabs(q) + std::fabs(x) + std::fabs(y) + std::fabs(lastw));
                            Complex cc = Complex(x * lastra - lastw * ra + q * sa, x * lastsa - lastw * sa - q * ra) / Complex(vr, vi);
                            matT(i, nn - 1) = cc.real();
                            matT(i, nn) = cc.imag();
                            if (std::fabs(x) > (std::fabs(lastw) + std::fabs(q)))
                            {
                                matT(i + 1, nn - 1) = (-ra - w * matT(i, nn - 1) + q * matT(i, nn)) / x;
                                matT(i + 1, nn) = (-sa - w * matT(i, nn) - q * matT(i, nn - 1)) / x;
                            }
                            else
                            {
                                cc = Complex(-lastra - y * matT(i, nn - 1), -lastsa - y * matT(i, nn)) / Complex(lastw, q);
                                matT(i + 1, nn - 1) = cc.real();
                                matT(i + 1, nn) = cc.imag();
                            }
                        }
                        const double t = std::max(std::fabs(matT(i, nn - 1)), std::fabs(matT(i, nn)));
                        if ((kEps * t) * t > 1.0)
                            for (Index k = i; k < size; k++)  // block(i, nn-1, size-i, 2) /= t
                            {
                                matT(k, nn - 1) /= t;
                                matT(k, nn) /= t;
                            }
                    }
                }
                nn--;
            }
        }
        // Back transformation :211-217
        std::vector<double> tmp(size);
        for (Index j = size - 1; j >= 0; j--)
        {
            std::fill(tmp.begin(), tmp.end(), 0.0);
            for (Index c = 0; c <= j; c++)
            {
                const double t = matT(c, j);
                const double* ec = eivec.col(c);
                for (Index r = 0; r < size; r++)
                    tmp[r] += ec[r] * t;
            }
            std::copy(tmp.begin(), tmp.end(), eivec.col(j));
        }
    }

    // :231-285
    void compute(const Mat& mat)
    {
        if (mat.rows != mat.cols)
            throw std::invalid_argument("UpperHessenbergEigen: matrix must be square");
        n = mat.rows;
        const double scale = max_abs(mat.a.data(), Index(mat.a.size()));
        Mat scaled = mat;
        for (double& v : scaled.a)
            v /= scale;
        UpperHessenbergSchur schur;
        schur.compute(scaled);
        matT = schur.T;
        eivec = schur.U;
        eivalues.assign(n, Complex(0.0, 0.0));
        Index i = 0;
        while (i < n)
        {
            if (i == n - 1 || matT(i + 1, i) == 0.0)
            {
                eivalues[i] = Complex(matT(i, i), 0.0);
                ++i;
            }
            else
            {
                const double p = 0.5 * (matT(i, i) - matT(i + 1, i + 1));
                double z;
                {
                    double t0 = matT(i + 1, i), t1 = matT(i, i + 1);
                    const double maxval = std::max(std::fabs(p), std::max(std::fabs(t0), std::fabs(t1)));
                    t0 /= maxval;
                    t1 /= maxval;
                    const double p0 = p / maxval;
                    z = maxval * std::sqrt(std::fabs(p0 * p0 + t0 * t1));
                }
                eivalues[i] = Complex(matT(i + 1, i + 1) + p, z);
                eivalues[i + 1] = Complex(matT(i + 1, i + 1) + p, -z);
                i += 2;
            }
        }
        compute_eigenvectors();
        for (Complex& e : eivalues)
            e *= scale;
        computed = true;
    }

    // :296-327  complex eigenvectors, each column normalised ([Eigen] normalize(): x /= sqrt(sum |x|^2) if > 0)
    std::vector<Complex> eigenvectors() const
    {
        std::vector<Complex> V(size_t(n) * n);
        auto normalize = [&](Index j) {
            double z = 0.0;
            for (Index i = 0; i < n; i++)
                z += std::norm(V[size_t(j) * n + i]);
            if (z > 0.0)
            {
                const double s = std::sqrt(z);
                for (Index i = 0; i < n; i++)
                    V[size_t(j) * n + i] /= s;
            }
        };
        for (Index j = 0; j < n; ++j)
        {
            if (eivalues[j].imag() == 0.0 || j + 1 == n)
            {
                for (Index i = 0; i < n; i++)
                    V[size_t(j) * n + i] = Complex(eivec(i, j), 0.0);
                normalize(j);
            }
            else
            {
                for (Index i = 0; i < n; ++i)
                {
                    V[size_t(j) * n + i] = Complex(eivec(i, j), eivec(i, j + 1));
                    V[size_t(j + 1) * n + i] = Complex(eivec(i, j), -eivec(i, j + 1));
                }
                normalize(j);
                normalize(j + 1);
                ++j;
            }
        }
        return V;
    }
};

// ----------------------------------------------------------------------------
// Util/SelectionRule.h:62-225 sorting targets for complex values
// ----------------------------------------------------------------------------
inline std::vector<Index> argsort_complex(SortRule rule, const Complex* values, Index len)
{
    std::function<double(const Complex&)> target;
    switch (rule)
    {
        case SortRule::LargestMagn:
            target = [](const Complex& v) { return -std::abs(v); };
            break;
        case SortRule::LargestReal:
            target = [](const Complex& v) { return -v.real(); };
            break;
        case SortRule::LargestImag:
            target = [](const Complex& v) { return -std::fabs(v.imag()); };
            break;
        case SortRule::SmallestMagn:
            target = [](const Complex& v) { return std::abs(v); };
            break;
        case SortRule::SmallestReal:
            target = [](const Complex& v) { return v.real(); };
            break;
        case SortRule::SmallestImag:
            target = [](const Complex& v) { return std::fabs(v.imag()); };
            break;
        default:
            throw std::invalid_argument("unsupported selection rule");
    }
    std::vector<Index> ind(len);
    for (Index i = 0; i < len; i++)
        ind[i] = i;
    std::sort(ind.begin(), ind.end(), [&](Index i, Index j) { return target(values[i]) < target(values[j]); });
    return ind;
}

// ----------------------------------------------------------------------------
// GenEigsBase.h:43-611 (real Scalar) + GenEigsSolver.h
// ----------------------------------------------------------------------------
class GenEigs
{
public:
    const Op& op;
    const Index n, nev, ncv;
    Index nmatop = 0, niter = 0;
    Factorization fac;
    std::vector<Complex> ritz_val, ritz_est;
    std::vector<Complex> ritz_vec;  // ncv x nev column-major
    std::vector<char> ritz_conv;
    CompInfo info = CompInfo::NotComputed;
    // GenEigsRealShiftSolver.h:52-58: lambda = 1 / nu + sigma before the final sort
    bool shift_invert = false;
    double sigma = 0.0;
    // GenEigsComplexShiftSolver.h:20-150: `op` is x -> Re((A - sigma I)^{-1} x) with sigma = sigmar + i sigmai; op_probe is
    // the same operator at the real probe shift the reference draws from SimpleRandom(0) (:69-72)
    bool complex_shift = false;
    double sigmar = 0.0, sigmai = 0.0;
    const Op* op_probe = nullptr;
    static double probe_shift(double sigmar_)
    {
        SimpleRandom rng(0);
        const double a = rng.random();
        const double b = rng.random();
        return a * sigmar_ + b;
    }

    static bool is_complex(const Complex& v) { return v.imag() != 0.0; }
    static bool is_conj(const Complex& a, const Complex& b) { return a == std::conj(b); }

    GenEigs(const Op& op_, Index nev_, Index ncv_) :
        op(op_), n(op_.rows()), nev(nev_), ncv(ncv_ > n ? n : ncv_), fac(op_, ncv_ > n ? n : ncv_)
    {
        // GenEigsBase.h:419-423
        if (nev_ < 1 || nev_ > n - 2)
            throw std::invalid_argument("nev must satisfy 1 <= nev <= n - 2, n is the size of matrix");
        if (ncv_ < nev_ + 2 || ncv_ > n)
            throw std::invalid_argument("ncv must satisfy nev + 2 <= ncv <= n, n is the size of matrix");
    }

    void init(const double* init_resid)  // :442-462
    {
        ritz_val.assign(ncv, Complex(0, 0));
        ritz_vec.assign(size_t(ncv) * nev, Complex(0, 0));
        ritz_est.assign(ncv, Complex(0, 0));
        ritz_conv.assign(nev, 0);
        nmatop = 0;
        niter = 0;
        fac.init(init_resid, nmatop);
    }
    void init()  // :471-476
    {
        SimpleRandom rng(0);
        std::vector<double> v0(n);
        rng.fill(v0.data(), n);
        init(v0.data());
    }

    // :280-340
    void retrieve_ritzpair(SortRule selection)
    {
        UpperHessenbergEigen decomp;
        decomp.compute(fac.H);
        const std::vector<Complex>& evals = decomp.eivalues;
        const std::vector<Complex> evecs = decomp.eigenvectors();
        std::vector<Index> ind = argsort_complex(selection, evals.data(), ncv);
        for (Index i = 0; i < ncv; i++)
        {
            ritz_val[i] = evals[ind[i]];
            ritz_est[i] = evecs[size_t(ind[i]) * ncv + (ncv - 1)];
        }
        for (Index i = 0; i < nev; i++)
            for (Index r = 0; r < ncv; r++)
                ritz_vec[size_t(i) * ncv + r] = evecs[size_t(ind[i]) * ncv + r];
    }

    // :225-242
    Index num_converged(double tol)
    {
        const double eps23 = std::pow(kEps, 2.0 / 3.0);
        Index cnt = 0;
        for (Index i = 0; i < nev; i++)
        {
            const double thresh = tol * std::max(eps23, std::abs(ritz_val[i]));
            const double resid = std::abs(ritz_est[i]) * fac.beta;
            ritz_conv[i] = (resid < thresh);
            cnt += ritz_conv[i];
        }
        return cnt;
    }

    // :245-277
    Index nev_adjusted(Index nconv)
    {
        Index nev_new = nev;
        for (Index i = nev; i < ncv; i++)
            if (std::abs(ritz_est[i]) < kNear0)
                nev_new++;
        nev_new += std::min(nconv, (ncv - nev_new) / 2);
        if (nev_new == 1 && ncv >= 6)
            nev_new = ncv / 2;
        else if (nev_new == 1 && ncv > 3)
            nev_new = 2;
        if (nev_new > ncv - 2)
            nev_new = ncv - 2;
        if (is_complex(ritz_val[nev_new - 1]) && is_conj(ritz_val[nev_new - 1], ritz_val[nev_new]))
            nev_new++;
        return nev_new;
    }

    // :204-222 with RestartArnoldi<double>::run (:60-107)
    void restart(Index k, SortRule selection)
    {
        if (k >= ncv)
            return;
        Mat Q(ncv, ncv);
        Q.set_identity();
        DoubleShiftQR decomp_ds;
        UpperHessenbergQR decomp_hb;
        for (Index i = k; i < ncv; i++)
        {
            if (is_complex(ritz_val[i]) && i + 1 < ncv && is_conj(ritz_val[i], ritz_val[i + 1]))
            {
                const double s = 2.0 * ritz_val[i].real();
                const double t = std::norm(ritz_val[i]);
                decomp_ds.compute(fac.H, s, t);
                decomp_ds.apply_YQ(Q);
                decomp_ds.matrix_QtHQ(fac.H);  // Arnoldi.h:299-303 compress_H: k -= 2
                fac.k -= 2;
                i++;
            }
            else
            {
                decomp_hb.compute(fac.H, ritz_val[i].real());
                decomp_hb.apply_YQ(Q);
                decomp_hb.matrix_QtHQ(fac.H);  // Arnoldi.h:306-310: k--
                fac.k--;
            }
        }
        fac.compress_V(Q);
        fac.factorize_from_arnoldi(k, ncv, nmatop);
        retrieve_ritzpair(selection);
    }

    // :345-401
    void sort_ritzpair(SortRule sort_rule)
    {
        if (shift_invert)
            for (Index i = 0; i < nev; i++)
                ritz_val[i] = Complex(1.0, 0.0) / ritz_val[i] + sigma;
        if (complex_shift)  // GenEigsComplexShiftSolver.h:39-123
        {
            const double shiftr = probe_shift(sigmar);
            const Complex shift(shiftr, 0.0);
            const double eps = std::numeric_limits<double>::epsilon();
            std::vector<double> v_re(n), v_im(n), op_re(n), op_im(n);
            for (Index i = 0; i < nev; i++)
            {
                std::fill(v_re.begin(), v_re.end(), 0.0);
                std::fill(v_im.begin(), v_im.end(), 0.0);
                for (Index c = 0; c < ncv; c++)  // v = V * ritz_vec.col(i)  (:80-81)
                {
                    const Complex y = ritz_vec[size_t(i) * ncv + c];
                    const double* vc = &fac.V(0, c);
                    for (Index r = 0; r < n; r++)
                    {
                        v_re[r] += vc[r] * y.real();
                        v_im[r] += vc[r] * y.imag();
                    }
                }
                op_probe->perform_op(v_re.data(), op_re.data());
                op_probe->perform_op(v_im.data(), op_im.data());
                const Complex nu = ritz_val[i];
                const Complex part1 = sigmar + 0.5 / nu;  // :85-90
                const Complex part2 = 0.5 * std::sqrt(1.0 - 4.0 * sigmai * sigmai * (nu * nu)) / nu;
                const Complex root1 = part1 + part2, root2 = part1 - part2;
                double err1 = 0.0, err2 = 0.0;
                for (Index k = 0; k < n; k++)  // :93-101
                {
                    const Complex v(v_re[k], v_im[k]), opv(op_re[k], op_im[k]);
                    err1 += std::norm(opv - v / (root1 - shift));
                    err2 += std::norm(opv - v / (root2 - shift));
                }
                const Complex lambdaj = (err1 < err2) ? root1 : root2;
                ritz_val[i] = lambdaj;
                if (std::fabs(lambdaj.imag()) > eps)  // :106-114
                {
                    if (i + 1 < ncv)
                        ritz_val[i + 1] = std::conj(lambdaj);
                    i++;
                }
                else
                    ritz_val[i] = Complex(lambdaj.real(), 0.0);
            }
        }
        std::vector<Index> ind;
        try
        {
            ind = argsort_complex(sort_rule, ritz_val.data(), nev);
        }
        catch (const std::invalid_argument&)
        {
            throw std::invalid_argument("unsupported sorting rule");
        }
        std::vector<Complex> new_val(ncv, Complex(0, 0)), new_vec(size_t(ncv) * nev);
        std::vector<char> new_conv(nev, 0);
        for (Index i = 0; i < nev; i++)
        {
            new_val[i] = ritz_val[ind[i]];
            for (Index r = 0; r < ncv; r++)
                new_vec[size_t(i) * ncv + r] = ritz_vec[size_t(ind[i]) * ncv + r];
            new_conv[i] = ritz_conv[ind[i]];
        }
        ritz_val.swap(new_val);
        ritz_vec.swap(new_vec);
        ritz_conv.swap(new_conv);
    }

    // :501-525
    Index compute(SortRule selection = SortRule::LargestMagn, Index maxit = 1000, double tol = 1e-10,
                  SortRule sorting = SortRule::LargestMagn)
    {
        fac.factorize_from_arnoldi(1, ncv, nmatop);
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

    // :548-567
    std::vector<Complex> eigenvalues() const
    {
        std::vector<Complex> res;
        for (Index i = 0; i < nev; i++)
            if (ritz_conv[i])
                res.push_back(ritz_val[i]);
        return res;
    }

    // :578-602  n x nvec complex, column-major
    std::vector<Complex> eigenvectors(Index nvec, Index& ncols) const
    {
        Index nconv = 0;
        for (Index i = 0; i < nev; i++)
            nconv += ritz_conv[i];
        nvec = std::min(nvec, nconv);
        ncols = nvec;
        std::vector<Complex> res(size_t(n) * nvec, Complex(0, 0));
        Index j = 0;
        for (Index i = 0; i < nev && j < nvec; i++)
        {
            if (!ritz_conv[i])
                continue;
            for (Index c = 0; c < ncv; c++)
            {
                const Complex y = ritz_vec[size_t(i) * ncv + c];
                const double* vc = fac.V.col(c);
                for (Index r = 0; r < n; r++)
                    res[size_t(j) * n + r] += vc[r] * y;
            }
            j++;
        }
        return res;
    }
};

}  // namespace oracle
