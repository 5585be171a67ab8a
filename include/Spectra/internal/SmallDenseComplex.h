// Complex-scalar versions of the small dense routines (host only).  The device path computes in real fp64 — the complex
// cases exist because the reference's host-side LinAlg classes are templates over the scalar and its own unit tests
// (test/Givens.cpp, test/QR.cpp, test/Eigen.cpp) instantiate them with std::complex<double> next to double.
//
// Reference arithmetic followed (yixuan/spectra, include/Spectra/):
//   givens_rotation_complex   LinAlg/Givens.h:88-130 (StableScaling for complex operands), :236-339
//   hess_shifted_qr_complex   LinAlg/UpperHessenbergQR.h:136-195 (compute), :219-255 (matrix_QtHQ)
//   hess_eigen_complex        LinAlg/UpperHessenbergEigen.h:345-429: Schur form (the reference delegates it to
//                             Eigen::ComplexSchur::computeFromHessenberg, third party; here the single-shift QR iteration
//                             with Wilkinson shifts and the same deflation rule), back substitution, unit columns,
//                             eigenvalues in increasing modulus (:384-399)
#pragma once

#include <algorithm>
#include <cmath>
#include <complex>
#include <limits>
#include <stdexcept>
#include <vector>

#include "SmallDense.h"

namespace mispec {
namespace small {

// |a|_1 >= |b|_1 > 0: a2 = |a|^2, tc1 = sqrt(1 + t^2), tc2 = 1 / sqrt(1 + t^2) with t = |b| / |a|  (Givens.h:88-130)
inline void stable_scaling_complex(const std::complex<double>& a, const std::complex<double>& b, double& a2, double& tc1, double& tc2)
{
    const double b2 = std::norm(b);
    a2 = std::norm(a);
    const double t2 = b2 / a2;  // 0 < t2 <= 2
    const double cutoff = 0.1 * std::sqrt(kEps);
    if (t2 >= cutoff)
    {
        tc1 = std::sqrt(1.0 + t2);
        tc2 = std::sqrt(a2 / (a2 + b2));
    }
    else
    {
        tc1 = 1.0 + t2 * (0.5 - t2 * (0.125 - 0.0625 * t2));
        tc2 = 1.0 - t2 * (0.5 - t2 * (0.375 - 0.3125 * t2));
    }
}

// G = [c s; -conj(s) c] with real c:  c x - s y = r,  conj(s) x + c y = 0   (Givens.h:236-339)
inline void givens_rotation_complex(const std::complex<double>& x, const std::complex<double>& y, std::complex<double>& r, double& c,
                                    std::complex<double>& s)
{
    using cd = std::complex<double>;
    const cd zero(0.0, 0.0);
    if (y == zero)
    {
        c = 1.0;
        s = zero;
        r = x;
        return;
    }
    if (x == zero)
    {
        // r = |y|, s = -conj(y) / |y|: the real rotation of (-Re y, -Im y)
        c = 0.0;
        double rr, sr, si;
        givens_rotation(-y.real(), -y.imag(), rr, sr, si);
        s = cd(sr, si);
        r = cd(rr, 0.0);
        return;
    }
    const double xn1 = std::fabs(x.real()) + std::fabs(x.imag());
    const double yn1 = std::fabs(y.real()) + std::fabs(y.imag());
    if (xn1 > yn1)
    {
        double x2, tc1, tc2;
        stable_scaling_complex(x, y, x2, tc1, tc2);
        c = tc2;
        r = tc1 * x;
        s = -(c / x2) * (x * std::conj(y));
    }
    else
    {
        const double rho = std::sqrt(std::norm(x) + std::norm(y));
        double xnorm, zr, zi;  // z = x / |x| from the real rotation of (Re x, -Im x)
        givens_rotation(x.real(), -x.imag(), xnorm, zr, zi);
        const cd z(zr, zi);
        r = rho * z;
        c = xnorm / rho;
        s = -(z * std::conj(y)) / rho;
    }
}

// H - shift I = QR for an upper Hessenberg H (n x n, column-major, leading dimension ldh; entries below the sub-diagonal are
// not read).  Out: R (n x n, contiguous), the rotations (rot_cos[n-1] real, rot_sin[n-1]) with Q = G_0 G_1 ... G_{n-2}.
inline void hess_shifted_qr_complex(int n, const std::complex<double>* H, long ldh, const std::complex<double>& shift,
                                    std::complex<double>* R, double* rot_cos, std::complex<double>* rot_sin)
{
    using cd = std::complex<double>;
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++)
            R[(long) j * n + i] = (i <= j + 1) ? H[(long) j * ldh + i] - (i == j ? shift : cd(0.0)) : cd(0.0);
    for (int i = 0; i < n - 1; i++)
    {
        cd* Rii = R + (long) i * n + i;
        cd r, s;
        double c;
        givens_rotation_complex(Rii[0], Rii[1], r, c, s);
        rot_cos[i] = c;
        rot_sin[i] = s;
        Rii[0] = r;
        Rii[1] = cd(0.0);
        cd* p = Rii + n;  // rows i, i+1 of the remaining columns: G^H applied from the left
        for (int j = i + 1; j < n; j++, p += n)
        {
            const cd tmp = p[0];
            p[0] = c * tmp - s * p[1];
            p[1] = std::conj(s) * tmp + c * p[1];
        }
    }
}

// dest = RQ + shift I = Q^H H Q  (UpperHessenbergQR.h:219-255)
inline void hess_rq_complex(int n, const std::complex<double>* R, const std::complex<double>& shift, const double* rot_cos,
                            const std::complex<double>* rot_sin, std::complex<double>* dest)
{
    using cd = std::complex<double>;
    std::copy(R, R + (long) n * n, dest);
    for (int i = 0; i < n - 1; i++)
    {
        const double c = rot_cos[i];
        const cd s = rot_sin[i];
        cd* Yi = dest + (long) i * n;
        cd* Yi1 = Yi + n;
        for (int j = 0; j < i + 2; j++)
        {
            const cd tmp = Yi[j];
            Yi[j] = c * tmp - std::conj(s) * Yi1[j];
            Yi1[j] = s * tmp + c * Yi1[j];
        }
    }
    for (int i = 0; i < n; i++)
        dest[(long) i * n + i] += shift;
}

// Complex Schur form of an upper Hessenberg matrix, in place: T <- U^H T U (upper triangular), U accumulated from the identity.
// Single-shift QR iteration; a sub-diagonal entry is set to zero when |.|_1 <= eps (|T(i,i)|_1 + |T(i+1,i+1)|_1).
inline bool hess_complex_schur(int n, std::complex<double>* Tp, std::complex<double>* Up)
{
    using cd = std::complex<double>;
    auto T = [&](int i, int j) -> cd& { return Tp[(long) j * n + i]; };
    auto U = [&](int i, int j) -> cd& { return Up[(long) j * n + i]; };
    auto n1 = [](const cd& z) -> double { return std::fabs(z.real()) + std::fabs(z.imag()); };
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++)
        {
            U(i, j) = (i == j) ? cd(1.0) : cd(0.0);
            if (i > j + 1)
                T(i, j) = cd(0.0);
        }
    auto negligible = [&](int i) -> bool {
        if (n1(T(i + 1, i)) <= kEps * (n1(T(i, i)) + n1(T(i + 1, i + 1))))
        {
            T(i + 1, i) = cd(0.0);
            return true;
        }
        return false;
    };
    // one plane rotation on the pair (p, p+1): rows of T from column j0, columns of T down to row i1, columns of U
    auto rotate = [&](int p, int j0, int i1, const cd& a, const cd& b) {
        cd r, s;
        double c;
        givens_rotation_complex(a, b, r, c, s);  // G^H (a, b)' = (r, 0)'
        for (int j = j0; j < n; j++)
        {
            const cd x = T(p, j), y = T(p + 1, j);
            T(p, j) = c * x - s * y;
            T(p + 1, j) = std::conj(s) * x + c * y;
        }
        for (int i = 0; i <= i1; i++)
        {
            const cd x = T(i, p), y = T(i, p + 1);
            T(i, p) = c * x - std::conj(s) * y;
            T(i, p + 1) = s * x + c * y;
        }
        for (int i = 0; i < n; i++)
        {
            const cd x = U(i, p), y = U(i, p + 1);
            U(i, p) = c * x - std::conj(s) * y;
            U(i, p + 1) = s * x + c * y;
        }
    };
    int iu = n - 1, iter = 0, total = 0;
    const int maxit = 30 * n;
    while (true)
    {
        while (iu > 0)
        {
            if (!negligible(iu - 1))
                break;
            iter = 0;
            --iu;
        }
        if (iu <= 0)
            return true;
        iter++;
        if (++total > maxit)
            return false;
        int il = iu - 1;
        while (il > 0 && !negligible(il - 1))
            --il;
        cd mu;
        if (iter == 10 || iter == 20)  // exceptional shift
            mu = cd(std::fabs(T(iu, iu - 1).real()) + (iu > 1 ? std::fabs(T(iu - 1, iu - 2).real()) : 0.0));
        else
        {
            // the eigenvalue of the trailing 2 x 2 block closer to its last diagonal entry
            cd t00 = T(iu - 1, iu - 1), t01 = T(iu - 1, iu), t10 = T(iu, iu - 1), t11 = T(iu, iu);
            const double nt = std::abs(t00) + std::abs(t01) + std::abs(t10) + std::abs(t11);
            t00 /= nt, t01 /= nt, t10 /= nt, t11 /= nt;
            const cd b = t01 * t10, d = t00 - t11, disc = std::sqrt(d * d + 4.0 * b);
            const cd det = t00 * t11 - b, tr = t00 + t11;
            cd e1 = (tr + disc) / 2.0, e2 = (tr - disc) / 2.0;
            if (n1(e1) > n1(e2))
                e2 = det / e1;
            else if (e2 != cd(0.0))
                e1 = det / e2;
            mu = nt * (n1(e1 - t11) < n1(e2 - t11) ? e1 : e2);
        }
        rotate(il, il > 0 ? il - 1 : 0, std::min(il + 2, iu), T(il, il) - mu, T(il + 1, il));
        for (int i = il + 1; i < iu; i++)
        {
            rotate(i, i - 1, std::min(i + 2, iu), T(i, i - 1), T(i + 1, i - 1));
            T(i + 1, i - 1) = cd(0.0);
        }
    }
}

// Eigenvalues (increasing modulus) and unit eigenvectors of a complex upper Hessenberg matrix.
inline void hess_eigen_complex(int n, const std::complex<double>* Hin, long ldh, std::complex<double>* evals, std::complex<double>* evecs)
{
    using cd = std::complex<double>;
    std::vector<cd> T((size_t) n * n), U((size_t) n * n), X((size_t) n * n, cd(0.0));
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++)
            T[(size_t) j * n + i] = Hin[(long) j * ldh + i];
    if (!hess_complex_schur(n, T.data(), U.data()))
        throw std::runtime_error("UpperHessenbergEigen: eigen decomposition failed");
    auto t = [&](int i, int j) -> const cd& { return T[(size_t) j * n + i]; };
    auto x = [&](int i, int j) -> cd& { return X[(size_t) j * n + i]; };
    double tnorm = 0.0;
    for (int j = 0; j < n; j++)
        for (int i = 0; i <= j; i++)
            tnorm += std::norm(t(i, j));
    tnorm = std::max(std::sqrt(tnorm), kMinPos);
    // X unit upper triangular with T X = X D  (UpperHessenbergEigen.h:352-376)
    for (int k = n - 1; k >= 0; k--)
    {
        x(k, k) = cd(1.0);
        for (int i = k - 1; i >= 0; i--)
        {
            cd acc = -t(i, k);
            for (int l = i + 1; l < k; l++)
                acc -= t(i, l) * x(l, k);
            cd z = t(i, i) - t(k, k);
            if (z == cd(0.0))
                z = cd(kEps * tnorm, 0.0);
            x(i, k) = acc / z;
        }
    }
    for (int k = 0; k < n; k++)
    {
        evals[k] = t(k, k);
        double nrm2 = 0.0;
        cd* v = evecs + (long) k * n;
        for (int i = 0; i < n; i++)
        {
            cd acc(0.0);
            for (int l = 0; l <= k; l++)
                acc += U[(size_t) l * n + i] * x(l, k);
            v[i] = acc;
            nrm2 += std::norm(acc);
        }
        const double nrm = std::sqrt(nrm2);
        for (int i = 0; i < n; i++)
            v[i] /= nrm;
    }
    // selection sort by modulus (:384-399)
    for (int i = 0; i < n; i++)
    {
        int k = i;
        for (int j = i + 1; j < n; j++)
            if (std::abs(evals[j]) < std::abs(evals[k]))
                k = j;
        if (k != i)
        {
            std::swap(evals[k], evals[i]);
            std::swap_ranges(evecs + (long) i * n, evecs + (long) (i + 1) * n, evecs + (long) k * n);
        }
    }
}

}  // namespace small
}  // namespace mispec
