// ncv x ncv dense algorithms of the restart, written once for two targets:
//   * device: one wavefront, T / rotations / Q resident in LDS, lane l owning rows l, l+64, ... of Q
//     (spectra_amd/csrc/small.hip);
//   * host: the same code with a single "lane" (used by include/Spectra/LinAlg/*.h when the Krylov
//     dimension exceeds what the device kernels hold, and for user operators).
// The scalar recurrences are executed redundantly by every lane (identical values, uniform control
// flow), only the row updates of Q are split across lanes — so no intra-wave synchronisation is needed.
//
// Reference arithmetic followed (yixuan/spectra v1.2.0, include/Spectra/):
//   givens_rotation        LinAlg/Givens.h:22-86, :149-206
//   tridiag_shifted_qr     LinAlg/UpperHessenbergQR.h:515-598 (compute), :383-417 (apply_YQ), :627-693 (matrix_QtHQ)
//   tridiag_eigen          LinAlg/TridiagEigen.h:44-108, :121-210 (+ Eigen's JacobiRotation::makeGivens / numext::hypot,
//                          restated from Eigen 3.4.0's published source; Eigen itself is not available here)
#pragma once

#include <cfloat>
#include <cmath>

#if defined(__HIPCC__)
#define MISPEC_HD __host__ __device__
#else
#define MISPEC_HD
#endif

namespace mispec {
namespace small {

constexpr double kEps = DBL_EPSILON;     // TypeTraits<double>::epsilon()
constexpr double kMinPos = DBL_MIN;      // TypeTraits<double>::min()
constexpr double kNear0 = DBL_MIN * 10;  // near_0 (Arnoldi.h:50)

// Which rows (or columns) of a small matrix this executor updates, and how executors rendezvous.
// Host: one lane, no synchronisation.  Device: the 64 lanes of the single wavefront that runs the
// kernel; sync() is the workgroup barrier (also a compiler barrier for the LDS-resident arrays).
struct Lanes
{
    int first;   // lane id
    int stride;  // number of lanes
    MISPEC_HD void sync() const
    {
#if defined(__HIP_DEVICE_COMPILE__)
        // == __syncthreads(), spelled with builtins so that this header needs no HIP runtime include
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
    }
};

// r = sqrt(a^2+b^2), c = a/r, s = b/r for a >= b > 0 (Givens.h:28-86)
MISPEC_HD inline void stable_scaling(double a, double b, double& r, double& c, double& s)
{
    const double t = b / a;
    const double cutoff = 0.1 * 1.220703125e-4;  // 0.1 * eps^(1/4), eps = 2^-52
    if (t >= cutoff)
    {
        r = hypot(a, b);
        c = a / r;
        s = b / r;
    }
    else
    {
        const double t2 = t * t;
        c = 1.0 - t2 * (0.5 - t2 * (0.375 - 0.3125 * t2));
        s = t * c;
        r = a + 0.5 * b * t * (1.0 - t2 * (0.25 - 0.125 * t2));
    }
}

// c*x - s*y = r, s*x + c*y = 0 (Givens.h:166-205)
MISPEC_HD inline void givens_rotation(double x, double y, double& r, double& c, double& s)
{
    const double xsign = (x > 0.0) ? 1.0 : -1.0;
    const double xabs = fabs(x);
    if (y == 0.0)
    {
        c = (x == 0.0) ? 1.0 : xsign;
        s = 0.0;
        r = xabs;
        return;
    }
    const double ysign = (y > 0.0) ? 1.0 : -1.0;
    const double yabs = fabs(y);
    if (x == 0.0)
    {
        c = 0.0;
        s = -ysign;
        r = yabs;
        return;
    }
    if (xabs >= yabs)
    {
        stable_scaling(xabs, yabs, r, c, s);
        c = xsign * c;
        s = -ysign * s;
    }
    else
    {
        stable_scaling(yabs, xabs, r, s, c);
        c = xsign * c;
        s = -ysign * s;
    }
}

// One implicit-restart shift on the tridiagonal T (diag[n], subd[n-1], overwritten with Q'TQ) and the
// accumulated orthogonal factor Q (qrows x n, column-major, leading dimension ldq), Q <- Q * Qi.
//   work: 4*n doubles (rot_cos, rot_sin, saved T_diag, saved T_subd)
// Follows HermEigsBase.h:124-147 for a single shift: TridiagQR::compute, apply_YQ, compress_H.
MISPEC_HD inline void tridiag_shifted_qr(int n, double* diag, double* subd, double shift, double* Q, int ldq, int qrows,
                                         double* work, Lanes lanes)
{
    double* rot_cos = work;
    double* rot_sin = work + n;
    double* T_diag = work + 2 * n;
    double* T_subd = work + 3 * n;

    // UpperHessenbergQR.h:526-539: save T, deflate tiny sub-diagonals
    for (int i = 0; i < n; i++)
        T_diag[i] = diag[i];
    for (int i = 0; i < n - 1; i++)
    {
        double e = subd[i];
        if (fabs(e) <= kEps * (fabs(diag[i]) + fabs(diag[i + 1])))
            e = 0.0;
        T_subd[i] = e;
    }

    // :541-590 Givens sweep on T - shift*I; only the running R[i,i] and R[i,i+1] are needed
    const int n1 = n - 1, n2 = n - 2;
    double r_diag = T_diag[0] - shift;
    double r_supd = (n1 > 0) ? T_subd[0] : 0.0;
    for (int i = 0; i < n1; i++)
    {
        double r, c, s;
        givens_rotation(r_diag, T_subd[i], r, c, s);
        rot_cos[i] = c;
        rot_sin[i] = s;
        const double Tii1 = r_supd;
        const double Ti1i1 = T_diag[i + 1] - shift;
        r_diag = s * Tii1 + c * Ti1i1;  // R[i+1,i+1]  (:575)
        if (i < n2)
            r_supd = c * T_subd[i + 1];  // R[i+1,i+2] (:582)

        // apply_YQ (:403-416): columns i, i+1 of Q, this lane's rows
        double* Yi = Q + (long) i * ldq;
        double* Yi1 = Yi + ldq;
        for (int j = lanes.first; j < qrows; j += lanes.stride)
        {
            const double tmp = Yi[j];
            Yi[j] = c * tmp - s * Yi1[j];
            Yi1[j] = s * tmp + c * Yi1[j];
        }
    }

    // matrix_QtHQ (:627-693): Q'TQ from the saved T, then re-deflate
    for (int i = 0; i < n; i++)
        diag[i] = T_diag[i];
    for (int i = 0; i < n1; i++)
        subd[i] = T_subd[i];
    for (int i = 0; i < n1; i++)
    {
        const double c = rot_cos[i], s = rot_sin[i];
        const double cs = c * s, c2 = c * c, s2 = s * s;
        const double x = diag[i], y = subd[i], z = diag[i + 1];
        const double c2x = c2 * x, s2x = s2 * x, c2z = c2 * z, s2z = s2 * z;
        const double csy2 = 2.0 * c * s * y;
        diag[i] = c2x - csy2 + s2z;
        subd[i] = cs * (x - z) + (c2 - s2) * y;
        diag[i + 1] = s2x + csy2 + c2z;
        if (i < n2)
        {
            const double ci1 = rot_cos[i + 1], si1 = rot_sin[i + 1];
            const double o = -s * T_subd[i + 1];
            subd[i + 1] *= c;
            subd[i] = ci1 * subd[i] - si1 * o;
        }
    }
    for (int i = 0; i < n1; i++)
    {
        const double dsum = fabs(diag[i]) + fabs(diag[i + 1]);
        if (fabs(subd[i]) <= kEps * dsum)
            subd[i] = 0.0;
    }
}

// Eigen 3.4.0 JacobiRotation<double>::makeGivens(p, q), real branch (used at TridiagEigen.h:79-80)
MISPEC_HD inline void eigen_make_givens(double p, double q, double& c, double& s)
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
    else if (fabs(p) > fabs(q))
    {
        const double t = q / p;
        double u = sqrt(1.0 + t * t);
        if (p < 0.0)
            u = -u;
        c = 1.0 / u;
        s = -t * c;
    }
    else
    {
        const double t = p / q;
        double u = sqrt(1.0 + t * t);
        if (q < 0.0)
            u = -u;
        s = -1.0 / u;
        c = -t * s;
    }
}

// Eigen 3.4.0 numext::hypot for finite reals (used at TridiagEigen.h:64)
MISPEC_HD inline double eigen_hypot(double x, double y)
{
    x = fabs(x);
    y = fabs(y);
    const double p = x > y ? x : y;
    if (p == 0.0)
        return 0.0;
    const double qp = (x > y ? y : x) / p;
    return p * sqrt(1.0 + qp * qp);
}

// TridiagEigen.h:44-108
MISPEC_HD inline void tridiag_qr_step(double* diag, double* subdiag, int start, int end, double* Q, int ldq, int qrows,
                                      Lanes lanes)
{
    const double td = (diag[end - 1] - diag[end]) * 0.5;
    const double e = subdiag[end - 1];
    double mu = diag[end];
    if (td == 0.0)
        mu -= fabs(e);
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
    for (int k = start; k < end && z != 0.0; ++k)
    {
        double c, s;
        eigen_make_givens(x, z, c, s);
        const double dk = diag[k], sk = subdiag[k], dk1 = diag[k + 1];
        const double sdk = s * dk + c * sk;
        const double dkp1 = s * sk + c * dk1;
        diag[k] = c * (c * dk - s * sk) - s * (c * sk - s * dk1);
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
        // Q <- Q * G (applyOnTheRight): x' = c x - s y, y' = s x + c y
        double* qk = Q + (long) k * ldq;
        double* qk1 = qk + ldq;
        for (int i = lanes.first; i < qrows; i += lanes.stride)
        {
            const double xi = qk[i], yi = qk1[i];
            qk[i] = c * xi - s * yi;
            qk1[i] = s * xi + c * yi;
        }
    }
}

// TridiagEigen.h:121-210.  diag[n] / subd[n-1] in: the tridiagonal; out: diag = eigenvalues.
// Q (n x n, ldq) must hold the identity on entry and holds the eigenvectors on exit.
// Returns 0 on success, 1 if the iteration limit (30*n) was hit (the reference throws std::runtime_error).
MISPEC_HD inline int tridiag_eigen(int n, double* diag, double* subd, double* Q, int ldq, Lanes lanes)
{
    double scale = 0.0;
    for (int i = 0; i < n; i++)
        scale = fmax(scale, fabs(diag[i]));
    for (int i = 0; i < n - 1; i++)
        scale = fmax(scale, fabs(subd[i]));
    if (scale < kNear0)
    {
        for (int i = 0; i < n; i++)
            diag[i] = 0.0;
        return 0;
    }
    for (int i = 0; i < n; i++)
        diag[i] = diag[i] / scale;
    for (int i = 0; i < n - 1; i++)
        subd[i] = subd[i] / scale;

    int end = n - 1, start = 0, iter = 0, info = 0;
    const double considerAsZero = kMinPos;
    const double precision_inv = 1.0 / kEps;
    while (end > 0)
    {
        for (int i = start; i < end; i++)
        {
            if (fabs(subd[i]) <= considerAsZero)
                subd[i] = 0.0;
            else
            {
                const double scaled = precision_inv * subd[i];
                if (scaled * scaled <= (fabs(diag[i]) + fabs(diag[i + 1])))
                    subd[i] = 0.0;
            }
        }
        while (end > 0 && subd[end - 1] == 0.0)
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
        while (start > 0 && subd[start - 1] != 0.0)
            start--;
        tridiag_qr_step(diag, subd, start, end, Q, ldq, n, lanes);
    }
    for (int i = 0; i < n; i++)
        diag[i] *= scale;
    return info;
}

}  // namespace small
}  // namespace mispec
