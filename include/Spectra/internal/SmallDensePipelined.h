// The (m - k) shifted QR sweeps of one implicit restart as a SKEWED PIPELINE.
//
// The reference applies the shifts one after the other (HermEigsBase.h:124-147: for each shift TridiagQR::compute,
// apply_YQ, matrix_QtHQ — UpperHessenbergQR.h:515-693): p sweeps of m - 1 rotations, every rotation waiting for the one
// before it: p (m - 1) serial Givens rotations (702 at m = 40, p = 18).  All p shifts are known before the first sweep
// starts, and sweep s + 1 needs of sweep s's result only the entries that are already final three rotations behind
// sweep s's front:
//     rotation j of a sweep needs   e_j, d_{j+1}      of its input T      (r_supd is formed one rotation later),
//     Q'TQ row j - 1 needs          rotation j                            (UpperHessenbergQR.h:668-674),
//     the deflation of e'_{j-2}     d'_{j-2}, d'_{j-1}                    (:684-692).
// So at "tick" t sweep s works on rotation j = t - 3 s - 1, emits d'_{j-1} and e'_{j-2}, and sweep s + 1 picks both up at
// the next tick: m + 2 + 3 (p - 1) ticks instead of p (m - 1) serial rotations, every sweep executing exactly the
// operations of tridiag_shifted_qr (SmallDense.h) on exactly the same operands in the same order — bit-identical T and Q.
// The rotations of one tick act on disjoint column pairs of Q (their indices differ by 3), and every column of Q meets
// its rotations in the reference's order, so Q can be accumulated tick by tick (device: a second wavefront, one barrier
// per tick) or sweep by sweep afterwards (host: a SIMD loop over the rows) with identical bits.
//
// Written once for two targets, like SmallDense.h:
//   * host  — restart_sweeps_pipelined(): the lanes are a loop (the out-of-order core overlaps the independent chains
//             of the sweeps in flight), then Q in one vectorisable pass;
//   * device — k_restart_pipelined (spectra_amd/csrc/small.hip): lane s of wave 0 owns sweep s, __shfl_up hands the
//             emitted pair to lane s + 1, waves 1.. rotate the columns of Q in LDS.
// The device cannot call the host's libm: hypot_glibc below restates glibc 2.35's e_hypot.c (the non-FMA kernel, the
// one x86-64 runs) operation by operation; tests/test_small_pipelined.py holds it against std::hypot on the host and
// tests/test_gpu_small.py holds the device kernel against the host routine bit for bit.
#pragma once

#include "SmallDense.h"

namespace mispec {
namespace small {

#if defined(__clang__)
#define MISPEC_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define MISPEC_NO_CONTRACT
#endif

// glibc 2.35 sysdeps/ieee754/dbl-64/e_hypot.c for finite arguments (C. F. Borges' corrected algorithm without FMA):
// the value std::hypot returns on the hosts this library and the reference run on.
MISPEC_HD inline double hypot_glibc(double x, double y)
{
    MISPEC_NO_CONTRACT
    x = fabs(x);
    y = fabs(y);
    double ax = x < y ? y : x;
    double ay = x < y ? x : y;
    double scale = 1.0;
    if (ax > 0x1p+511)  // LARGE_VAL
    {
        if (ay <= ax * 0x1p-54)
            return ax + ay;
        ax *= 0x1p-600;
        ay *= 0x1p-600;
        scale = 0x1p+600;
    }
    else if (ay < 0x1p-459)  // TINY_VAL
    {
        if (ax >= ay * 0x1p+54)
            return ax + ay;
        ax *= 0x1p+600;
        ay *= 0x1p+600;
        scale = 0x1p-600;
    }
    else if (ax >= ay * 0x1p+54)
        return ax + ay;
    double h = sqrt(ax * ax + ay * ay);
    double t1, t2;
    if (h <= 2.0 * ay)
    {
        const double delta = h - ay;
        t1 = ax * (2.0 * delta - ax);
        t2 = (delta - 2.0 * (ax - ay)) * delta;
    }
    else
    {
        const double delta = h - ax;
        t1 = 2.0 * delta * (ax - 2.0 * ay);
        t2 = (4.0 * delta - ay) * ay + delta * delta;
    }
    h -= (t1 + t2) / (2.0 * h);
    return h * scale;
}

// givens_rotation of SmallDense.h (Givens.h:149-206, StableScaling :28-86) with the hypot both targets agree on, and
// without fused multiply-adds (the host build has none).
MISPEC_HD inline void givens_rotation_exact(double x, double y, double& r, double& c, double& s)
{
    MISPEC_NO_CONTRACT
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
    const bool xbig = xabs >= yabs;
    const double a = xbig ? xabs : yabs, b = xbig ? yabs : xabs;
    const double t = b / a;
    const double cutoff = 0.1 * 1.220703125e-4;  // 0.1 * eps^(1/4)
    double ca, sb;                               // a / r, b / r
    if (t >= cutoff)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        r = hypot_glibc(a, b);
#else
        r = hypot(a, b);
#endif
        ca = a / r;
        sb = b / r;
    }
    else
    {
        const double t2 = t * t;
        ca = 1.0 - t2 * (0.5 - t2 * (0.375 - 0.3125 * t2));
        sb = t * ca;
        r = a + 0.5 * b * t * (1.0 - t2 * (0.25 - 0.125 * t2));
    }
    c = xsign * (xbig ? ca : sb);
    s = -ysign * (xbig ? sb : ca);
}

// One sweep in flight.  in_d / in_e: what the sweep reads at its next tick (d_{j+1}, e_j of its input T);
// out_d / out_e: what it emitted at its last tick (d'_{j-1}, e'_{j-2} of its result).
struct SweepLane
{
    double mu;
    double dj;            // d_j of the input (z of Q'TQ row j - 1)
    double r_diag;        // R[j, j] of T - mu I so far
    double c_prev, s_prev;
    double x, y;          // Q'TQ: the running diag[i], subd[i]
    double dp_prev;       // d'_{j-2}
    double e_pre;         // e'_{j-2} before its deflation test
    double out_d, out_e;
};

// Tick of one sweep at local index j in [-1, n]: din = d_{j+1} (j + 1 <= n - 1), ein = e_j (0 <= j <= n - 2) of the
// sweep's input; rot_c / rot_s: the sweep's rotations (n - 1 each).  Emits out_d = d'_{j-1} (j >= 1) and
// out_e = e'_{j-2} (j >= 2).  Statement for statement tridiag_shifted_qr: :541-590 (B), :627-680 (C), :684-692 (D).
MISPEC_HD inline void sweep_tick(SweepLane& L, int n, int j, double din, double ein, double* rot_c, double* rot_s)
{
    MISPEC_NO_CONTRACT
    const int n2 = n - 2;
    if (j < 0)
    {
        L.r_diag = din - L.mu;
        L.x = din;
        L.dj = din;
        return;
    }
    double c = 0.0, s = 0.0;
    if (j <= n2)  // B(j)
    {
        const double r_supd = (j == 0) ? ein : L.c_prev * ein;
        double r;
        givens_rotation_exact(L.r_diag, ein, r, c, s);
        rot_c[j] = c;
        rot_s[j] = s;
        L.r_diag = s * r_supd + c * (din - L.mu);
        if (j == 0)
            L.y = ein;
    }
    double dp = 0.0;  // d'_{j-1}
    if (j >= 1)       // C(j - 1)
    {
        const int i = j - 1;
        const double ci = L.c_prev, si = L.s_prev;
        const double x = L.x, y = L.y, z = L.dj;
        const double cs = ci * si, c2 = ci * ci, s2 = si * si;
        const double c2x = c2 * x, s2x = s2 * x, c2z = c2 * z, s2z = s2 * z;
        const double csy2 = 2.0 * ci * si * y;
        dp = c2x - csy2 + s2z;
        double ny = cs * (x - z) + (c2 - s2) * y;
        const double nz = s2x + csy2 + c2z;
        double nw = 0.0;
        if (i < n2)
        {
            const double o = -si * ein;
            nw = ein * ci;
            ny = c * ny - s * o;
        }
        if (j >= 2)  // D(j - 2)
        {
            const double dsum = fabs(L.dp_prev) + fabs(dp);
            L.out_e = (fabs(L.e_pre) <= kEps * dsum) ? 0.0 : L.e_pre;
        }
        L.out_d = dp;
        L.dp_prev = dp;
        L.e_pre = ny;
        L.x = nz;
        L.y = nw;
    }
    L.c_prev = c;
    L.s_prev = s;
    L.dj = din;
}
// The two ticks behind the last rotation: j = n - 1 ran C(n - 2) above (x now holds d'_{n-1}); j = n only deflates
// e'_{n-2} and emits d'_{n-1}.
MISPEC_HD inline void sweep_tail(SweepLane& L)
{
    MISPEC_NO_CONTRACT
    const double dp = L.x;
    const double dsum = fabs(L.dp_prev) + fabs(dp);
    L.out_e = (fabs(L.e_pre) <= kEps * dsum) ? 0.0 : L.e_pre;
    L.out_d = dp;
}

// Host: all nshift sweeps on (diag[n], subd[n-1]) -> Q'TQ in place, the rotations of sweep s in rot_c / rot_s [s * n + j].
// work: (2 * nshift + 2) * n doubles (rot_c = work, rot_s = work + nshift * n).  Same T, bit for bit, as nshift calls of
// tridiag_shifted_qr.
inline void restart_rotations_pipelined(int n, double* diag, double* subd, const double* shifts, int nshift, double* work,
                                        SweepLane* lanes)
{
    if (nshift <= 0 || n < 2)
        return;  // (n == 1: no rotation — the serial routine's loops are empty)
    const int n1 = n - 1;
    double* rot_c = work;                        // [nshift][n]
    double* rot_s = work + size_t(nshift) * n;   // [nshift][n]
    double* e0 = work + size_t(2) * nshift * n;  // deflated input sub-diagonal (:526-539)
    double* dout = e0 + n;                       // the last sweep's emissions
    for (int i = 0; i < n1; i++)
    {
        const double e = subd[i];
        e0[i] = (fabs(e) <= kEps * (fabs(diag[i]) + fabs(diag[i + 1]))) ? 0.0 : e;
    }
    for (int s = 0; s < nshift; s++)
    {
        lanes[s] = SweepLane{};
        lanes[s].mu = shifts[s];
    }
    const int last = nshift - 1;
    const int ticks = (n + 2) + 3 * last;  // local j runs from -1 to n
    for (int t = 0; t < ticks; t++)
    {
        // descending: sweep s + 1 reads what sweep s emitted at the previous tick before sweep s overwrites it
        int s_hi = t / 3;
        if (s_hi > last)
            s_hi = last;
        for (int s = s_hi; s >= 0; s--)
        {
            const int j = t - 3 * s - 1;
            if (j > n)
                break;  // this sweep and all earlier ones are finished
            SweepLane& L = lanes[s];
            if (j == n)
                sweep_tail(L);
            else
            {
                double din = 0.0, ein = 0.0;
                if (s == 0)
                {
                    if (j + 1 <= n1)
                        din = diag[j + 1];
                    if (j >= 0 && j <= n - 2)
                        ein = e0[j];
                }
                else
                {
                    din = lanes[s - 1].out_d;
                    ein = lanes[s - 1].out_e;
                }
                sweep_tick(L, n, j, din, ein, rot_c + size_t(s) * n, rot_s + size_t(s) * n);
            }
            if (s == last)
            {
                if (j >= 1)
                    dout[j - 1] = L.out_d;  // d'_{j-1}
                if (j >= 2)
                    subd[j - 2] = L.out_e;  // e'_{j-2}
            }
        }
    }
    for (int i = 0; i < n; i++)
        diag[i] = dout[i];
}

// apply_YQ (UpperHessenbergQR.h:403-416) for all sweeps: Q <- Q * Q_1 ... Q_p on a block of W * NV rows, the running column
// carried in registers from one rotation to the next.  V: a vector of W doubles (plain mul / sub / add per element — the
// same three operations per entry as the scalar loop, no fused multiply-add).
typedef double v2d_t __attribute__((vector_size(16)));
typedef double v4d_t __attribute__((vector_size(32)));

template <class V, int NV>
inline __attribute__((always_inline)) void rotate_row_block(double* Qb, int ldq, int n, const double* rot_c, const double* rot_s,
                                                            int nshift)
{
    MISPEC_NO_CONTRACT
    constexpr int W = int(sizeof(V) / sizeof(double));
    const int n1 = n - 1;
    for (int s = 0; s < nshift; s++)
    {
        const double* rc = rot_c + size_t(s) * n;
        const double* rs = rot_s + size_t(s) * n;
        V carry[NV];
        for (int k = 0; k < NV; k++)
            __builtin_memcpy(&carry[k], Qb + k * W, sizeof(V));  // (an unaligned vector move)
        for (int i = 0; i < n1; i++)
        {
            const double c = rc[i], sn = rs[i];
            double* Yi = Qb + (long) i * ldq;
            double* Yi1 = Yi + ldq;
            for (int k = 0; k < NV; k++)
            {
                V qb;
                __builtin_memcpy(&qb, Yi1 + k * W, sizeof(V));
                const V qa = carry[k];
                const V na = c * qa - sn * qb;
                __builtin_memcpy(Yi + k * W, &na, sizeof(V));
                carry[k] = sn * qa + c * qb;
            }
        }
        double* Yn = Qb + (long) n1 * ldq;
        for (int k = 0; k < NV; k++)
            __builtin_memcpy(Yn + k * W, &carry[k], sizeof(V));
    }
}
template <class V>
inline __attribute__((always_inline)) void rotate_rows(double* Q, int ldq, int rows8, int n, const double* rot_c, const double* rot_s,
                                                       int nshift)
{
    constexpr int W = int(sizeof(V) / sizeof(double));
    int r = 0;
    for (; r + 8 * W <= rows8; r += 8 * W)
        rotate_row_block<V, 8>(Q + r, ldq, n, rot_c, rot_s, nshift);
    for (; r + 4 * W <= rows8; r += 4 * W)
        rotate_row_block<V, 4>(Q + r, ldq, n, rot_c, rot_s, nshift);
    for (; r + 8 <= rows8; r += 8)
        rotate_row_block<V, 8 / W>(Q + r, ldq, n, rot_c, rot_s, nshift);
}
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("avx"))) inline void rotate_rows_avx(double* Q, int ldq, int rows8, int n, const double* rot_c,
                                                           const double* rot_s, int nshift)
{
    rotate_rows<v4d_t>(Q, ldq, rows8, n, rot_c, rot_s, nshift);
}
#endif
// Q: ldq >= rows8, rows8 a multiple of 8 (rows beyond the matrix: zero padding, rotated along — 0 stays 0).
inline void apply_sweeps_to_Q(double* Q, int ldq, int rows8, int n, const double* rot_c, const double* rot_s, int nshift)
{
    if (nshift <= 0 || n < 2)
        return;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    static const bool have_avx = __builtin_cpu_supports("avx");
    if (have_avx)
    {
        rotate_rows_avx(Q, ldq, rows8, n, rot_c, rot_s, nshift);
        return;
    }
#endif
    rotate_rows<v2d_t>(Q, ldq, rows8, n, rot_c, rot_s, nshift);
}
}  // namespace small
}  // namespace mispec
