// The ncv x ncv Hessenberg sweeps of the GENERAL (non-symmetric) implicit restart, written once for two targets like
// SmallDense.h: a single wavefront with H, Q and the work arrays resident in LDS (spectra_amd/csrc/small.hip:
// k_hess_restart), and the host with one "lane" (tests compare the two and both against internal/SmallDenseGen.h and the
// oracle).  SURVEY.md §8a row a19.
//
// Reference arithmetic followed (yixuan/spectra v1.2.0, include/Spectra/LinAlg/):
//   hess_shifted_qr_lanes   UpperHessenbergQR.h:136-195 (compute), :219-255 (matrix_QtHQ = RQ + sI), :383-417 (apply_YQ)
//   double_shift_qr_lanes   DoubleShiftQR.h:51-231 (reflectors, update_block), :334-438 (compute, matrix_QtHQ), :455-467 (apply_YQ)
//
// Parallel structure.  Every scalar recurrence (a Givens rotation, a 3-element Householder reflector) is evaluated
// redundantly by all lanes from LDS-resident entries; the O(n) updates they imply are split: ROW operations
// (G' * rows, P * rows) give lane l the columns l, l + stride, ...; COLUMN operations (columns * G, columns * P) give it the
// rows l, l + stride, ....  A lane only ever touches "its" columns in a row phase and "its" rows in a column phase, so a
// barrier is needed exactly where a phase reads what the other kind of phase (or another lane's scalar step) wrote.
#pragma once

#include "SmallDense.h"

namespace mispec {
namespace small {

// H - sI = QR by Givens rotations, H <- RQ + sI, Qacc <- Qacc * Q.   H: n x n (leading dimension ldh, entries below the
// sub-diagonal are treated as and set to zero), Qacc: qrows x n.   work: 3n doubles.
MISPEC_HD inline void hess_shifted_qr_lanes(int n, double* H, int ldh, double shift, double* Q, int ldq, int qrows, double* work,
                                            Lanes lanes)
{
#define MISPEC_HA(i, j) H[(long) (j) * ldh + (i)]
    double* rc = work;
    double* rs = work + n;
    double* rr = work + 2 * n;
    for (int i = lanes.first; i < n; i += lanes.stride)
    {
        MISPEC_HA(i, i) -= shift;
        for (int r = i + 2; r < n; r++)  // column i: below the sub-diagonal
            MISPEC_HA(r, i) = 0.0;
    }
    lanes.sync();
    // row phase: lane l owns columns l, l+stride, ...; the rotation of step i is read from column i (owner: lane i % stride)
    for (int i = 0; i < n - 1; i++)
    {
        double r, c, s;
        givens_rotation(MISPEC_HA(i, i), MISPEC_HA(i + 1, i), r, c, s);
        if (lanes.first == 0)
        {
            rc[i] = c;
            rs[i] = s;
            rr[i] = r;
        }
        for (int j = i + 1 + lanes.first; j < n; j += lanes.stride)  // rows i, i+1 <- G' * rows
        {
            const double t = MISPEC_HA(i, j);
            MISPEC_HA(i, j) = c * t - s * MISPEC_HA(i + 1, j);
            MISPEC_HA(i + 1, j) = s * t + c * MISPEC_HA(i + 1, j);
        }
        lanes.sync();  // column i+1 (rows i+1, i+2) is final before the next rotation reads it
    }
    for (int i = lanes.first; i < n - 1; i += lanes.stride)
    {
        MISPEC_HA(i, i) = rr[i];
        MISPEC_HA(i + 1, i) = 0.0;
    }
    lanes.sync();
    // column phase: lane l owns rows l, l+stride, ... of H and of Qacc; no dependence between lanes
    for (int i = 0; i < n - 1; i++)
    {
        const double c = rc[i], s = rs[i];
        double* a = H + (long) i * ldh;
        double* b = H + (long) (i + 1) * ldh;
        for (int j = lanes.first; j < i + 2; j += lanes.stride)
        {
            const double t = a[j];
            a[j] = c * t - s * b[j];
            b[j] = s * t + c * b[j];
        }
        double* qa = Q + (long) i * ldq;
        double* qb = Q + (long) (i + 1) * ldq;
        for (int j = lanes.first; j < qrows; j += lanes.stride)
        {
            const double t = qa[j];
            qa[j] = c * t - s * qb[j];
            qb[j] = s * t + c * qb[j];
        }
    }
    lanes.sync();
    for (int i = lanes.first; i < n; i += lanes.stride)
        MISPEC_HA(i, i) += shift;
    lanes.sync();
#undef MISPEC_HA
}

namespace detail {
// |(a, b, c)|_2 without overflow (DoubleShiftQR.h:66-87's stable norm)
MISPEC_HD inline double ds_norm3(double a, double b, double c)
{
    a = fabs(a);
    b = fabs(b);
    c = fabs(c);
    if (a < b)
    {
        const double t = a;
        a = b;
        b = t;
    }
    if (a < c)
    {
        const double t = a;
        a = c;
        c = t;
    }
    if (a < kNear0)
        return 0.0;
    const double r2 = b / a, r3 = c / a;
    const double cutoff = 0.1 * 1.220703125e-4;
    const double r = r2 * r2 + r3 * r3;
    return a * ((r2 >= cutoff || r3 >= cutoff) ? sqrt(1.0 + r) : (1.0 + r * (0.5 - 0.125 * r)));
}
// (x1, x2, x3) /= |x|, given |x1| largest and non-zero (DoubleShiftQR.h:92-117)
MISPEC_HD inline void ds_unit3(double& x1, double& x2, double& x3)
{
    const double sgn = (x1 > 0.0) ? 1.0 : -1.0;
    x1 = fabs(x1);
    const double r2 = x2 / x1, r3 = x3 / x1;
    const double cutoff = 0.1 * 1.220703125e-4;
    double r = r2 * r2 + r3 * r3;
    r = (fabs(r2) >= cutoff || fabs(r3) >= cutoff) ? 1.0 / sqrt(1.0 + r) : (1.0 - r * (0.5 - 0.375 * r));
    x1 = sgn * r;
    x2 = r2 * r;
    x3 = r3 * r;
}
// hypot as Eigen's numext::hypot evaluates it (positive_real_hypot)
MISPEC_HD inline double ds_hypot(double x, double y)
{
    x = fabs(x);
    y = fabs(y);
    const double p = x > y ? x : y;
    if (p == 0.0)
        return 0.0;
    const double qp = (x > y ? y : x) / p;
    return p * sqrt(1.0 + qp * qp);
}
// The reflector P = I - 2 v v' with P x = +-|x| e1 (DoubleShiftQR.h:121-171).  Returns the number of rows it touches
// (3 general, 2 Givens-like, 1 identity) and v.
MISPEC_HD inline int ds_reflector(double x1, double x2, double x3, double v[3])
{
    const double a2 = fabs(x2), a3 = fabs(x3);
    v[0] = v[1] = v[2] = 0.0;
    if (a2 < kNear0 && a3 < kNear0)
        return 1;
    const int nr = (a3 < kNear0) ? 2 : 3;
    const double nrm = (a3 < kNear0) ? ds_hypot(x1, x2) : ds_norm3(x1, x2, x3);
    const double rho = (x1 <= 0.0) ? 1.0 : -1.0;
    const double y1 = x1 - rho * nrm, a1 = fabs(y1);
    v[0] = y1;
    v[1] = x2;
    v[2] = x3;
    if (a1 >= a2 && a1 >= a3)
        ds_unit3(v[0], v[1], v[2]);
    else if (a2 >= a1 && a2 >= a3)
        ds_unit3(v[1], v[0], v[2]);
    else
        ds_unit3(v[2], v[0], v[1]);
    return nr;
}
// rows r0.. (nrow of them) of columns [c0, c0 + ncol) of H: X <- X - 2 v (v'X); this lane's columns
MISPEC_HD inline void ds_left(double* H, int ldh, int r0, int c0, int nrow, int ncol, int nr, const double v[3], Lanes lanes)
{
    if (nr == 1)
        return;
    const bool two = (nr == 2 || nrow == 2);
    for (int j = lanes.first; j < ncol; j += lanes.stride)
    {
        double* x = H + (long) (c0 + j) * ldh + r0;
        const double d = 2.0 * v[0] * x[0] + 2.0 * v[1] * x[1] + (two ? 0.0 : 2.0 * v[2] * x[2]);
        x[0] -= d * v[0];
        x[1] -= d * v[1];
        if (!two)
            x[2] -= d * v[2];
    }
}
// rows [0, nrow) of columns c0.. (ncol of them) of M: X <- X - 2 (X v) v'; this lane's rows
MISPEC_HD inline void ds_right(double* M, int ldm, int c0, int nrow, int ncol, int nr, const double v[3], Lanes lanes)
{
    if (nr == 1)
        return;
    const bool two = (nr == 2 || ncol == 2);
    double* x0 = M + (long) c0 * ldm;
    double* x1 = M + (long) (c0 + 1) * ldm;
    double* x2 = two ? x1 : M + (long) (c0 + 2) * ldm;
    for (int i = lanes.first; i < nrow; i += lanes.stride)
    {
        const double d = 2.0 * v[0] * x0[i] + 2.0 * v[1] * x1[i] + (two ? 0.0 : 2.0 * v[2] * x2[i]);
        x0[i] -= d * v[0];
        x1[i] -= d * v[1];
        if (!two)
            x2[i] -= d * v[2];
    }
}
}  // namespace detail

// Francis double-shift step: H^2 - s H + t I = QR implicitly; H <- Q'HQ, Qacc <- Qacc * Q.
//   work: 3n doubles (the reflector vectors);  iwork: 2n + 2 ints (rows per reflector, block boundaries)
MISPEC_HD inline void double_shift_qr_lanes(int n, double* H, int ldh, double s, double t, double* Q, int ldq, int qrows, double* work,
                                            int* iwork, Lanes lanes)
{
#define MISPEC_HA(i, j) H[(long) (j) * ldh + (i)]
    using namespace detail;
    double* u = work;            // 3 x n
    int* nrows = iwork;          // n
    int* cuts = iwork + n;       // <= n + 1 block starts (+ n at the end)
    const double eps_abs = kNear0 * (double(n) / kEps);
    // deflation (DoubleShiftQR.h:358-381): independent per sub-diagonal entry; lower part cleared
    for (int i = lanes.first; i < n - 1; i += lanes.stride)
    {
        const double h = fabs(MISPEC_HA(i + 1, i));
        const double d = fabs(MISPEC_HA(i, i)) + fabs(MISPEC_HA(i + 1, i + 1));
        if (h <= eps_abs || h <= kEps * d)
            MISPEC_HA(i + 1, i) = 0.0;
        for (int r = i + 2; r < n; r++)
            MISPEC_HA(r, i) = 0.0;
    }
    for (int i = lanes.first; i < n; i += lanes.stride)
        nrows[i] = 0;
    lanes.sync();
    int ncuts = 0;  // every lane builds the same list (registers would do; the array keeps it simple)
    if (lanes.first == 0)
    {
        cuts[ncuts++] = 0;
        for (int i = 0; i < n - 1; i++)
            if (MISPEC_HA(i + 1, i) == 0.0)
                cuts[ncuts++] = i + 1;
        cuts[ncuts++] = n;
        cuts[n + 1] = ncuts;
    }
    lanes.sync();
    ncuts = cuts[n + 1];
    for (int blk = 0; blk + 1 < ncuts; blk++)
    {
        const int il = cuts[blk], iu = cuts[blk + 1] - 1;
        const int bs = iu - il + 1;
        double v[3];
        if (bs == 1)
        {
            if (lanes.first == 0)
                nrows[il] = 1;
            continue;
        }
        const double x00 = MISPEC_HA(il, il), x01 = MISPEC_HA(il, il + 1), x10 = MISPEC_HA(il + 1, il), x11 = MISPEC_HA(il + 1, il + 1);
        const double m00 = x00 * (x00 - s) + x01 * x10 + t;
        const double m10 = x10 * (x00 + x11 - s);
        if (bs == 2)
        {
            lanes.sync();  // everybody has read the block before it changes
            const int nr = ds_reflector(m00, m10, 0.0, v);
            if (lanes.first == 0)
            {
                nrows[il] = nr;
                u[3 * il] = v[0];
                u[3 * il + 1] = v[1];
                u[3 * il + 2] = v[2];
                nrows[il + 1] = 1;
            }
            ds_left(H, ldh, il, il, 2, n - il, nr, v, lanes);
            lanes.sync();
            ds_right(H, ldh, il, il + 2, 2, nr, v, lanes);
            lanes.sync();
            continue;
        }
        const double m20 = MISPEC_HA(il + 2, il + 1) * MISPEC_HA(il + 1, il);
        lanes.sync();
        int nr = ds_reflector(m00, m10, m20, v);
        if (lanes.first == 0)
        {
            nrows[il] = nr;
            u[3 * il] = v[0];
            u[3 * il + 1] = v[1];
            u[3 * il + 2] = v[2];
        }
        ds_left(H, ldh, il, il, 3, n - il, nr, v, lanes);
        lanes.sync();
        ds_right(H, ldh, il, il + (bs < 4 ? bs : 4), 3, nr, v, lanes);
        lanes.sync();
        for (int i = 1; i < bs - 2; i++)
        {
            const double* x = &MISPEC_HA(il + i, il + i - 1);
            const double x1 = x[0], x2 = x[1], x3 = x[2];
            lanes.sync();
            nr = ds_reflector(x1, x2, x3, v);
            if (lanes.first == 0)
            {
                nrows[il + i] = nr;
                u[3 * (il + i)] = v[0];
                u[3 * (il + i) + 1] = v[1];
                u[3 * (il + i) + 2] = v[2];
            }
            ds_left(H, ldh, il + i, il + i - 1, 3, n - il - i + 1, nr, v, lanes);
            lanes.sync();
            ds_right(H, ldh, il + i, il + (bs < i + 4 ? bs : i + 4), 3, nr, v, lanes);
            lanes.sync();
        }
        {
            const double x1 = MISPEC_HA(iu - 1, iu - 2), x2 = MISPEC_HA(iu, iu - 2);
            lanes.sync();
            nr = ds_reflector(x1, x2, 0.0, v);
            if (lanes.first == 0)
            {
                nrows[iu - 1] = nr;
                u[3 * (iu - 1)] = v[0];
                u[3 * (iu - 1) + 1] = v[1];
                u[3 * (iu - 1) + 2] = v[2];
                nrows[iu] = 1;
            }
            ds_left(H, ldh, iu - 1, iu - 2, 2, n - iu + 2, nr, v, lanes);
            lanes.sync();
            ds_right(H, ldh, iu - 1, il + bs, 2, nr, v, lanes);
            lanes.sync();
        }
    }
    lanes.sync();
    // second deflation pass (DoubleShiftQR.h:421-424)
    for (int i = lanes.first; i < n - 1; i += lanes.stride)
    {
        const double h = fabs(MISPEC_HA(i + 1, i));
        const double d = fabs(MISPEC_HA(i, i)) + fabs(MISPEC_HA(i + 1, i + 1));
        if (h <= eps_abs || h <= kEps * d)
            MISPEC_HA(i + 1, i) = 0.0;
    }
    // Qacc <- Qacc * P0 * P1 * ... (apply_YQ, DoubleShiftQR.h:455-467): row-parallel, no dependence between lanes
    for (int i = 0; i < n - 2; i++)
    {
        const double v3[3] = {u[3 * i], u[3 * i + 1], u[3 * i + 2]};
        ds_right(Q, ldq, i, qrows, 3, nrows[i], v3, lanes);
    }
    if (n >= 2)
    {
        const double v3[3] = {u[3 * (n - 2)], u[3 * (n - 2) + 1], u[3 * (n - 2) + 2]};
        ds_right(Q, ldq, n - 2, qrows, 2, nrows[n - 2], v3, lanes);
    }
    lanes.sync();
#undef MISPEC_HA
}

}  // namespace small
}  // namespace mispec
