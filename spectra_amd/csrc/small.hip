// ncv x ncv kernels of the implicit restart: one workgroup, everything in LDS — one wavefront each, except
// k_restart_pipelined (round 6: the shifted QR sweeps as a skewed pipeline over four wavefronts, the host's bits).
// They are latency-bound (no roofline) and SLOWER than a host core (serial chains of fp64 divisions and square roots:
// 110 us pipelined / 380 us on one wavefront against 13 us on the core at ncv = 40, profiles/r11g): the default runs
// the host routines, option small=device these kernels (H, the rotations and Q then never leave the device between
// the factorisation and compress_V).
#include "small.hpp"

#include <Spectra/internal/SmallDense.h>
#include <Spectra/internal/SmallDensePipelined.h>
#include <Spectra/internal/SmallDenseGenLanes.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <vector>

using namespace mispec;

namespace {

// diag/subd: in = tridiagonal H, out = eigenvalues (diag).  evecs: n x n column-major.
__global__ __launch_bounds__(64) void k_tridiag_eigen(int n, const double* __restrict__ diag_in,
                                                       const double* __restrict__ subd_in, double* __restrict__ evals,
                                                       double* __restrict__ evecs, int* __restrict__ info)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* diag = sm;
    double* subd = sm + n;
    double* Q = sm + 2 * n;
    const int lane = threadIdx.x;
    for (int i = lane; i < n; i += 64)
    {
        diag[i] = diag_in[i];
        subd[i] = (i < n - 1) ? subd_in[i] : 0.0;
    }
    for (int idx = lane; idx < n * n; idx += 64)
        Q[idx] = (idx / n == idx % n) ? 1.0 : 0.0;
    __syncthreads();
    const int rc = small::tridiag_eigen(n, diag, subd, Q, n, small::Lanes{lane, 64});
    __syncthreads();
    for (int i = lane; i < n; i += 64)
        evals[i] = diag[i];
    for (int idx = lane; idx < n * n; idx += 64)
        evecs[idx] = Q[idx];
    if (lane == 0)
        *info = rc;
}

// Applies nshift shifted-QR steps to the tridiagonal (diag, subd) and accumulates Q (m x m).
__global__ __launch_bounds__(64) void k_restart_sym(int m, double* __restrict__ diag_io, double* __restrict__ subd_io,
                                                     ShiftList shifts, int nshift, double* __restrict__ Qout)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* diag = sm;
    double* subd = sm + m;
    double* work = sm + 2 * m;  // 4m
    double* Q = sm + 6 * m;     // m*m
    const int lane = threadIdx.x;
    for (int i = lane; i < m; i += 64)
    {
        diag[i] = diag_io[i];
        subd[i] = (i < m - 1) ? subd_io[i] : 0.0;
    }
    for (int idx = lane; idx < m * m; idx += 64)
        Q[idx] = (idx / m == idx % m) ? 1.0 : 0.0;
    __syncthreads();
    for (int s = 0; s < nshift; s++)
        small::tridiag_shifted_qr(m, diag, subd, shifts.mu[s], Q, m, m, work, small::Lanes{lane, 64});
    __syncthreads();
    for (int i = lane; i < m; i += 64)
    {
        diag_io[i] = diag[i];
        if (i < m - 1)
            subd_io[i] = subd[i];
    }
    for (int idx = lane; idx < m * m; idx += 64)
        Qout[idx] = Q[idx];
}

// General restart (a19): the shifts of one implicit restart of the Arnoldi process applied to the m x m upper Hessenberg
// H — a real shift is UpperHessenbergQR::compute + matrix_QtHQ (UpperHessenbergQR.h:136-255), a conjugate pair one Francis
// double-shift step (DoubleShiftQR.h:334-438) — with Q <- Q * Q_i accumulated (apply_YQ) for the V*Q kernel.  One
// wavefront; H, Q (leading dimension m|1: row walks hit distinct banks) and the work arrays in LDS; the arithmetic is
// internal/SmallDenseGenLanes.h, the same source the host compiles.
__global__ __launch_bounds__(64) void k_hess_restart(int m, int ld, double* __restrict__ H_io, GenShiftList shifts, double* __restrict__ Qout)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* H = sm;
    double* Q = sm + size_t(m) * ld;
    double* work = Q + size_t(m) * ld;  // 3m
    int* iwork = reinterpret_cast<int*>(work + 3 * m);  // 2m + 2
    const int lane = threadIdx.x;
    for (int idx = lane; idx < m * m; idx += 64)
    {
        const int i = idx % m, j = idx / m;
        H[size_t(j) * ld + i] = H_io[idx];
        Q[size_t(j) * ld + i] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    const small::Lanes lanes{lane, 64};
    for (int s = 0; s < shifts.count; s++)
    {
        if (shifts.kind[s] == 0)
            small::hess_shifted_qr_lanes(m, H, ld, shifts.a[s], Q, ld, m, work, lanes);
        else
            small::double_shift_qr_lanes(m, H, ld, shifts.a[s], shifts.b[s], Q, ld, m, work, iwork, lanes);
    }
    __syncthreads();
    for (int idx = lane; idx < m * m; idx += 64)
    {
        const int i = idx % m, j = idx / m;
        H_io[idx] = H[size_t(j) * ld + i];
        Qout[idx] = Q[size_t(j) * ld + i];
    }
}

// ===================================================================================================
// Register-resident variants for m <= 64 (one wavefront, lane i <-> row/entry i).
//
// The generic kernels above keep diag / subdiag in LDS, so every step of the scalar recurrences pays LDS
// round trips on its critical path.  Here lane i holds diag[i] and subdiag[i] in registers; a scalar
// read is a v_readlane, a scalar write a v_cndmask, the deflation scans are one ballot, and the chain
// d[k], e[k] -> rotation -> d[k+1], e[k+1] is carried in registers from one rotation to the next.  Only the
// accumulated Q lives in LDS, and of its two active columns one stays in registers between rotations.
// Same arithmetic as internal/SmallDense.h (tridiag_eigen, tridiag_shifted_qr), except that
// stable_scaling's hypot(a, b) is evaluated as a * sqrt(1 + (b/a)^2) (within 1 ulp of it).
// ===================================================================================================
__device__ __forceinline__ double rdlane(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void wrlane(double& reg, int lane, int target, double value)
{
    reg = (lane == target) ? value : reg;
}
__device__ __forceinline__ double lane_below(double v)  // value of lane+1 (0 past the end)
{
    return __shfl_down(v, 1, 64);
}
__device__ __forceinline__ int highest_bit(unsigned long long m) { return 63 - __clzll((long long) m); }

__device__ __forceinline__ void givens_fast(double x, double y, double& r, double& c, double& s)
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
    const bool xbig = xabs >= yabs;
    const double a = xbig ? xabs : yabs, b = xbig ? yabs : xabs;
    const double t = b / a;
    double ca, sb;  // a/r, b/r
    if (t >= 0.1 * 0x1p-13)
    {
        r = a * sqrt(1.0 + t * t);
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

// Q <- Q * G over columns (k, k+1) with column k carried in `qa` (this lane's row); returns the new carried column.
__device__ __forceinline__ double rotate_cols(double* Q, int n, int lane, int k, double c, double s, double qa)
{
    double qb = 0.0;
    if (lane < n)
        qb = Q[(k + 1) * n + lane];
    const double na = c * qa - s * qb;
    const double nb = s * qa + c * qb;
    if (lane < n)
        Q[k * n + lane] = na;
    return nb;
}

__global__ __launch_bounds__(64) void k_tridiag_eigen_w64(int n, const double* __restrict__ diag_in,
                                                           const double* __restrict__ subd_in, double* __restrict__ evals,
                                                           double* __restrict__ evecs, int* __restrict__ info)
{
    extern __shared__ __attribute__((aligned(16))) double Q[];  // n x n
    const int lane = threadIdx.x;
    double d = (lane < n) ? diag_in[lane] : 0.0;
    double e = (lane < n - 1) ? subd_in[lane] : 0.0;
    for (int idx = lane; idx < n * n; idx += 64)
        Q[idx] = (idx / n == idx % n) ? 1.0 : 0.0;
    __syncthreads();

    // scale by the largest magnitude (TridiagEigen.h:139-152)
    double scale = fmax(fabs(d), fabs(e));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        scale = fmax(scale, __shfl_xor(scale, off, 64));
    int rc = 0;
    if (scale < small::kNear0)
        d = 0.0;
    else
    {
        d = d / scale;
        e = e / scale;
        int end = n - 1, start = 0, iter = 0;
        const double precision_inv = 1.0 / small::kEps;
        while (end > 0)
        {
            // deflation scan over [start, end) (TridiagEigen.h:167-178), all lanes at once
            const double dn = lane_below(d);
            const bool inrange = lane >= start && lane < end;
            const double sc = precision_inv * e;
            if (inrange && (fabs(e) <= small::kMinPos || sc * sc <= (fabs(d) + fabs(dn))))
                e = 0.0;
            const unsigned long long nz = __ballot(e != 0.0 && lane < n - 1);
            const unsigned long long below_end = nz & ((end >= 64) ? ~0ull : ((1ull << end) - 1ull));
            end = below_end ? highest_bit(below_end) + 1 : 0;  // largest unreduced block at the end (:181-182)
            if (end <= 0)
                break;
            iter++;
            if (iter > 30 * n)
            {
                rc = 1;
                break;
            }
            const unsigned long long zeros = ~nz & ((1ull << (end - 1)) - 1ull);  // e[j] == 0, j < end-1
            start = zeros ? highest_bit(zeros) + 1 : 0;                         // (:195-197)

            // one implicit QR step with Wilkinson shift on [start, end] (TridiagEigen.h:44-108)
            const double d_end1 = rdlane(d, end - 1), d_end = rdlane(d, end), e_end1 = rdlane(e, end - 1);
            const double td = (d_end1 - d_end) * 0.5;
            double mu = d_end;
            if (td == 0.0)
                mu -= fabs(e_end1);
            else if (e_end1 != 0.0)
            {
                const double e2 = e_end1 * e_end1;
                const double h = small::eigen_hypot(td, e_end1);
                if (e2 == 0.0)
                    mu -= e_end1 / ((td + (td > 0.0 ? h : -h)) / e_end1);
                else
                    mu -= e2 / (td + (td > 0.0 ? h : -h));
            }
            double dk = rdlane(d, start), ek = rdlane(e, start), ekm1 = 0.0;
            double x = dk - mu, z = ek;
            double qa = (lane < n) ? Q[start * n + lane] : 0.0;
            int k = start;
            for (; k < end && z != 0.0; ++k)
            {
                double c, s;
                small::eigen_make_givens(x, z, c, s);
                const double dk1 = rdlane(d, k + 1);
                const double ek1 = (k < end - 1) ? rdlane(e, k + 1) : 0.0;
                const double sdk = s * dk + c * ek;
                const double dkp1 = s * ek + c * dk1;
                const double ndk = c * (c * dk - s * ek) - s * (c * ek - s * dk1);
                const double ndk1 = s * sdk + c * dkp1;
                const double nek = c * sdk - s * dkp1;
                if (k > start)
                    wrlane(e, lane, k - 1, c * ekm1 - s * z);
                wrlane(d, lane, k, ndk);
                wrlane(d, lane, k + 1, ndk1);
                wrlane(e, lane, k, nek);
                x = nek;
                double nek1 = ek1;
                if (k < end - 1)
                {
                    z = -s * ek1;
                    nek1 = c * ek1;
                    wrlane(e, lane, k + 1, nek1);
                }
                dk = ndk1;
                ek = nek1;
                ekm1 = nek;
                qa = rotate_cols(Q, n, lane, k, c, s, qa);
            }
            if (lane < n)
                Q[k * n + lane] = qa;  // the carried column goes back to LDS
        }
        d *= scale;
    }
    __syncthreads();
    if (lane < n)
        evals[lane] = d;
    for (int idx = lane; idx < n * n; idx += 64)
        evecs[idx] = Q[idx];
    if (lane == 0)
        *info = rc;
}

__global__ __launch_bounds__(64) void k_restart_sym_w64(int m, double* __restrict__ diag_io, double* __restrict__ subd_io,
                                                         ShiftList shifts, int nshift, double* __restrict__ Qout)
{
    extern __shared__ __attribute__((aligned(16))) double Q[];  // m x m
    const int lane = threadIdx.x;
    double d = (lane < m) ? diag_io[lane] : 0.0;
    double e = (lane < m - 1) ? subd_io[lane] : 0.0;
    for (int idx = lane; idx < m * m; idx += 64)
        Q[idx] = (idx / m == idx % m) ? 1.0 : 0.0;
    __syncthreads();
    const int n1 = m - 1, n2 = m - 2;
    for (int sh = 0; sh < nshift; sh++)
    {
        const double shift = shifts.mu[sh];
        // saved T with tiny sub-diagonals deflated (UpperHessenbergQR.h:526-539)
        const double Td = d;
        double Te = e;
        {
            const double dn = lane_below(d);
            if (lane < n1 && fabs(e) <= small::kEps * (fabs(d) + fabs(dn)))
                Te = 0.0;
        }
        // Givens sweep on T - shift*I (:541-590); lane i keeps rotation i
        double rc = 0.0, rs = 0.0;
        double r_diag = rdlane(Td, 0) - shift;
        double r_supd = (n1 > 0) ? rdlane(Te, 0) : 0.0;
        double qa = (lane < m) ? Q[lane] : 0.0;
        for (int i = 0; i < n1; i++)
        {
            const double te_i = rdlane(Te, i);
            const double td_i1 = rdlane(Td, i + 1) - shift;
            const double te_i1 = (i < n2) ? rdlane(Te, i + 1) : 0.0;
            double r, c, s;
            givens_fast(r_diag, te_i, r, c, s);
            wrlane(rc, lane, i, c);
            wrlane(rs, lane, i, s);
            r_diag = s * r_supd + c * td_i1;
            if (i < n2)
                r_supd = c * te_i1;
            qa = rotate_cols(Q, m, lane, i, c, s, qa);  // apply_YQ (:403-416)
        }
        if (lane < m)
            Q[n1 * m + lane] = qa;
        // Q'TQ from the saved T (:627-693)
        d = Td;
        e = Te;
        {
            double x = rdlane(Td, 0), y = (n1 > 0) ? rdlane(Te, 0) : 0.0;
            for (int i = 0; i < n1; i++)
            {
                const double c = rdlane(rc, i), s = rdlane(rs, i);
                const double z = rdlane(Td, i + 1);
                const double cs = c * s, c2 = c * c, s2 = s * s;
                const double c2x = c2 * x, s2x = s2 * x, c2z = c2 * z, s2z = s2 * z;
                const double csy2 = 2.0 * c * s * y;
                const double nx = c2x - csy2 + s2z;
                double ny = cs * (x - z) + (c2 - s2) * y;
                const double nz = s2x + csy2 + c2z;
                double nw = 0.0;
                if (i < n2)
                {
                    const double ci1 = rdlane(rc, i + 1), si1 = rdlane(rs, i + 1);
                    const double te_i1 = rdlane(Te, i + 1);
                    const double o = -s * te_i1;
                    nw = te_i1 * c;
                    ny = ci1 * ny - si1 * o;
                }
                wrlane(d, lane, i, nx);
                wrlane(e, lane, i, ny);
                x = nz;
                y = nw;
            }
            wrlane(d, lane, n1, x);
        }
        {
            const double dn = lane_below(d);
            if (lane < n1 && fabs(e) <= small::kEps * (fabs(d) + fabs(dn)))
                e = 0.0;
        }
    }
    __syncthreads();
    if (lane < m)
        diag_io[lane] = d;
    if (lane < m - 1)
        subd_io[lane] = e;
    for (int idx = lane; idx < m * m; idx += 64)
        Qout[idx] = Q[idx];
}


// ===================================================================================================
// The restart's shifted QR sweeps as a skewed pipeline (internal/SmallDensePipelined.h): lane s of
// wave 0 owns sweep s and runs sweep_tick once per tick (its rotation index trails sweep s - 1 by three), the pair it emits
// goes to lane s + 1 by __shfl_up; waves 1-3 rotate the columns of Q (LDS, m x m) with the rotations of the tick before —
// disjoint column pairs, one barrier per tick.  m + 2 + 3 (p - 1) ticks instead of p (m - 1) serial rotations, no fused
// multiply-add, glibc's hypot restated: T and Q are bit-identical to the host routine (tests/test_gpu_small.py).
// LDS: rot_c / rot_s [p][m], d0 / e0 / dout / eout [m], Q [m][m].
// ===================================================================================================
__global__ __launch_bounds__(256) void k_restart_pipelined(int m, double* __restrict__ diag_io, double* __restrict__ subd_io,
                                                           ShiftList shifts, int nshift, double* __restrict__ Qout)
{
    MISPEC_NO_CONTRACT
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* rot_c = sm;
    double* rot_s = rot_c + size_t(nshift) * m;
    double* d0 = rot_s + size_t(nshift) * m;
    double* e0 = d0 + m;
    double* dout = e0 + m;
    double* eout = dout + m;
    double* Q = eout + m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = m, n1 = m - 1;
    for (int i = tid; i < m; i += 256)
    {
        const double d = diag_io[i];
        d0[i] = d;
        double e = 0.0;
        if (i < n1)
        {
            e = subd_io[i];
            if (fabs(e) <= small::kEps * (fabs(d) + fabs(diag_io[i + 1])))  // UpperHessenbergQR.h:526-539
                e = 0.0;
        }
        e0[i] = e;
    }
    for (int idx = tid; idx < m * m; idx += 256)
        Q[idx] = (idx / m == idx % m) ? 1.0 : 0.0;
    __syncthreads();
    const int last = nshift - 1;
    const int ticks = (n + 2) + 3 * last;
    small::SweepLane L = {};
    if (wave == 0 && lane < nshift)
        L.mu = shifts.mu[lane];
    for (int t = 0; t <= ticks; t++)
    {
        if (wave == 0)
        {
            if (t < ticks)
            {
                const int j = t - 3 * lane - 1;
                double din = __shfl_up(L.out_d, 1, 64);
                double ein = __shfl_up(L.out_e, 1, 64);
                if (lane == 0)
                {
                    din = (j + 1 <= n1) ? d0[j + 1] : 0.0;
                    ein = (j >= 0 && j <= n - 2) ? e0[j] : 0.0;
                }
                if (lane < nshift && j >= -1 && j <= n)
                {
                    if (j == n)
                        small::sweep_tail(L);
                    else
                        small::sweep_tick(L, n, j, din, ein, rot_c + size_t(lane) * m, rot_s + size_t(lane) * m);
                    if (lane == last)
                    {
                        if (j >= 1)
                            dout[j - 1] = L.out_d;
                        if (j >= 2)
                            eout[j - 2] = L.out_e;
                    }
                }
            }
        }
        else if (t >= 1)
        {
            // the rotations of tick t - 1: sweep s at j = t - 2 - 3 s, 0 <= j <= n - 2
            const int tp = t - 2;
            int s_hi = tp / 3;  // j >= 0
            if (tp < 0)
                s_hi = -1;
            if (s_hi > last)
                s_hi = last;
            int s_lo = (tp - (n - 2) + 2) / 3;  // j <= n - 2  <=>  s >= (tp - n + 2) / 3
            if (tp - (n - 2) <= 0)
                s_lo = 0;
            const int nact = s_hi - s_lo + 1;
            for (int item = tid - 64; item < nact * m; item += 192)
            {
                const int s = s_lo + item / m, r = item % m;
                const int j = tp - 3 * s;
                const double c = rot_c[size_t(s) * m + j], sn = rot_s[size_t(s) * m + j];
                double* Yi = Q + size_t(j) * m;
                const double qa = Yi[r], qb = Yi[m + r];
                Yi[r] = c * qa - sn * qb;  // apply_YQ (:403-416)
                Yi[m + r] = sn * qa + c * qb;
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < m; i += 256)
    {
        diag_io[i] = dout[i];
        if (i < n1)
            subd_io[i] = eout[i];
    }
    for (int idx = tid; idx < m * m; idx += 256)
        Qout[idx] = Q[idx];
}

}  // namespace

namespace mispec {

void launch_tridiag_eigen(const mispec_ctx& ctx, int n, const double* diag, const double* subd, double* evals, double* evecs,
                          int* info)
{
    MISPEC_REQUIRE(n >= 1 && n <= kMaxSmallDim, "tridiag_eigen kernel: dimension out of range");
    if (n <= 64)  // one wavefront, register / LDS resident; larger matrices: the LDS-resident generic kernel
    {
        hipLaunchKernelGGL(k_tridiag_eigen_w64, dim3(1), dim3(64), size_t(n) * n * sizeof(double), ctx.stream, n, diag, subd, evals,
                           evecs, info);
        MISPEC_HIP(hipGetLastError());
        return;
    }
    const size_t lds = (size_t(2) * n + size_t(n) * n) * sizeof(double);
    MISPEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tridiag_eigen), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   int(lds)));
    hipLaunchKernelGGL(k_tridiag_eigen, dim3(1), dim3(64), lds, ctx.stream, n, diag, subd, evals, evecs, info);
    MISPEC_HIP(hipGetLastError());
}

void launch_restart_sym(const mispec_ctx& ctx, int m, double* diag, double* subd, const double* shifts_host, int nshift,
                        double* Q, bool allow_pipelined)
{
    MISPEC_REQUIRE(m >= 2 && m <= kMaxSmallDim, "restart kernel: dimension out of range");
    MISPEC_REQUIRE(nshift >= 0 && nshift <= kMaxShifts, "restart kernel: too many shifts");
    // the skewed pipeline over four wavefronts where it applies: 110 against 380 us at m = 40, 18 shifts, and the host's bits
    // (profiles/r11g_restart_sweeps_latency.jsonl)
    if (allow_pipelined && m <= kMaxPipelinedDim && nshift >= 1 && nshift <= 64 && nshift < m)
    {
        launch_restart_pipelined(ctx, m, diag, subd, shifts_host, nshift, Q);
        return;
    }
    ShiftList sl;
    for (int i = 0; i < nshift; i++)
        sl.mu[i] = shifts_host[i];
    if (m <= 64)
    {
        hipLaunchKernelGGL(k_restart_sym_w64, dim3(1), dim3(64), size_t(m) * m * sizeof(double), ctx.stream, m, diag, subd, sl,
                           nshift, Q);
        MISPEC_HIP(hipGetLastError());
        return;
    }
    const size_t lds = (size_t(6) * m + size_t(m) * m) * sizeof(double);
    MISPEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_restart_sym), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   int(lds)));
    hipLaunchKernelGGL(k_restart_sym, dim3(1), dim3(64), lds, ctx.stream, m, diag, subd, sl, nshift, Q);
    MISPEC_HIP(hipGetLastError());
}

void launch_restart_pipelined(const mispec_ctx& ctx, int m, double* diag, double* subd, const double* shifts_host, int nshift,
                              double* Q)
{
    MISPEC_REQUIRE(m >= 2 && m <= kMaxPipelinedDim, "pipelined restart kernel: dimension out of range");
    MISPEC_REQUIRE(nshift >= 1 && nshift <= 64 && nshift < m, "pipelined restart kernel: 1 <= shifts <= 64");
    ShiftList sl;
    for (int i = 0; i < nshift; i++)
        sl.mu[i] = shifts_host[i];
    const size_t lds = (size_t(2) * nshift * m + size_t(4) * m + size_t(m) * m) * sizeof(double);
    if (lds > 65536)
        MISPEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_restart_pipelined), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       int(lds)));
    hipLaunchKernelGGL(k_restart_pipelined, dim3(1), dim3(256), lds, ctx.stream, m, diag, subd, sl, nshift, Q);
    MISPEC_HIP(hipGetLastError());
}

void launch_restart_gen(const mispec_ctx& ctx, int m, double* H, const GenShiftList& shifts, double* Q)
{
    MISPEC_REQUIRE(m >= 3 && m <= kMaxGenDim, "general restart kernel: dimension out of range (3 <= ncv <= 96)");
    MISPEC_REQUIRE(shifts.count >= 0 && shifts.count <= kMaxShifts, "general restart kernel: too many shifts");
    const int ld = m | 1;
    const size_t lds = (size_t(2) * m * ld + size_t(3) * m) * sizeof(double) + (size_t(2) * m + 2) * sizeof(int);
    MISPEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hess_restart), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipLaunchKernelGGL(k_hess_restart, dim3(1), dim3(64), lds, ctx.stream, m, ld, H, shifts, Q);
    MISPEC_HIP(hipGetLastError());
}

}  // namespace mispec

// ---- the general sweeps as stand-alone entry points: on the device (one shift per launch) and — the same source with one
// ---- lane — on the host (mirror test/QR.cpp "QR of upper Hessenberg matrix" / "QR decomposition with double shift")
namespace {
void gen_sweep_device(mispec_ctx* ctx, int n, const double* H_host, int kind, double a, double b, double* Q_host, double* QtHQ_host)
{
    ctx->make_current();
    DevBuf<double> H, Q;
    H.alloc(size_t(n) * n);
    Q.alloc(size_t(n) * n);
    MISPEC_HIP(hipMemcpyAsync(H.p, H_host, size_t(n) * n * 8, hipMemcpyHostToDevice, ctx->stream));
    GenShiftList sl;
    sl.count = 1;
    sl.kind[0] = kind;
    sl.a[0] = a;
    sl.b[0] = b;
    launch_restart_gen(*ctx, n, H.p, sl, Q.p);
    if (Q_host)
        MISPEC_HIP(hipMemcpyAsync(Q_host, Q.p, size_t(n) * n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (QtHQ_host)
        MISPEC_HIP(hipMemcpyAsync(QtHQ_host, H.p, size_t(n) * n * 8, hipMemcpyDeviceToHost, ctx->stream));
    MISPEC_HIP(hipStreamSynchronize(ctx->stream));
}
void gen_sweep_host_lanes(int n, const double* H_in, int kind, double a, double b, double* Q_out, double* QtHQ_out)
{
    std::vector<double> H(H_in, H_in + size_t(n) * n), Q(size_t(n) * n, 0.0), work(size_t(3) * n);
    std::vector<int> iwork(size_t(2) * n + 2);
    for (int i = 0; i < n; i++)
        Q[size_t(i) * n + i] = 1.0;
    if (kind == 0)
        small::hess_shifted_qr_lanes(n, H.data(), n, a, Q.data(), n, n, work.data(), small::Lanes{0, 1});
    else
        small::double_shift_qr_lanes(n, H.data(), n, a, b, Q.data(), n, n, work.data(), iwork.data(), small::Lanes{0, 1});
    if (Q_out)
        std::copy(Q.begin(), Q.end(), Q_out);
    if (QtHQ_out)
        std::copy(H.begin(), H.end(), QtHQ_out);
}
}  // namespace

extern "C" int mispec_hess_qr(mispec_ctx* ctx, int n, const double* H_host, double shift, double* Q_host, double* QtHQ_host)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && H_host && n >= 3, "mispec_hess_qr: bad argument");
        gen_sweep_device(ctx, n, H_host, 0, shift, 0.0, Q_host, QtHQ_host);
    });
}
extern "C" int mispec_double_shift_qr(mispec_ctx* ctx, int n, const double* H_host, double s, double t, double* Q_host, double* QtHQ_host)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && H_host && n >= 3, "mispec_double_shift_qr: bad argument");
        gen_sweep_device(ctx, n, H_host, 1, s, t, Q_host, QtHQ_host);
    });
}
extern "C" int mispec_hess_qr_lanes_host(int n, const double* H, double shift, double* Q, double* QtHQ)
{
    return guarded([&] {
        MISPEC_REQUIRE(H && n >= 2, "mispec_hess_qr_lanes_host: bad argument");
        gen_sweep_host_lanes(n, H, 0, shift, 0.0, Q, QtHQ);
    });
}
extern "C" int mispec_double_shift_qr_lanes_host(int n, const double* H, double s, double t, double* Q, double* QtHQ)
{
    return guarded([&] {
        MISPEC_REQUIRE(H && n >= 3, "mispec_double_shift_qr_lanes_host: bad argument");
        gen_sweep_host_lanes(n, H, 1, s, t, Q, QtHQ);
    });
}

// ---- stand-alone unit-test entry points (mirror test/QR.cpp "QR of real tridiagonal matrix" and
// ---- test/Eigen.cpp "Eigen decomposition of symmetric real tridiagonal matrix") ---------------------
namespace {
struct SmallBufs
{
    DevBuf<double> diag, subd, evals, mat;
    DevBuf<int> info;
};
void split_tridiag(int n, const double* T, std::vector<double>& d, std::vector<double>& e)
{
    d.resize(size_t(n));
    e.assign(size_t(n), 0.0);
    for (int i = 0; i < n; i++)
        d[size_t(i)] = T[size_t(i) * n + i];
    for (int i = 0; i < n - 1; i++)
        e[size_t(i)] = T[size_t(i) * n + i + 1];  // T(i+1, i): column i, row i+1
}
}  // namespace

extern "C" int mispec_tridiag_qr(mispec_ctx* ctx, int n, const double* T_host, double shift, double* Q_host, double* QtHQ_host)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && T_host && n >= 2, "mispec_tridiag_qr: bad argument");
        ctx->make_current();
        std::vector<double> d, e;
        split_tridiag(n, T_host, d, e);
        SmallBufs b;
        b.diag.alloc(size_t(n));
        b.subd.alloc(size_t(n));
        b.mat.alloc(size_t(n) * n);
        MISPEC_HIP(hipMemcpyAsync(b.diag.p, d.data(), size_t(n) * 8, hipMemcpyHostToDevice, ctx->stream));
        MISPEC_HIP(hipMemcpyAsync(b.subd.p, e.data(), size_t(n) * 8, hipMemcpyHostToDevice, ctx->stream));
        launch_restart_sym(*ctx, n, b.diag.p, b.subd.p, &shift, 1, b.mat.p);
        MISPEC_HIP(hipMemcpyAsync(d.data(), b.diag.p, size_t(n) * 8, hipMemcpyDeviceToHost, ctx->stream));
        MISPEC_HIP(hipMemcpyAsync(e.data(), b.subd.p, size_t(n) * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (Q_host)
            MISPEC_HIP(hipMemcpyAsync(Q_host, b.mat.p, size_t(n) * n * 8, hipMemcpyDeviceToHost, ctx->stream));
        MISPEC_HIP(hipStreamSynchronize(ctx->stream));
        if (QtHQ_host)
        {
            for (size_t i = 0; i < size_t(n) * n; i++)
                QtHQ_host[i] = 0.0;
            for (int i = 0; i < n; i++)
                QtHQ_host[size_t(i) * n + i] = d[size_t(i)];
            for (int i = 0; i < n - 1; i++)
            {
                QtHQ_host[size_t(i) * n + i + 1] = e[size_t(i)];
                QtHQ_host[size_t(i + 1) * n + i] = e[size_t(i)];
            }
        }
    });
}

// All shifts of one restart, by variant (unit tests and latency probe; mispec.h): 0 host, the reference's serial order
// (tridiag_shifted_qr per shift), 1 host, skewed pipeline + SIMD rows, 2 device, k_restart_pipelined, 3 device, one wavefront.
extern "C" int mispec_restart_sweeps(mispec_ctx* ctx, int n, const double* diag_host, const double* subd_host, const double* shifts_host,
                                     int nshift, int variant, int reps, double* diag_out, double* subd_out, double* Q_out,
                                     double* us_per_call)
{
    return guarded([&] {
        MISPEC_REQUIRE(diag_host && subd_host && shifts_host && n >= 2 && nshift >= 1 && nshift < n && reps >= 1 && variant >= 0 &&
                           variant <= 3,
                       "mispec_restart_sweeps: bad argument");
        MISPEC_REQUIRE(variant < 2 || ctx, "mispec_restart_sweeps: the device variants need a context");
        std::vector<double> d((size_t) n), e((size_t) n, 0.0), Q(size_t(n) * n);
        double us = 0.0;
        if (variant < 2)
        {
            const int ld = (n + 7) / 8 * 8;
            std::vector<double> Qp(size_t(ld) * n), work(size_t(2 * nshift + 2) * n + size_t(4) * n);
            std::vector<small::SweepLane> lanes((size_t) nshift);
            double best = 1e300;
            for (int rep = 0; rep < reps; rep++)
            {
                std::copy(diag_host, diag_host + n, d.begin());
                std::copy(subd_host, subd_host + n - 1, e.begin());
                e[size_t(n) - 1] = 0.0;
                const auto t0 = std::chrono::steady_clock::now();
                if (variant == 0)
                {
                    std::fill(Q.begin(), Q.end(), 0.0);
                    for (int i = 0; i < n; i++)
                        Q[size_t(i) * n + i] = 1.0;
                    for (int sft = 0; sft < nshift; sft++)
                        small::tridiag_shifted_qr(n, d.data(), e.data(), shifts_host[sft], Q.data(), n, n, work.data(), small::Lanes{0, 1});
                }
                else
                {
                    std::fill(Qp.begin(), Qp.end(), 0.0);
                    for (int i = 0; i < n; i++)
                        Qp[size_t(i) * ld + i] = 1.0;
                    small::restart_rotations_pipelined(n, d.data(), e.data(), shifts_host, nshift, work.data(), lanes.data());
                    small::apply_sweeps_to_Q(Qp.data(), ld, ld, n, work.data(), work.data() + size_t(nshift) * n, nshift);
                    for (int c = 0; c < n; c++)
                        std::copy(Qp.begin() + size_t(c) * ld, Qp.begin() + size_t(c) * ld + n, Q.begin() + size_t(c) * n);
                }
                best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
            }
            us = best;
        }
        else
        {
            ctx->make_current();
            SmallBufs b;
            b.diag.alloc(size_t(n) * size_t(reps));
            b.subd.alloc(size_t(n) * size_t(reps));
            b.mat.alloc(size_t(n) * n);
            std::copy(subd_host, subd_host + n - 1, e.begin());
            for (int rep = 0; rep < reps; rep++)
            {
                MISPEC_HIP(hipMemcpyAsync(b.diag.p + size_t(rep) * n, diag_host, size_t(n) * 8, hipMemcpyHostToDevice, ctx->stream));
                MISPEC_HIP(hipMemcpyAsync(b.subd.p + size_t(rep) * n, e.data(), size_t(n) * 8, hipMemcpyHostToDevice, ctx->stream));
            }
            hipEvent_t e0, e1;
            MISPEC_HIP(hipEventCreate(&e0));
            MISPEC_HIP(hipEventCreate(&e1));
            MISPEC_HIP(hipEventRecord(e0, ctx->stream));
            for (int rep = 0; rep < reps; rep++)
            {
                if (variant == 2)
                    launch_restart_pipelined(*ctx, n, b.diag.p + size_t(rep) * n, b.subd.p + size_t(rep) * n, shifts_host, nshift, b.mat.p);
                else
                    launch_restart_sym(*ctx, n, b.diag.p + size_t(rep) * n, b.subd.p + size_t(rep) * n, shifts_host, nshift, b.mat.p, false);
            }
            MISPEC_HIP(hipEventRecord(e1, ctx->stream));
            MISPEC_HIP(hipMemcpyAsync(d.data(), b.diag.p, size_t(n) * 8, hipMemcpyDeviceToHost, ctx->stream));
            MISPEC_HIP(hipMemcpyAsync(e.data(), b.subd.p, size_t(n) * 8, hipMemcpyDeviceToHost, ctx->stream));
            MISPEC_HIP(hipMemcpyAsync(Q.data(), b.mat.p, size_t(n) * n * 8, hipMemcpyDeviceToHost, ctx->stream));
            MISPEC_HIP(hipStreamSynchronize(ctx->stream));
            float ms = 0.f;
            MISPEC_HIP(hipEventElapsedTime(&ms, e0, e1));
            MISPEC_HIP(hipEventDestroy(e0));
            MISPEC_HIP(hipEventDestroy(e1));
            us = 1e3 * double(ms) / reps;
        }
        if (diag_out)
            std::copy(d.begin(), d.end(), diag_out);
        if (subd_out)
            std::copy(e.begin(), e.begin() + n - 1, subd_out);
        if (Q_out)
            std::copy(Q.begin(), Q.end(), Q_out);
        if (us_per_call)
            *us_per_call = us;
    });
}

extern "C" int mispec_tridiag_eigen(mispec_ctx* ctx, int n, const double* T_host, double* evals_host, double* evecs_host)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && T_host && evals_host && n >= 1, "mispec_tridiag_eigen: bad argument");
        ctx->make_current();
        std::vector<double> d, e;
        split_tridiag(n, T_host, d, e);
        SmallBufs b;
        b.diag.alloc(size_t(n));
        b.subd.alloc(size_t(n));
        b.evals.alloc(size_t(n));
        b.mat.alloc(size_t(n) * n);
        b.info.alloc(1);
        MISPEC_HIP(hipMemcpyAsync(b.diag.p, d.data(), size_t(n) * 8, hipMemcpyHostToDevice, ctx->stream));
        MISPEC_HIP(hipMemcpyAsync(b.subd.p, e.data(), size_t(n) * 8, hipMemcpyHostToDevice, ctx->stream));
        launch_tridiag_eigen(*ctx, n, b.diag.p, b.subd.p, b.evals.p, b.mat.p, b.info.p);
        int info = 0;
        MISPEC_HIP(hipMemcpyAsync(evals_host, b.evals.p, size_t(n) * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (evecs_host)
            MISPEC_HIP(hipMemcpyAsync(evecs_host, b.mat.p, size_t(n) * n * 8, hipMemcpyDeviceToHost, ctx->stream));
        MISPEC_HIP(hipMemcpyAsync(&info, b.info.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        MISPEC_HIP(hipStreamSynchronize(ctx->stream));
        if (info != 0)
            throw Error(MISPEC_ERUNTIME, "TridiagEigen: eigen decomposition failed");
    });
}
