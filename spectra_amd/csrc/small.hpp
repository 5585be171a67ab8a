// Launchers of the ncv x ncv restart kernels (small.hip).
#pragma once
#include "common.hpp"

namespace mispec {

constexpr int kMaxSmallDim = 128;  // LDS: (6 m + m^2) * 8 B <= 160 KiB
constexpr int kMaxShifts = 128;
struct ShiftList
{
    double mu[kMaxShifts];
};

// TridiagEigen::compute on (diag, subd) -> evals[n], evecs[n*n] (device pointers); *info = 0 on success.
void launch_tridiag_eigen(const mispec_ctx& ctx, int n, const double* diag, const double* subd, double* evals, double* evecs,
                          int* info);
// nshift x { TridiagQR::compute(T, mu); Q <- Q Qi; T <- Qi' T Qi } ; diag/subd updated in place, Q[m*m] written.
// allow_pipelined: k_restart_pipelined where it applies (m <= 64), else the one-wavefront kernels.
void launch_restart_sym(const mispec_ctx& ctx, int m, double* diag, double* subd, const double* shifts_host, int nshift,
                        double* Q, bool allow_pipelined = true);

// The same sweeps as a skewed pipeline over 256 threads (internal/SmallDensePipelined.h): bit-identical to the host routine.
constexpr int kMaxPipelinedDim = 64;  // LDS: (2 p m + 4 m + m^2) * 8 B <= 100 KiB at m = 64, p = 63
void launch_restart_pipelined(const mispec_ctx& ctx, int m, double* diag, double* subd, const double* shifts_host, int nshift,
                              double* Q);

// General (Hessenberg) restart on the device (a19): a list of shifts applied to the m x m Hessenberg H (device, in/out,
// leading dimension m), accumulating Q (m x m, written).  kind 0: one real shift `a` (UpperHessenbergQR); kind 1: a
// conjugate pair as the double shift (s, t) = (a, b) (DoubleShiftQR).  One wavefront, H and Q resident in LDS: m <= kMaxGenDim.
constexpr int kMaxGenDim = 96;  // LDS: 2 m (m|1) + 3 m doubles + (2m+2) ints <= 160 KiB
struct GenShiftList
{
    int count;
    int kind[kMaxShifts];
    double a[kMaxShifts], b[kMaxShifts];
};
void launch_restart_gen(const mispec_ctx& ctx, int m, double* H, const GenShiftList& shifts, double* Q);

}  // namespace mispec
