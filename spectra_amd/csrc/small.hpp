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
void launch_restart_sym(const mispec_ctx& ctx, int m, double* diag, double* subd, const double* shifts_host, int nshift,
                        double* Q);

}  // namespace mispec
