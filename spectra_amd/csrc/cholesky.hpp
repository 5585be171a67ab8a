// B = L L' for the Cholesky mode of the generalized solver (cholesky.hip): y = L^{-1} x and y = L^{-T} x on the device.
#pragma once
#include "common.hpp"
#include "csr.hpp"
#include "shiftsolve.hpp"

struct mispec_cholesky
{
    mispec_ctx* ctx = nullptr;
    int64_t n = 0;
    int info = 0;                       // CompInfo: 0 Successful, 3 NumericalIssue (B not positive definite)
    mispec::DevBuf<double> linv, linvt; // L^{-1} and its transpose, dense n x n row-major (n <= 4096)
    mispec_symshift* band = nullptr;    // n > 4096, half-bandwidth <= 8: the partitioned band factorisation of B (shiftsolve.hip)
    ~mispec_cholesky();
    mutable mispec::DevBuf<double> stage_x, stage_y;
};

namespace mispec {
constexpr int64_t kMaxCholesky = 4096;  // dense factor: matrix dimension
// y = L^{-1} x  (upper == false)  or  y = L^{-T} x  (upper == true); device pointers, n doubles
void launch_cholesky_solve(const mispec_cholesky& C, bool upper, const double* x_dev, double* y_dev);
}  // namespace mispec
