// B operator of a generalized symmetric problem A x = lambda B x in regular-inverse mode (reginv.hip):
// y = B x as a CSR-stream SpMV and y = B^{-1} x by a conjugate-gradient iteration, both on the device.
#pragma once
#include "common.hpp"
#include "csr.hpp"

struct mispec_reginv
{
    mispec_ctx* ctx = nullptr;
    int64_t n = 0;
    mispec_csr* B = nullptr;                   // the mirrored triangle as a full CSR matrix (owned)
    mispec::DevBuf<double> invdiag;            // Jacobi preconditioner: 1 / B(i,i)  (1 where the diagonal is zero)
    mutable mispec::DevBuf<double> r, p, z, t; // CG work vectors
    mutable mispec::DevBuf<double> partials, scal;  // two-slot block partials and their sums
    mutable mispec::PinnedBuf<double> h_scal;
    mutable mispec::DevBuf<double> stage_x, stage_y;
    mutable int64_t last_iterations = 0;
    ~mispec_reginv();
};

namespace mispec {
// x = B^{-1} rhs (device pointers, n doubles each; x must not alias rhs).  Synchronises the stream every
// iteration (the stopping test is evaluated on the host).  Throws if the iteration limit is reached.
void reginv_solve(const mispec_reginv& R, const double* rhs_dev, double* x_dev);
// s = sum x_i y_i and (if mx != nullptr) max |x_i| as one partial record: partials[kSlotBeta2*pstride + b] and
// partials[kSlotMaxAbs*pstride + b] for exactly `nrec` workgroups — the scalar slots of an orthogonalisation record.
void launch_dot_record(const mispec_ctx& ctx, const double* x, const double* y, int64_t n, double* partials, int64_t pstride,
                       int nrec);
}  // namespace mispec
