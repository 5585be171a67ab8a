// Shift-and-invert operator y = (A - sigma I)^{-1} x held on the device (shiftsolve.hip).
#pragma once
#include <memory>

#include "common.hpp"

namespace mispec {
struct BandLevel;
constexpr int64_t kNarrowBandwidth = 8;  // up to here the top level is factored on the device and runs the LDS-staged sweeps
constexpr int64_t kMaxBandwidth = 64;    // banded path: half-bandwidth of A - sigma I the chunk kernels take at any level
constexpr int64_t kMaxDense = 4096;      // dense path: matrix dimension
// Which path a symmetric operator of dimension n and half-bandwidth b takes: the partitioned band factorisation for narrow bands
// (any n) and for bands of up to 64 beyond the dense limit; the dense inverse otherwise (n <= kMaxDense), else unsupported.
inline bool band_path(int64_t n, int64_t b) { return b <= kNarrowBandwidth || (b <= kMaxBandwidth && n > kMaxDense); }
}  // namespace mispec

struct mispec_symshift
{
    mispec_ctx* ctx = nullptr;
    int64_t n = 0;
    // the selected triangle of A as (row >= col) triplets, host memory (the factorisation is redone per shift)
    std::vector<int64_t> rows, cols;
    std::vector<double> vals;
    int64_t half_bandwidth = 0;
    // banded path: the unshifted band A(i, i-d), n x (band_b + 1) row-major, built once at construction — on the
    // host (separator rows, pivots' scale) and resident in HBM (the device factorisation shifts a copy of it)
    std::vector<double> band0;
    mispec::DevBuf<double> band0_dev;
    int band_b = 0;
    // pencil form (SymShiftInvert, MatOp/SymShiftInvert.h:140-208): the operator is (A - sigma B)^{-1}; B is kept the
    // same way as A (band on host + device, or triplets for the dense path).  Empty: B = I.
    // general (non-symmetric) matrix — SparseGenRealShiftSolve: every stored entry is used as it is; dense path only
    bool general = false;
    bool pencil = false;
    std::vector<double> bandB0;
    mispec::DevBuf<double> bandB0_dev;
    std::vector<int64_t> rowsB, colsB;
    std::vector<double> valsB;
    double sigma = 0.0;
    bool factored = false;
    bool dense = false;
    std::unique_ptr<mispec::BandLevel> top;  // banded path
    mispec::DevBuf<double> inverse;          // dense path: (A - sigma I)^{-1}, n x n column-major
    mutable mispec::DevBuf<double> stage_x, stage_y;
    // Robustness for indefinite A - sigma I (banded path): what the last factorisation saw and the number of iterative
    // refinement steps every solve performs (calibrated by set_shift on a probe right-hand side; 0 for definite matrices)
    long long boosted_pivots = 0;
    double min_pivot_ratio = 1.0;
    int refine_steps = 0;
    // SparseCholesky beyond the dense limit (cholesky.hip): factor the band itself (sigma = 0) and keep what G^{-1} / G^{-T}
    // need; cholesky_ready only when every pivot of every level was positive (the matrix is positive definite)
    bool want_cholesky = false, cholesky_ready = false;
    long long negative_pivots = 0;
    double probe_backward_error = 0.0;  // of the calibrated solve
    mutable mispec::DevBuf<double> ref_r, ref_dy;
    // Reordered at construction (round 5): a matrix beyond the dense limit whose band is too wide as it comes, but narrow after a
    // reverse Cuthill-McKee ordering (a banded matrix in a scattering row order, a path- or ladder-like graph), is stored as
    // P (A - sigma B) P' with (P x)[i] = x[perm[i]]; the solves keep the caller's index order (gather, solve, scatter).
    std::vector<int32_t> perm_host;     // new -> old; empty: not reordered
    mispec::DevBuf<int32_t> perm_dev;
    mutable mispec::DevBuf<double> perm_x, perm_y;
    int64_t half_bandwidth_as_given = 0;
    bool reordered() const { return perm_dev.p != nullptr; }
    ~mispec_symshift();
};

namespace mispec {
// y = (A - sigma I)^{-1} x, device pointers of n doubles, enqueued on the context stream
void launch_shiftsolve(const mispec_symshift& S, const double* x_dev, double* y_dev);
// y = G^{-1} x (upper == false) / y = G^{-T} x (upper == true) for the factor G G' = A of a positive definite banded A that
// was factored with want_cholesky and sigma = 0
void launch_band_cholesky_solve(const mispec_symshift& S, bool upper, const double* x_dev, double* y_dev);
}  // namespace mispec
