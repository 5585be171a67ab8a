// Length-n dense kernels of the Lanczos/Arnoldi factorisation.
#pragma once
#include "common.hpp"

namespace mispec {

// Per-workgroup partial record (kPartialLd slots) written by the orthogonalisation kernels — stored
// slot-major so that the summing kernel reads contiguously — and summed by launch_reduce_partials in a fixed order (deterministic: no floating-point atomics anywhere).
constexpr int kMaxCols = 1024;   // basis columns of a factorisation (ncv); beyond kMaxSmallDim (128) the m x m restart work runs on the host
constexpr int kPanelCols = 64;   // basis columns handled by ONE orthogonalisation / V*Q launch; wider bases go in panels
constexpr int kPartialLd = kMaxCols + 8;
constexpr int kSlotBeta2 = kMaxCols;       // sum f^2
constexpr int kSlotMaxAbs = kMaxCols + 1;  // max |f|
constexpr int kSlotBeta = kMaxCols + 2;    // sqrt(sum f^2)            (filled by the finish step)
constexpr int kSlotErr = kMaxCols + 3;     // max_j |(V'f)_j|          (filled by the finish step)

enum OrthMode
{
    ORTH_VTF = 0,          // c = V'f                                   (ArnoldiOp.h:145-148)
    ORTH_RESID_VTF = 1,    // f = w - alpha*v_i ; |f|^2 ; c = V'f       (Lanczos.h:145-152 fused)
    ORTH_CORRECT_VTF = 2,  // dst = src - V c_in ; |dst|^2 ; c = V'dst  (Lanczos.h:171-179 fused)
    ORTH_CORRECT_ONLY = 3, // dst = src - V c_in ; |dst|^2              (Arnoldi.h:254-255)
    // One-sweep variant (opt-in, fac.hip lanczos_step_lagged; NOT the reference's control flow):
    //   v_i = (f - V c_in)/beta -> vout ;  chk = V' v_i ;  dst = w - alpha v_i ;  c = [V, v_i]' dst ; |dst|^2
    // i.e. the correction of step i-1 and the projection of step i in ONE pass over V[:, :ncol], ncol = i <= 63.
    // Record slots: [0, i) c ; i <v_i, dst> ; [i+1, 2i+1) chk ; kSlotBeta2 / kSlotMaxAbs of dst.
    ORTH_LAGGED = 4
};

// Device-resident bookkeeping of the Lanczos recurrence, so that a whole factorize_from(k, m) can be
// enqueued without the host reading anything back: every kernel of a step looks at `status` (and the
// correction kernels at `need_corr`) and turns into a no-op once the fast path has to stop.
struct StepState
{
    double beta;   // |f| after the last completed reduction
    double alpha;  // Lanczos: H(i,i) before corrections; Arnoldi: |h| of the current step
    double err;    // max |V'f|
    int count;     // corrections applied in the current step (Lanczos.h:155)
    int need_corr; // the while-condition of Lanczos.h:156 for the next correction
    int status;    // kStepOk or the reason the device path stopped
    int stop_step; // step at which it stopped
    int stop_count;
    int lag_pending;  // one-sweep variant: the residual in f is not yet corrected, its accepted V'f is the latest record
    double diag[kMaxCols];  // H(i,i)
    double subd[kMaxCols];  // H(i+1,i)
    // one-sweep variant, diagnostics of a device run: largest accepted |c|/|f|, largest |V'v_i| after a lagged correction,
    // lagged steps executed
    double lag_rel_c_max;
    double lag_chk_max;
    int64_t lag_steps;
    // fused restart without a host turn (kFinishFusedRestart): max |V'f| and |f| of the corrected residual the restart's V*Q pass
    // measured (Lanczos.h:156), read by the host at the end of the sweep that follows
    double rst_err;
    double rst_beta_corr;
    int64_t onered_steps;  // lagged steps that took the one-reduction form
};
enum
{
    kStepOk = 0,
    kStepSmallBeta = 1,  // beta < sqrt(eps) at the start of a step: the reference's restart heuristics run on the host path
    kStepMoreCorr = 2,   // a third correction is needed: continue the loop on the host path
    kStepTinyF = 3,      // beta < eps*sqrt(n) inside the loop (Lanczos.h:163-168): host zeroes f
    kStepLagCheck = 4,   // one-sweep variant: column stop_step needs a second correction (Lanczos.h:156 after the first one);
                         // the host finishes step stop_step-1 with the reference's loop and repeats step stop_step
    kStepRestartCheck = 5  // fused restart: the corrected residual fails the reference's test (Lanczos.h:156): no step of the sweep
                           // that was enqueued behind the restart runs, the host continues the reference's loop and repeats it
};
enum
{
    kFinishNone = 0,
    kFinishNorms = 1,      // beta = sqrt(sum f^2), err = max|c|
    kFinishStepFirst = 2,  // + bookkeeping after f = w - alpha v (Lanczos.h:142-153)
    kFinishStepCorr = 3,   // + bookkeeping after one correction (Lanczos.h:171-180)
    kFinishArnoldiH = 4,   // Arnoldi: red[0..ncol) is h = V'w -> H(:, step), |h| (Arnoldi.h:251)
    kFinishArnoldiF = 5,   // Arnoldi: after f = w - Vh: beta, the 0.717 test and the need for corrections (Arnoldi.h:255-266)
    kFinishLagged = 6,     // one-sweep variant: bookkeeping after an ORTH_LAGGED pass
    kFinishFusedRestart = 7  // record of k_vq_fused (slots [0, m) V'f_corr, m |f_new|^2, kSlotBeta2 |f_corr|^2; ncol = m + 1): the start
                             // state of the next sweep — beta = |f_new| — and the reference's test of the corrected residual
};
struct FinishArgs
{
    int mode = kFinishNorms;
    StepState* st = nullptr;
    int step = 0;
    const double* alpha_src = nullptr;  // first: device scalar <v, w>
    const double* prev_red = nullptr;   // corr: the coefficients the correction used (H(i-1,i) += c[i-1], H(i,i) += c[i])
    double eps = 0.0;
    double beta_thresh = 0.0;
    int max_spec = 2;  // corrections that are enqueued speculatively per step
    double* hcol = nullptr;  // Arnoldi: device column `step` of H
    int lag_last = 0;        // kFinishLagged: last step of the sweep — the following CORRECT_VTF launches finish f the reference's way
    double lag_limit = 1e-6; // kFinishLagged: a correction is lagged only while |c|^2 <= lag_limit |f|^2
    double eps_sqrt = 0.0;   // kFinishLagged: ... and the corrected norm stays >= sqrt(eps) (Lanczos.h:107 needs a finished f)
    // Sharded runs: the sums travel as ONE contiguous message [0, ncol] — the reduction also stores sum f^2 in slot `ncol` of the
    // staging record, the finish step takes it from there (a record is (2 ncv) doubles on the wire, not kSlotBeta2 + 1 = 1025)
    int packed = 0;
    // One reduction per step (kFinishLagged, fac.hip lanczos_step_lagged with F.onered; CPU restatement: oracle/onesweep_variant.hpp,
    // flavour one-reduction): the operator was applied to the UN-normalised residual of this record's pass, u = A f~, and the
    // partial sums of <f~, u> are reduced by the same kernel (sharded: travel in the same message, behind sum f^2); after the
    // bookkeeping of the step the tail forms alpha~ = <f~, u> / beta^2 - <f~, v_i> for the step that follows (-> alpha_out) and
    // takes that step's beta < sqrt(eps) stop (Lanczos.h:107, what k_scale_step / the post-scaled SpMV do otherwise).
    const double* alpha_parts = nullptr;
    int64_t alpha_count = 0;
    double* alpha_out = nullptr;
};

struct OrthArgs
{
    const double* V = nullptr;  // basis, column-major
    int64_t ldv = 0;
    int ncol = 0;               // columns 0..ncol-1 take part (launch_orth splits more than kPanelCols into panels)
    int col0 = 0;               // set by launch_orth: first basis column of this panel (V column, c_in entry, record slot)
    int norms = 1;              // set by launch_orth: this panel writes the |f|^2 / max|f| slots
    int64_t n = 0;              // local rows
    const double* src = nullptr;  // VTF: f ; RESID: w ; CORRECT: input vector
    double* dst = nullptr;        // RESID / CORRECT output (may alias src)
    const double* vi = nullptr;   // RESID: basis column i
    const double* alpha_dev = nullptr;  // RESID: device scalar
    const double* c_in = nullptr;       // CORRECT: device coefficients [ncol]
    double* partials = nullptr;         // slot-major: partials[slot * pstride + workgroup]
    int64_t pstride = 0;                // >= number of workgroups of any launch
    const int* status = nullptr;        // optional predicate: run only while *status == kStepOk ...
    const int* need_corr = nullptr;     // ... and (if given) *need_corr != 0
    // ORTH_LAGGED: src = w, vi = f (the uncorrected residual of the previous step), dst = f (next residual), vout = column i,
    // alpha_dev = <col_i, w>, beta_dev = the divisor column i was formed with, pending = use c_in (else c_in is taken as 0)
    double* vout = nullptr;
    const double* beta_dev = nullptr;
    const int* pending = nullptr;
    // one reduction per step: src = u = A f~ (un-normalised); the pass forms w = u / beta - beta V[:, ncol - 1] on the fly
    int onered = 0;
};

// ORTH_LAGGED with the basis streamed through an LDS ring by LDS-DMA (orth_dma.hip): eligible for one column panel on vectors of
// at least 512 tiles of 128 rows; returns the number of partial records (= workgroups: one per CU).  depth_override 2: two ring slots.
bool orth_lagged_dma_eligible(const OrthArgs& a);
int launch_orth_lagged_dma(const mispec_ctx& ctx, const OrthArgs& a, int depth_override, int flags = 0);

// The other modes (reference flow, Arnoldi) the same way (orth_dma_modes.hip): one column panel, vectors of at least 1024 tiles.
bool orth_dma_modes_eligible(OrthMode mode, const OrthArgs& a);
int launch_orth_dma_mode(const mispec_ctx& ctx, OrthMode mode, const OrthArgs& a);

// All launchers enqueue on ctx.stream and return immediately.
int launch_orth(const mispec_ctx& ctx, OrthMode mode, const OrthArgs& a);  // returns the number of partial records
// red[0..kPartialLd): column sums of the records (+ max for kSlotMaxAbs); finish additionally fills kSlotBeta / kSlotErr.
void launch_reduce_partials(const mispec_ctx& ctx, const double* partials, int64_t pstride, int nrec, int ncol, double* red,
                            const FinishArgs& fin);
// While the device-driven run `fin.st` is live both kernels write `red`; once it has stopped they leave it alone (the host
// continues from the records of the stopping pass).  launch_finish: red <- stage (an all-reduced record), then the scalar tail.
void launch_finish(const mispec_ctx& ctx, const double* stage, double* red, int ncol, const FinishArgs& fin);
// Host turn of a restart without DMA-engine copies (option host_turn): the state of a finished sweep written into pinned host
// memory by a kernel + a sequence flag the host spins on; Q and the next sweep's start state fetched from pinned host memory.
void launch_publish_state(const mispec_ctx& ctx, const StepState* st, int m, StepState* host_dst, unsigned long long* host_flag,
                          unsigned long long seq);
void launch_fetch_restart(const mispec_ctx& ctx, const double* host_src, int m, double* Qdev, StepState* st, int with_state);
// out[0] = sum of `count` doubles (SpMV alpha partials), fixed order
void launch_reduce_sum(const mispec_ctx& ctx, const double* in, int64_t count, double* out);
// dst = src / divisor over npad elements (v = f / beta, Lanczos.h:106)
void launch_scale(const mispec_ctx& ctx, const double* src, double* dst, int64_t npad, double divisor);
// Step start on the device path: if st->beta < eps_sqrt stop (kStepSmallBeta); else dst = src / st->beta and
// st->subd[step-1] = st->beta (Lanczos.h:99-128 without the restart branch)
void launch_scale_step(const mispec_ctx& ctx, const double* src, double* dst, int64_t npad, StepState* st, int step,
                       double eps_sqrt);
// f = f*a + v*b, partial |f|^2 records (Arnoldi.h:337-339); returns the number of records
int launch_axpby(const mispec_ctx& ctx, double* f, double a, const double* v, double b, int64_t n, double* partials,
                 int64_t pstride);
// X[:, 0:p] = V[:, 0:m] * Q (Q device, m x p col-major, ldq); X may alias V (in-place compress_V).
void launch_vq(const mispec_ctx& ctx, const double* V, int64_t ldv, int m, const double* Q, int ldq, int p, double* X,
               int64_t ldx, int64_t n);
// One-sweep steps, end of a sweep: the restart's V <- V Q with the PENDING correction of the last Lanczos step riding on it
// (fac.hip restart_sym).  X[:, 0:p] = V[:, 0:m] Q (X == V allowed: in place), and from the same tile of V:
//   f_corr = ftilde - V c ;  chk = V' f_corr  (the reference's test for a second correction, Lanczos.h:156) ;  |f_corr|^2
//   fnew = f_corr * q_last + X[:, kcol] * h_sub  (Arnoldi.h:337) ;  |fnew|^2
// Records: slots [0, m) chk, slot m |fnew|^2, kSlotBeta2 |f_corr|^2.  m, p <= kPanelCols.  Returns the number of records.
struct VqFusedArgs
{
    const double* c = nullptr;       // device: the accepted V'f of the last step (m entries)
    const double* ftilde = nullptr;  // the uncorrected residual
    double* fnew = nullptr;          // out (must not alias ftilde)
    double q_last = 0.0, h_sub = 0.0;
    int kcol = 0;                    // output column that enters fnew (= the new subspace dimension k)
    double* partials = nullptr;
    int64_t pstride = 0;
};
int launch_vq_fused(const mispec_ctx& ctx, const double* V, int64_t ldv, int m, const double* Q, int ldq, int p, double* X, int64_t ldx,
                    int64_t n, const VqFusedArgs& fa);
// res[j] = || A x_j - lambda_j x_j ||^2 partials and ||x_j||^2 partials are produced by the caller with the kernels above.
// r = y - lambda*x ; records hold |r|^2 in kSlotBeta2 and |x|^2 in slot 0.
int launch_resid_norms(const mispec_ctx& ctx, const double* y, const double* x, double lambda, int64_t n, double* partials,
                       int64_t pstride);
// complex pair (x_r + i x_i, lambda = a + i b): records hold |A x - lambda x|^2 in kSlotBeta2 and |x|^2 in slot 0,
// given y_r = A x_r, y_i = A x_i
int launch_resid_norms_complex(const mispec_ctx& ctx, const double* yr, const double* yi, const double* xr, const double* xi,
                               double a, double b, int64_t n, double* partials, int64_t pstride);
// Un-fused Lanczos epilogue for user operators (Lanczos.h:139-142): w -= h_prev*v_prev (if v_prev), one partial
// of <v, w> per workgroup in partials[0 .. lanczos_epilogue_records).
int lanczos_epilogue_records(const mispec_ctx& ctx, int64_t n);
// h_prev_dev (device-driven steps): H(i,i-1) is read from device memory instead; status: the launch is a no-op unless *status == 0
void launch_lanczos_epilogue(const mispec_ctx& ctx, double* w, const double* v, const double* v_prev, double h_prev, int64_t n,
                             double* partials, const double* h_prev_dev = nullptr, const int* status = nullptr);
// fill v[i] = SimpleRandom(seed) stream element (row_begin + i), i < nloc, by LCG jump-ahead (Util/SimpleRandom.h:30-123)
void launch_simple_random(const mispec_ctx& ctx, double* v, int64_t row_begin, int64_t nloc, uint64_t seed);

}  // namespace mispec
