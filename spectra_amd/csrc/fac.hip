// Device-resident Lanczos / Arnoldi factorisation  A V = V H + f e'  and the restart primitives.
//
// Replaces (yixuan/spectra v1.2.0, include/Spectra/): LinAlg/Arnoldi.h (init :136-195, expand_basis
// :66-115, factorize_from :198-295, compress_V :320-340), LinAlg/Lanczos.h (factorize_from :62-187)
// and the ArnoldiOp reductions (MatOp/internal/ArnoldiOp.h:137-161).  Control flow, thresholds and
// the order of the updates are the reference's; what differs is where the data lives and how many
// passes over V each step makes:
//   * V (local_rows x ncv, column-major), f and w stay in HBM for the whole solve; per step the host
//     sees ~70 doubles (alpha, beta, V'f) through one pinned D2H copy per decision;
//   * SpMV, "w -= beta v_prev" and the alpha dot product are one kernel (csr.hip);
//   * "f = w - alpha v", |f| and V'f are one pass over V; each re-orthogonalisation
//     (f -= V c, |f|, V'f) is ONE pass over V instead of the reference's two (krylov.hip);
//   * H is kept on the host (ncv x ncv, authoritative) — every rank of a row-sharded run holds the
//     same H because all reductions end in an all-reduce.
#include "csr.hpp"
#include "dense.hpp"
#include "krylov.hpp"
#include <chrono>
#include "cholesky.hpp"
#include "reginv.hpp"
#include "shiftsolve.hpp"
#include "small.hpp"

#include <sys/mman.h>
#include <Spectra/internal/SmallDense.h>
#include <Spectra/internal/SmallDensePipelined.h>

#include <cmath>
#include <memory>
#include <cstdlib>
#include <cstring>
#include <limits>

using namespace mispec;

namespace {
constexpr double kEps = 2.220446049250313e-16;           // TypeTraits<double>::epsilon()
constexpr double kNear0 = 2.2250738585072014e-308 * 10;  // TypeTraits<double>::min() * 10 (Arnoldi.h:50)

enum Family
{
    FAM_SPMV = 0,
    FAM_VTF,
    FAM_GEMV,
    FAM_SCALE,
    FAM_COMPRESS,
    FAM_SMALL,
    FAM_REDUCE,  // the record reduction behind a one-sweep pass, timed a second time on its own (level 1; also inside FAM_VTF)
    // profile level 3 ("wire", sharded runs): the exchange on its stream, the part of it the product waits for, the all-reduces
    FAM_EXCH,
    FAM_XWAIT,
    FAM_ALLRED,
    FAM_COUNT
};
}  // namespace

struct mispec_fac
{
    mispec_ctx* ctx = nullptr;
    const mispec_csr* A = nullptr;
    const mispec_csr* A2 = nullptr;      // product operator y = A2 (A x) (contrib/PartialSVDSolver.h: A'A or AA'); else nullptr
    const mispec_symshift* S = nullptr;  // operator = (A - sigma I)^{-1} on the device
    // Generalized problem in regular-inverse mode (SymGEigsSolver.h:224-238): operator y = B^{-1}(A x) and every
    // inner product taken as x'By (ArnoldiOp.h:68-101).  bx holds B*(the vector the product is taken with).
    const mispec_reginv* Bop = nullptr;
    // Cholesky mode (SymGEigsSolver.h:142-208): operator y = L^{-1} A L^{-T} x with B = L L', plain inner products
    const mispec_cholesky* Chol = nullptr;
    // ... or in one of the shift modes of SymGEigsShiftSolver.h (operator (A - sigma B)^{-1} M x through F.S, with
    // M = B for shift-invert / buckling and M = A + sigma B for Cayley): Bcsr is the matrix of the inner product;
    // the Cayley operator is evaluated as x + 2 sigma (A - sigma B)^{-1} B x (SymGEigsCayleyOp.h:88-99).
    const mispec_csr* Bcsr = nullptr;
    bool cayley = false;
    double cay_sigma = 0.0;
    bool bmode() const { return Bop != nullptr || Bcsr != nullptr; }
    mispec_op_fn op = nullptr;
    void* op_user = nullptr;
    // dense operator in HBM (DenseSymMatProd / DenseGenMatProd), or a user operator that works on DEVICE pointers
    const mispec_dense* D = nullptr;
    mispec_device_op_fn dop = nullptr;
    void* dop_user = nullptr;
    int64_t n = 0;     // global dimension
    int64_t nloc = 0;  // rows of this shard
    int64_t row_begin = 0;
    int64_t block = 0;  // all-gather block (rows per rank, padded)
    int64_t ldv = 0;
    int m = 0;
    bool symmetric = true;
    int k = 0;
    double beta = 0.0;
    std::vector<double> H;  // m x m column-major, host

    DevBuf<double> mid;  // product operator: A x
    DevBuf<double> bx;
    DevBuf<double> V, f, w, tmp, xfull, X, partials, alpha_partials, red, Qdev, d_diag, d_subd, d_evals, d_evecs, d_Y, gmax;
    DevBuf<int> d_info;
    DevBuf<StepState> d_state;     // device-driven step bookkeeping (krylov.hpp)
    DevBuf<double> d_H;            // device-driven Arnoldi: the columns of H written by the steps (m x m)
    PinnedBuf<double> h_H;
    PinnedBuf<StepState> h_state;  // its pinned host mirror
    // Host turn without DMA-engine copies (option host_turn, fetch_state / restart_sym): the sequence word the publishing kernel
    // writes behind the state, the upload staging of a restart [Q m*m][diag m][subd m], and what the turns cost on the host
    PinnedBuf<unsigned long long> h_flag;
    unsigned long long pub_seq = 0;
    PinnedBuf<double> h_up;
    int64_t turn_count = 0, turn_fallbacks = 0;
    double turn_host_s = 0.0;      // host time between "state seen" and "restart enqueued", summed over the restarts
    std::chrono::steady_clock::time_point turn_t0;
    // scratch of the restart's pipelined QR sweeps (restart_sym), kept between restarts
    std::vector<double> sweep_work, sweep_q;
    std::vector<small::SweepLane> sweep_lanes;
    bool turn_open = false;
    PinnedBuf<double> h_red, h_small, h_x, h_y;
    PinnedBuf<double> h_stage[2];  // pinned staging of download_columns (allocated on first use)
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    bool device_steps = true;      // MISPEC_HOST_STEPS=1 forces the host-synchronous path
    // One-sweep variant of the Lanczos steps: the correction of a step rides on the next step's pass over V.
    // On by default since round 4 (set at creation from MISPEC_ORTH, default "onesweep"); mispec_fac_set_orth_mode /
    // MISPEC_ORTH=reference select the reference's two-pass control flow.
    bool onesweep = true;
    double lag_limit = 1e-6;
    int64_t lag_steps = 0, lag_check_stops = 0, lag_state_stops = 0;
    double lag_rel_c_max = 0.0, lag_chk_max = 0.0;
    int64_t pstride = 0;  // stride between slots of the partial records
    // Neighbour exchange plan (sharded device matrices): per peer, which part of my slice it reads and which
    // part of its slice I read.  halo == false: the full all-gather is used.
    bool halo = false;
    std::vector<int64_t> send_off, send_count, recv_off, recv_count;
    int64_t halo_recv = 0;  // doubles received per exchange
    // Overlap of the exchange with the product: the longest run of 256-row blocks that reference only this
    // rank's own slice of x is multiplied on the solver's stream while the exchange runs on a second stream; the remaining
    // blocks follow when it has landed.  interior_count == 0: no overlap (everything after the exchange).
    int interior_first = 0, interior_count = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_x_ready = nullptr, ev_x_landed = nullptr;
    // The matrix is stored reordered (P A P', reorder.hip) and this factorisation works in that order: start vectors are
    // permuted on the way in, V / f / Ritz vectors on the way out; plain operators only (product, generalized and
    // Cholesky operators use the order-preserving product instead)
    int post_scale_step = 0;  // > 0 while apply_op is to let the SpMV take the un-normalised residual of that one-sweep step
    bool perm_mode = false;
    bool x_original = false;  // the columns of X have been put back into the caller's order
    DevBuf<double> pscratch;
    int red_cur = 0;   // which half of `red` holds the latest reduced record
    // One-sweep steps: the correction of the LAST step of a full sweep is left pending (c = red_buf(end_rec)[0, m), H and beta
    // already carry it) so that the restart's V*Q pass can apply it on the way (mispec_fac_restart_sym); everything else that
    // needs f calls finish_pending first.
    bool end_pending = false;
    bool eager_last = false, test_recorrect = false;  // MISPEC_ORTH_EAGER_LAST / MISPEC_ORTH_TEST_RECORRECT
    bool test_restart_check = false;                   // MISPEC_ORTH_TEST_RESTART_CHECK
    // One reduction per lagged step (MISPEC_ORTH_ONE_REDUCTION): the record of a lagged pass is not reduced at
    // once — the next step's product runs on the un-normalised residual and ONE kernel (sharded: one all-reduce) reduces that
    // record together with the product's <f~, A f~>.  `lag_def` is the record waiting for that.
    bool onered = false, skip_alpha_reduce = false;
    int64_t onered_steps = 0;
    struct DeferredRecord
    {
        bool have = false;
        int nrec = 0, ncol = 0, half = 0;
        FinishArgs fin;
    } lag_def;
    // set when a fused restart's test (Lanczos.h:156 on the corrected residual) failed or came within a factor of two of its bar:
    // the remaining sweeps of this solve apply their last correction before the restart, the reference's order (cleared by init)
    bool eager_sticky = false;
    int end_rec = 0;
    // Fused restart without a host turn: the restart left the start state of the next sweep (beta = |f_new|, the reference's test
    // of the corrected residual) in d_state and enqueued nothing that needs the host; F.beta is stale until resolve_restart /
    // the end of the next device-driven sweep.  MISPEC_RESTART_SYNC=1 restores the synchronising restart.
    bool restart_unresolved = false;
    int64_t fused_restarts = 0, fused_recorrected = 0;
    int x_cols = 0;    // columns currently held in X

    // profile
    int prof = 0;  // 0 off, 1 every kernel family, 2 only the operator applications
    int64_t counts[FAM_COUNT] = {};
    int64_t n_sync = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[FAM_COUNT];
    std::vector<hipEvent_t> ev_pool;
    double ms_acc[FAM_COUNT] = {};
    double bytes_acc[FAM_COUNT] = {};  // algorithmic bytes of the n-sized dense kernels (mispec_profile)
    void count_bytes(int fam, int vectors) { bytes_acc[fam] += 8.0 * double(nloc) * double(vectors); }

    double& Hat(int i, int j) { return H[size_t(j) * m + i]; }
    double* col(int j) { return V.p + int64_t(j) * ldv; }
    double* red_buf(int which) { return red.p + which * kPartialLd; }
    // <v, w> of the fused SpMV epilogue: a device scalar of its own behind the two record halves (a reduction rewrites every
    // slot of the half it targets, and the one-sweep steps let records land in either half while alpha is still needed)
    double* alpha_slot() { return red.p + 2 * kPartialLd; }
    // sharded runs: where the local sums of a record are all-reduced before they become a record half (krylov.hpp launch_finish)
    double* red_stage() { return red.p + 2 * kPartialLd + 8; }
    hipStream_t stream() const { return ctx->stream; }
    // a communicator is attached (world may be 1: the collectives are then still issued, which is how the
    // RCCL / torch transports are smoke-tested on a single GPU)
    bool sharded() const { return ctx->comm.allgather != nullptr; }

    ~mispec_fac()
    {
        for (auto& fam : ev)
            for (auto& pr : fam)
            {
                (void) hipEventDestroy(pr.first);
                (void) hipEventDestroy(pr.second);
            }
        for (auto e : ev_pool)
            (void) hipEventDestroy(e);
        for (auto e : ev_stage)
            if (e)
                (void) hipEventDestroy(e);
        if (ev_x_ready)
            (void) hipEventDestroy(ev_x_ready);
        if (ev_x_landed)
            (void) hipEventDestroy(ev_x_landed);
        if (comm_stream)
            (void) hipStreamDestroy(comm_stream);
    }
};

namespace {

hipEvent_t take_event(mispec_fac& F)
{
    if (!F.ev_pool.empty())
    {
        hipEvent_t e = F.ev_pool.back();
        F.ev_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    MISPEC_HIP(hipEventCreate(&e));
    return e;
}

// RAII timing scope: records a HIP-event pair on the context stream around the enclosed launches.
struct Timed
{
    mispec_fac& F;
    int fam;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t on = nullptr;
    Timed(mispec_fac& f, int family, hipStream_t other_stream = nullptr, bool is_pass = true)
        : F(f), fam(family), on(other_stream ? other_stream : f.stream())
    {
        F.counts[fam]++;
        const bool wire = fam == FAM_EXCH || fam == FAM_XWAIT || fam == FAM_ALLRED;
        // level 2: only the operator applications are timed; level 3: those and the collectives; level 1: every kernel family;
        // level 4: only the passes over the basis (FAM_VTF without the record reductions behind them) — an event pair costs a
        // few microseconds and keeps the next launch from being prepared behind the running kernel, so a family is measured
        // best when it is the only one bracketed (the one-sweep pass: 467 us per launch in a rocprofv3 trace, 472 at level 4,
        // 514 at level 1)
        if (!F.prof || (F.prof == 2 && fam != FAM_SPMV) || (F.prof == 3 && fam != FAM_SPMV && !wire) || (F.prof != 3 && wire) ||
            (F.prof == 4 && (fam != FAM_VTF || !is_pass)))
            return;
        e0 = take();
        e1 = take();
        (void) hipEventRecord(e0, on);
    }
    hipEvent_t take() { return take_event(F); }
    ~Timed()
    {
        if (!F.prof || !e0)
            return;
        (void) hipEventRecord(e1, on);
        F.ev[fam].emplace_back(e0, e1);
    }
};

void drain_profile(mispec_fac& F)
{
    MISPEC_HIP(hipStreamSynchronize(F.stream()));
    for (int fam = 0; fam < FAM_COUNT; fam++)
    {
        for (auto& pr : F.ev[fam])
        {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess)
                F.ms_acc[fam] += double(ms);
            F.ev_pool.push_back(pr.first);
            F.ev_pool.push_back(pr.second);
        }
        F.ev[fam].clear();
    }
}

void sync_stream(mispec_fac& F)
{
    MISPEC_HIP(hipStreamSynchronize(F.stream()));
    F.n_sync++;
}

// option host_turn = fast | copy (default fast): how the state of a finished device-driven sweep reaches the host and how a restart's
// Q reaches the device.  fast: kernels that write to / read from pinned host memory, the host spins on a sequence word (with
// the stream's own status as the arbiter: an error or a completed stream without the word falls back to the copy).
bool fast_host_turn()
{
    const char* v = option("host_turn");
    return !v || std::string(v) != "copy";
}

// d_state -> *F.h_state, waited for.  One host synchronisation either way (n_sync counts it).
void fetch_state(mispec_fac& F)
{
    StepState& hs = *F.h_state.p;
    if (!fast_host_turn())
    {
        MISPEC_HIP(hipMemcpyAsync(&hs, F.d_state.p, sizeof(StepState), hipMemcpyDeviceToHost, F.stream()));
        sync_stream(F);
        F.turn_t0 = std::chrono::steady_clock::now();
        F.turn_open = true;
        return;
    }
    const unsigned long long want = ++F.pub_seq;
    launch_publish_state(*F.ctx, F.d_state.p, F.m, F.h_state.p, F.h_flag.p, want);
    F.n_sync++;
    volatile unsigned long long* flag = F.h_flag.p;
    bool seen = false;
    for (uint64_t spins = 0;; spins++)
    {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == want)
        {
            seen = true;
            break;
        }
        if ((spins & 0x3fff) == 0x3fff)  // every ~16 K polls: is the stream still alive?
        {
            const hipError_t q = hipStreamQuery(F.stream());
            if (q == hipSuccess)
                break;  // everything enqueued has run: the word must be there now, or this memory is not coherent (fallback)
            if (q != hipErrorNotReady)
                MISPEC_HIP(q);
        }
        __builtin_ia32_pause();
    }
    if (!seen && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != want)
    {
        F.turn_fallbacks++;
        MISPEC_HIP(hipMemcpyAsync(&hs, F.d_state.p, sizeof(StepState), hipMemcpyDeviceToHost, F.stream()));
        MISPEC_HIP(hipStreamSynchronize(F.stream()));
    }
    F.turn_t0 = std::chrono::steady_clock::now();
    F.turn_open = true;
}

void comm_check(int rc, const char* what)
{
    if (rc != MISPEC_OK)
        throw Error(MISPEC_ERUNTIME, std::string(what) + " failed: " + mispec_last_error());
}

// vec (nloc entries, caller's order) -> stored order, in place; and back
void to_stored_order(mispec_fac& F, double* vec)
{
    if (!F.perm_mode)
        return;
    launch_to_stored_order(*F.A, vec, F.pscratch.p);
    MISPEC_HIP(hipMemcpyAsync(vec, F.pscratch.p, size_t(F.nloc) * sizeof(double), hipMemcpyDeviceToDevice, F.stream()));
}
void from_stored_order(mispec_fac& F, double* vec)
{
    if (!F.perm_mode)
        return;
    launch_from_stored_order(*F.A, vec, F.pscratch.p);
    MISPEC_HIP(hipMemcpyAsync(vec, F.pscratch.p, size_t(F.nloc) * sizeof(double), hipMemcpyDeviceToDevice, F.stream()));
}
// the product with the stored matrix when this factorisation works in its order, else the order-preserving one
void spmv_of(mispec_fac& F, const mispec_csr& M, const double* x, double* y, const SpmvEpilogue* epi, hipEvent_t e0 = nullptr,
             hipEvent_t e1 = nullptr)
{
    if (F.perm_mode)
        launch_spmv_raw(M, x, y, epi, e0, e1);
    else
        launch_spmv(M, x, y, epi, e0, e1);
}

void allreduce(mispec_fac& F, double* buf, int64_t count)
{
    if (F.sharded())
    {
        Timed t(F, FAM_ALLRED);
        comm_check(F.ctx->comm.allreduce_sum(F.ctx->comm.user, buf, count, F.stream()), "all-reduce");
    }
}

// max over ranks of a device scalar (sum-only communicator: every rank contributes into its own slot)
void allreduce_max_scalar(mispec_fac& F, double* dev_scalar)
{
    if (!F.sharded())
        return;
    const int W = F.ctx->world();
    MISPEC_HIP(hipMemsetAsync(F.gmax.p, 0, size_t(W) * sizeof(double), F.stream()));
    MISPEC_HIP(hipMemcpyAsync(F.gmax.p + F.ctx->rank(), dev_scalar, sizeof(double), hipMemcpyDeviceToDevice, F.stream()));
    allreduce(F, F.gmax.p, W);
    std::vector<double> h(static_cast<size_t>(W));
    MISPEC_HIP(hipMemcpyAsync(h.data(), F.gmax.p, size_t(W) * sizeof(double), hipMemcpyDeviceToHost, F.stream()));
    sync_stream(F);
    double mx = 0.0;
    for (double v : h)
        mx = std::max(mx, v);
    MISPEC_HIP(hipMemcpyAsync(dev_scalar, &mx, sizeof(double), hipMemcpyHostToDevice, F.stream()));
    sync_stream(F);
}

// Which row-blocks can be multiplied before the exchange has landed (collective-free: a local property of the shard).
// MISPEC_OVERLAP=0 turns the overlap off.
void plan_overlap(mispec_fac& F)
{
    F.interior_first = F.interior_count = 0;
    if (!F.A || !F.sharded() || F.ctx->world() < 2 || F.A2 || F.Bop || F.Chol || F.A->spmv_format() >= 3)  // tiles, staged: no row sub-ranges
        return;
    const char* e = option("overlap");
    if (e && atoi(e) == 0)
        return;
    int first = 0, count = 0;
    interior_blocks(*F.A, F.row_begin, F.row_begin + F.nloc, first, count);
    const int nblocks = spmv_num_blocks(F.nloc);
    if (count < nblocks / 4 || count == nblocks)  // too little to hide anything behind / nothing to wait for
        return;
    F.interior_first = first;
    F.interior_count = count;
    if (!F.comm_stream)
    {
        MISPEC_HIP(hipStreamCreateWithFlags(&F.comm_stream, hipStreamNonBlocking));
        MISPEC_HIP(hipEventCreateWithFlags(&F.ev_x_ready, hipEventDisableTiming));
        MISPEC_HIP(hipEventCreateWithFlags(&F.ev_x_landed, hipEventDisableTiming));
    }
}

// Decide between the all-gather and the neighbour exchange for this matrix (collective: every rank calls it and
// every rank reaches the same decision, because the decision is a function of the all-gathered table).
void plan_exchange(mispec_fac& F)
{
    F.halo = false;
    const mispec_comm& cm = F.ctx->comm;
    const int W = cm.world, me = cm.rank;
    if (!F.A || !F.sharded() || !cm.exchange || W < 2)
        return;
    const char* e = option("exchange");
    if (e && std::string(e) == "allgather")
        return;
    std::vector<int64_t> lo, hi;
    const bool have = column_ranges(*F.A, F.block, W, lo, hi);
    // table[q][2p], table[q][2p+1]: first row and row count of rank p's slice that rank q reads
    DevBuf<double> mine, table;
    mine.alloc(2 * size_t(W));
    table.alloc(2 * size_t(W) * W);
    std::vector<double> h(2 * size_t(W), 0.0);
    for (int p = 0; p < W; p++)
    {
        if (!have)
        {
            h[2 * size_t(p)] = -1.0;  // "cannot tell": forces the all-gather on every rank
            continue;
        }
        if (p == me || hi[size_t(p)] < 0)
            continue;
        h[2 * size_t(p)] = double(lo[size_t(p)]);
        h[2 * size_t(p) + 1] = double(hi[size_t(p)] - lo[size_t(p)] + 1);
    }
    MISPEC_HIP(hipMemcpyAsync(mine.p, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, F.stream()));
    comm_check(cm.allgather(cm.user, mine.p, table.p, 2 * int64_t(W), F.stream()), "all-gather (exchange plan)");
    std::vector<double> t(2 * size_t(W) * W);
    MISPEC_HIP(hipMemcpyAsync(t.data(), table.p, t.size() * sizeof(double), hipMemcpyDeviceToHost, F.stream()));
    MISPEC_HIP(hipStreamSynchronize(F.stream()));
    int64_t worst = 0;
    for (int q = 0; q < W; q++)
    {
        int64_t total = 0;
        for (int p = 0; p < W; p++)
        {
            if (t[(size_t(q) * W + p) * 2] < 0.0)
                return;
            total += int64_t(t[(size_t(q) * W + p) * 2 + 1]);
        }
        worst = std::max(worst, total);
    }
    const bool force = e && std::string(e) == "halo";
    if (!force && 2 * worst > int64_t(W - 1) * F.block)
        return;  // the referenced parts are most of the vector: one all-gather is the better collective
    F.send_off.assign(size_t(W), 0);
    F.send_count.assign(size_t(W), 0);
    F.recv_off.assign(size_t(W), 0);
    F.recv_count.assign(size_t(W), 0);
    F.halo_recv = 0;
    for (int p = 0; p < W; p++)
    {
        if (p == me)
            continue;
        // what I read of p's slice lands at its global position in x_full
        F.recv_off[size_t(p)] = int64_t(t[(size_t(me) * W + p) * 2]);
        F.recv_count[size_t(p)] = int64_t(t[(size_t(me) * W + p) * 2 + 1]);
        F.halo_recv += F.recv_count[size_t(p)];
        // what p reads of my slice, relative to my first row
        F.send_count[size_t(p)] = int64_t(t[(size_t(p) * W + me) * 2 + 1]);
        F.send_off[size_t(p)] = F.send_count[size_t(p)] ? int64_t(t[(size_t(p) * W + me) * 2]) - F.row_begin : 0;
        MISPEC_REQUIRE(F.send_off[size_t(p)] >= 0 && F.send_off[size_t(p)] + F.send_count[size_t(p)] <= F.nloc,
                       "exchange plan: a peer references rows outside this shard");
    }
    F.halo = true;
}

// ---- B-inner products (generalized problems) -------------------------------------------------------------
int persistent_grid_records(const mispec_fac& F)
{
    const int64_t g = std::min<int64_t>(int64_t(F.ctx->num_cu) * 4, (F.nloc + 255) / 256);
    return int(std::max<int64_t>(g, 1));
}
// F.bx = B y
void b_apply(mispec_fac& F, const double* y)
{
    launch_spmv(F.Bcsr ? *F.Bcsr : *F.Bop->B, y, F.bx.p, nullptr);
}
// out_dev[0] = x' B y  (device scalar, fixed-order two-stage sum)
void b_inner_to(mispec_fac& F, const double* x, const double* y, double* out_dev)
{
    b_apply(F, y);
    const int nrec = persistent_grid_records(F);
    launch_dot_record(*F.ctx, x, F.bx.p, F.nloc, F.partials.p, F.pstride, nrec);
    launch_reduce_sum(*F.ctx, F.partials.p + int64_t(kSlotBeta2) * F.pstride, nrec, out_dev);
}
// After an orthogonalisation launch that left `nrec` records for c = V'(B x) (taken with F.bx = B x): overwrite
// the two scalar slots with x'Bx and max|x|, so that the usual reduction yields the B-norm.
void b_norm_slots(mispec_fac& F, const double* x, int nrec)
{
    launch_dot_record(*F.ctx, x, F.bx.p, F.nloc, F.partials.p, F.pstride, nrec);
}

// The product of a row shard while its exchange is still in flight on the communication stream: first the row-blocks that
// read only this rank's slice of x, then — once the other slices have landed — the blocks before and after them.  Same
// kernels on the same blocks as the single launch, so the same y and the same alpha records.
void overlapped_spmv(mispec_fac& F, const mispec_csr& M, const double* x, double* y, const SpmvEpilogue* epi, hipEvent_t e0, hipEvent_t e1)
{
    const int nblocks = spmv_num_blocks(F.nloc);
    const int i0 = F.interior_first, i1 = F.interior_first + F.interior_count;
    if (e0)
        MISPEC_HIP(hipEventRecord(e0, F.stream()));
    launch_spmv_raw(M, x, y, epi, nullptr, nullptr, i0, i1 - i0);
    {
        Timed t(F, FAM_XWAIT);  // (profile level 3: what is left of the exchange once the interior blocks are done)
        MISPEC_HIP(hipStreamWaitEvent(F.stream(), F.ev_x_landed, 0));
    }
    launch_spmv_raw(M, x, y, epi, nullptr, nullptr, 0, i0);
    launch_spmv_raw(M, x, y, epi, nullptr, nullptr, i1, nblocks - i1);
    if (e1)
        MISPEC_HIP(hipEventRecord(e1, F.stream()));
}

// y = Op(x).  x_loc / y_loc: this shard's rows (device).  With `lanczos_epi`, additionally
// y -= h_prev * v_prev (when v_prev != nullptr) and alpha = <x, y> is left in F.alpha_slot()
// (device) — Lanczos.h:131-142.
void apply_op(mispec_fac& F, const double* x_loc, double* y_loc, bool lanczos_epi, const double* v_prev, double h_prev,
              const double* h_prev_dev = nullptr, const int* status = nullptr)
{
    double* alpha_dev = F.alpha_slot();
    if (F.A)
    {
        const double* x = x_loc;
        const bool overlap = F.sharded() && F.interior_count > 0;
        hipStream_t xs = overlap ? F.comm_stream : F.stream();  // the stream the exchange is enqueued on
        if (F.sharded())
        {
            // own slice by a local copy (the interior row-blocks read nothing else); the collective then runs in place /
            // on the referenced parts only
            MISPEC_HIP(hipMemcpyAsync(F.xfull.p + F.row_begin, x_loc, size_t(F.nloc) * sizeof(double), hipMemcpyDeviceToDevice,
                                      F.stream()));
            if (overlap)
            {
                MISPEC_HIP(hipEventRecord(F.ev_x_ready, F.stream()));
                MISPEC_HIP(hipStreamWaitEvent(F.comm_stream, F.ev_x_ready, 0));
            }
            {
                Timed t(F, FAM_EXCH, xs);
                if (F.halo)
                    // the matrix references only parts of the other slices: concurrent point-to-point transfers of exactly those
                    comm_check(F.ctx->comm.exchange(F.ctx->comm.user, F.xfull.p + F.row_begin, F.send_off.data(), F.send_count.data(),
                                                    F.xfull.p, F.recv_off.data(), F.recv_count.data(), xs),
                               "neighbour exchange");
                else  // all-gather of the Krylov vector over xGMI, in place: blocks are equal-sized and padded
                    comm_check(F.ctx->comm.allgather(F.ctx->comm.user, F.xfull.p + int64_t(F.ctx->rank()) * F.block, F.xfull.p, F.block, xs),
                               "all-gather");
            }
            if (overlap)
                MISPEC_HIP(hipEventRecord(F.ev_x_landed, F.comm_stream));
            x = F.xfull.p;
        }
        // A single SpMV is timed through its own dispatch (start/stop of the kernel, no marker packets in the
        // stream); the two-kernel product operator through an event pair around both.
        const bool own_events = F.prof && !F.A2;
        std::unique_ptr<Timed> scope;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (own_events)
        {
            F.counts[FAM_SPMV]++;
            e0 = take_event(F);
            e1 = take_event(F);
            F.ev[FAM_SPMV].emplace_back(e0, e1);
        }
        else
            scope.reset(new Timed(F, FAM_SPMV));
        if (F.Chol)
        {
            // y = L^{-1} A L^{-T} x  (SymGEigsCholeskyOp.h:63-71): two dense triangular-inverse GEMVs around the SpMV
            scope.reset();
            {
                Timed t(F, FAM_SPMV);
                launch_cholesky_solve(*F.Chol, true, x, F.mid.p);
                launch_spmv(*F.A, F.mid.p, F.bx.p, nullptr);
                launch_cholesky_solve(*F.Chol, false, F.bx.p, y_loc);
            }
            if (lanczos_epi)
            {
                launch_lanczos_epilogue(*F.ctx, y_loc, x_loc, v_prev, h_prev, F.nloc, F.alpha_partials.p, h_prev_dev, status);
                if (!F.skip_alpha_reduce)
                    launch_reduce_sum(*F.ctx, F.alpha_partials.p, lanczos_epilogue_records(*F.ctx, F.nloc), alpha_dev);
            }
            return;
        }
        if (F.Bop)
        {
            // y = B^{-1} (A x)  (SymGEigsRegInvOp.h:76-81); the Lanczos epilogue in the B-inner product follows below
            scope.reset();
            {
                Timed t(F, FAM_SPMV);
                launch_spmv(*F.A, x, F.mid.p, nullptr);
                reginv_solve(*F.Bop, F.mid.p, y_loc);
            }
            if (lanczos_epi)
            {
                if (v_prev)  // w -= H(i,i-1) v_prev  (Lanczos.h:138-139); its plain-dot by-product is not used
                    launch_lanczos_epilogue(*F.ctx, y_loc, x_loc, v_prev, h_prev, F.nloc, F.alpha_partials.p);
                b_inner_to(F, x_loc, y_loc, alpha_dev);  // H(i,i) = <v, w>_B  (:142)
            }
            return;
        }
        const mispec_csr* last = F.A;
        if (F.A2)  // y = A2 (A x): the epilogue rides on the second product
        {
            launch_spmv(*F.A, x, F.mid.p, nullptr);
            x = F.mid.p;
            last = F.A2;
        }
        if (lanczos_epi)
        {
            SpmvEpilogue epi;
            epi.v_rows = x_loc;
            epi.v_prev = v_prev;
            epi.h_prev = h_prev;
            epi.h_prev_dev = h_prev_dev;
            epi.status = status;
            epi.partials = F.alpha_partials.p;
            if (F.post_scale_step > 0)
            {
                epi.post_scale_state = F.d_state.p;
                epi.post_scale_step = F.post_scale_step;
                epi.post_scale_eps_sqrt = std::sqrt(kEps);
            }
            if (overlap)
                overlapped_spmv(F, *last, x, y_loc, &epi, e0, e1);
            else
                spmv_of(F, *last, x, y_loc, &epi, e0, e1);
        }
        else if (overlap)
            overlapped_spmv(F, *last, x, y_loc, nullptr, e0, e1);
        else
            spmv_of(F, *last, x, y_loc, nullptr, e0, e1);
    }
    else if (F.S && F.Bcsr)
    {
        // generalized shift modes (SymGEigsShiftInvertOp.h:70-75, SymGEigsBucklingOp.h:53-57, SymGEigsCayleyOp.h:88-99):
        // y = (A - sigma B)^{-1} B x, and for Cayley y = x + 2 sigma * that
        {
            Timed t(F, FAM_SPMV);
            b_apply(F, x_loc);
            launch_shiftsolve(*F.S, F.bx.p, y_loc);
            if (F.cayley)
                (void) launch_axpby(*F.ctx, y_loc, 2.0 * F.cay_sigma, x_loc, 1.0, F.nloc, F.partials.p, F.pstride);
        }
        if (lanczos_epi)
        {
            if (v_prev)
                launch_lanczos_epilogue(*F.ctx, y_loc, x_loc, v_prev, h_prev, F.nloc, F.alpha_partials.p);
            b_inner_to(F, x_loc, y_loc, alpha_dev);
        }
        return;
    }
    else if (F.S)
    {
        Timed t(F, FAM_SPMV);
        launch_shiftsolve(*F.S, x_loc, y_loc);
        if (lanczos_epi)
            launch_lanczos_epilogue(*F.ctx, y_loc, x_loc, v_prev, h_prev, F.nloc, F.alpha_partials.p, h_prev_dev, status);
    }
    else if (F.D)
    {
        Timed t(F, FAM_SPMV);
        launch_row_gemv(*F.ctx, F.D->a.p, F.D->ld, F.D->rows, F.D->cols, x_loc, y_loc, true);
        if (lanczos_epi)
            launch_lanczos_epilogue(*F.ctx, y_loc, x_loc, v_prev, h_prev, F.nloc, F.alpha_partials.p, h_prev_dev, status);
    }
    else if (F.dop)
    {
        // user operator on device pointers: it enqueues its work on the factorisation's stream, nothing is staged
        {
            Timed t(F, FAM_SPMV);
            if (F.dop(F.dop_user, x_loc, y_loc, static_cast<void*>(F.stream())) != 0)
                throw Error(MISPEC_ERUNTIME, "user device operator callback reported failure");
        }
        if (lanczos_epi)
            launch_lanczos_epilogue(*F.ctx, y_loc, x_loc, v_prev, h_prev, F.nloc, F.alpha_partials.p, h_prev_dev, status);
    }
    else
    {
        // user operator with the reference's host-pointer contract (SymEigsSolver.h:43-51): staged through pinned memory
        MISPEC_HIP(hipMemcpyAsync(F.h_x.p, x_loc, size_t(F.n) * sizeof(double), hipMemcpyDeviceToHost, F.stream()));
        sync_stream(F);
        const int rc = F.op(F.op_user, F.h_x.p, F.h_y.p);
        if (rc != 0)
            throw Error(MISPEC_ERUNTIME, "user perform_op callback reported failure");
        MISPEC_HIP(hipMemcpyAsync(y_loc, F.h_y.p, size_t(F.n) * sizeof(double), hipMemcpyHostToDevice, F.stream()));
        F.counts[FAM_SPMV]++;
        if (lanczos_epi)
            launch_lanczos_epilogue(*F.ctx, y_loc, x_loc, v_prev, h_prev, F.nloc, F.alpha_partials.p);
    }
    if (lanczos_epi && !F.skip_alpha_reduce)  // (one-reduction steps: the partial sums travel with the record of the previous pass)
    {
        const int64_t nparts = F.A ? spmv_num_blocks(F.nloc) : lanczos_epilogue_records(*F.ctx, F.nloc);
        launch_reduce_sum(*F.ctx, F.alpha_partials.p, nparts, alpha_dev);
        allreduce(F, alpha_dev, 1);
    }
}

// Reduce the per-workgroup records of the last orth/axpby launch into red_buf(which); `fin` is the scalar
// tail executed on the device after the (all-)reduction.  No host synchronisation.
void reduce_record(mispec_fac& F, int nrec, int ncol, int which, const FinishArgs& fin)
{
    double* red = F.red_buf(which);
    if (!F.sharded())
        launch_reduce_partials(*F.ctx, F.partials.p, F.pstride, nrec, ncol, red, fin);
    else
    {
        FinishArgs none;
        none.mode = kFinishNone;
        none.packed = 1;
        none.alpha_parts = fin.alpha_parts;  // one-reduction steps: this rank's <f~, A f~> is packed behind sum f^2
        none.alpha_count = fin.alpha_count;
        launch_reduce_partials(*F.ctx, F.partials.p, F.pstride, nrec, ncol, F.red_stage(), none);
        allreduce(F, F.red_stage(), ncol + 1 + (fin.alpha_parts ? 1 : 0));  // the sums: slots [0, ncol) and, packed behind them, sum f^2
        FinishArgs tail = fin;
        tail.packed = 1;
        launch_finish(*F.ctx, F.red_stage(), red, ncol, tail);
    }
    F.red_cur = which;
}

// Same, then bring the record to the host (h_red).  One stream synchronisation.
void reduce_to_host(mispec_fac& F, int nrec, int ncol, int which)
{
    FinishArgs fin;  // norms only
    reduce_record(F, nrec, ncol, which, fin);
    MISPEC_HIP(hipMemcpyAsync(F.h_red.p, F.red_buf(which), kPartialLd * sizeof(double), hipMemcpyDeviceToHost, F.stream()));
    sync_stream(F);
}

OrthArgs orth_args(mispec_fac& F, int ncol)
{
    OrthArgs a;
    a.V = F.V.p;
    a.ldv = F.ldv;
    a.ncol = ncol;
    a.n = F.nloc;
    a.partials = F.partials.p;
    a.pstride = F.pstride;
    return a;
}

// c = V[:, :ncol]' x ; returns with h_red = {c, |x|^2, ...}
void vtf(mispec_fac& F, const double* x, int ncol, int which)
{
    OrthArgs a = orth_args(F, ncol);
    a.src = x;
    int nrec;
    {
        Timed t(F, FAM_VTF);
        if (F.bmode())  // c = V'(B x), |x|_B  (ArnoldiOp.h:68-101)
        {
            b_apply(F, x);
            a.src = F.bx.p;
        }
        nrec = launch_orth(*F.ctx, ORTH_VTF, a);
        if (F.bmode())
            b_norm_slots(F, x, nrec);
    }
    reduce_to_host(F, nrec, ncol, which);
}

// dst = src - V[:, :ncol] c (c = red_buf(F.red_cur)[0..ncol) on the device), then |dst| and V'dst
// in the same pass.  Result record in the other half of `red`.
void correct_vtf(mispec_fac& F, const double* src, double* dst, int ncol)
{
    OrthArgs a = orth_args(F, ncol);
    a.src = src;
    a.dst = dst;
    a.c_in = F.red_buf(F.red_cur);
    int nrec;
    {
        Timed t(F, FAM_GEMV);
        if (!F.bmode())
            nrec = launch_orth(*F.ctx, ORTH_CORRECT_VTF, a);
        else
        {
            (void) launch_orth(*F.ctx, ORTH_CORRECT_ONLY, a);  // dst = src - V c
            b_apply(F, dst);
            OrthArgs b = orth_args(F, ncol);
            b.src = F.bx.p;
            nrec = launch_orth(*F.ctx, ORTH_VTF, b);  // V'(B dst)
            b_norm_slots(F, dst, nrec);               // |dst|_B
        }
    }
    reduce_to_host(F, nrec, ncol, F.red_cur ^ 1);
}

// f = w - alpha v (alpha: device scalar), then |f| and V[:, :ncol]' f — Lanczos.h:145-153 / Arnoldi.h:177.
// Returns the record count; the caller reduces.
int resid_vtf(mispec_fac& F, const double* w, const double* v, const double* alpha_dev, double* f, int ncol)
{
    OrthArgs a = orth_args(F, F.bmode() ? 0 : ncol);
    a.src = w;
    a.dst = f;
    a.vi = v;
    a.alpha_dev = alpha_dev;
    Timed t(F, FAM_VTF);
    int nrec = launch_orth(*F.ctx, ORTH_RESID_VTF, a);
    if (F.bmode())  // the plain by-products of that launch are replaced by the B-inner-product ones
    {
        b_apply(F, f);
        OrthArgs b = orth_args(F, ncol);
        b.src = F.bx.p;
        nrec = launch_orth(*F.ctx, ORTH_VTF, b);
        b_norm_slots(F, f, nrec);
    }
    return nrec;
}

double host_norm(const double* x, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; i++)
        s += x[i] * x[i];
    return std::sqrt(s);
}

// Arnoldi.h:66-115.  On return f is (numerically) orthogonal to V[:, :ncol] and F.beta = |f|.
void expand_basis(mispec_fac& F, int ncol, int64_t seed, int64_t* nmatop)
{
    for (int iter = 0; iter < 5; iter++)
    {
        const uint64_t s = uint64_t(seed + 123 * iter);
        if (iter == 0)
        {
            launch_simple_random(*F.ctx, F.tmp.p, F.row_begin, F.nloc, s);  // :76
            to_stored_order(F, F.tmp.p);
            apply_op(F, F.tmp.p, F.f.p, false, nullptr, 0.0);               // :79  f = A * rand
            (*nmatop)++;
        }
        else
        {
            launch_simple_random(*F.ctx, F.f.p, F.row_begin, F.nloc, s);  // :84
            to_stored_order(F, F.f.p);
        }
        vtf(F, F.f.p, ncol, 0);                                         // :87
        correct_vtf(F, F.f.p, F.f.p, ncol);                             // :88-93 (f -= V Vf ; |f| ; V'f)
        double fnorm = F.h_red.p[kSlotBeta];
        double ortho_err = F.h_red.p[kSlotErr];
        int count = 0;
        while (count < 3 && ortho_err >= kEps * fnorm)  // :98-108
        {
            correct_vtf(F, F.f.p, F.f.p, ncol);
            fnorm = F.h_red.p[kSlotBeta];
            ortho_err = F.h_red.p[kSlotErr];
            count++;
        }
        F.beta = fnorm;
        if (ortho_err < kEps * fnorm)  // :112
            return;
    }
}

void zero_vector(mispec_fac& F, double* v)
{
    MISPEC_HIP(hipMemsetAsync(v, 0, size_t(F.ldv) * sizeof(double), F.stream()));
}

// Common start of Arnoldi::init once the start vector is in F.tmp (Arnoldi.h:145-195).
void init_from_tmp(mispec_fac& F, int64_t* nmatop)
{
    F.end_pending = false;
    F.restart_unresolved = false;
    F.eager_sticky = false;
    std::fill(F.H.begin(), F.H.end(), 0.0);
    MISPEC_HIP(hipMemsetAsync(F.V.p, 0, F.V.n * sizeof(double), F.stream()));
    zero_vector(F, F.f.p);
    zero_vector(F, F.w.p);

    vtf(F, F.tmp.p, 0, 0);  // :146 v0norm
    const double v0norm = F.h_red.p[kSlotBeta];
    if (v0norm < kNear0)
        throw Error(MISPEC_EINVAL, "initial residual vector cannot be zero");

    double* v = F.col(0);
    apply_op(F, F.tmp.p, v, false, nullptr, 0.0);  // :153  v = A v0
    (*nmatop)++;
    vtf(F, v, 0, 0);  // :157
    const double vnorm = F.h_red.p[kSlotBeta];
    {
        Timed t(F, FAM_SCALE);
        if (vnorm < kNear0)
            launch_scale(*F.ctx, F.tmp.p, v, F.ldv, v0norm);  // :162-165 v0 is in the null space of A
        else
            launch_scale(*F.ctx, v, v, F.ldv, vnorm);  // :168
    }
    apply_op(F, v, F.w.p, true, nullptr, 0.0);  // :173-176  w = A v ; H(0,0) = <v, w>
    (*nmatop)++;

    // f = w - v H(0,0)  (:177) ; the V'f by-product is not used here
    const int nrec = resid_vtf(F, F.w.p, v, F.alpha_slot(), F.f.p, 1);
    MISPEC_HIP(hipMemcpyAsync(F.h_red.p + kPartialLd, F.alpha_slot(), sizeof(double), hipMemcpyDeviceToHost,
                              F.stream()));
    reduce_to_host(F, nrec, 1, 1);
    const double alpha = F.h_red.p[kPartialLd];
    if (F.sharded())
    {
        allreduce_max_scalar(F, F.red_buf(1) + kSlotMaxAbs);
        MISPEC_HIP(hipMemcpyAsync(F.h_red.p + kSlotMaxAbs, F.red_buf(1) + kSlotMaxAbs, sizeof(double), hipMemcpyDeviceToHost,
                                  F.stream()));
        sync_stream(F);
    }
    F.Hat(0, 0) = alpha;
    if (F.h_red.p[kSlotMaxAbs] < kEps * std::fabs(alpha))  // :183-187
    {
        zero_vector(F, F.f.p);
        F.beta = 0.0;
    }
    else
        F.beta = F.h_red.p[kSlotBeta];
    F.k = 1;
}

void zero_H_outside(mispec_fac& F, int from_k)  // Lanczos.h:85-86 / Arnoldi.h:219-220
{
    const int m = F.m;
    for (int j = from_k; j < m; j++)
        for (int i = 0; i < m; i++)
            F.Hat(i, j) = 0.0;
    for (int j = 0; j < from_k; j++)
        for (int i = from_k; i < m; i++)
            F.Hat(i, j) = 0.0;
}

// The while loop of Lanczos.h:156-182 on the host path, entered with `count` corrections already applied and
// h_red / red_cur describing the latest record of step i.
void lanczos_corrections_host(mispec_fac& F, int i, int count)
{
    const double beta_thresh = kEps * std::sqrt(double(F.n));
    const int i1 = i + 1;
    double ortho_err = F.h_red.p[kSlotErr];
    while (count < 5 && ortho_err > kEps * F.beta)  // :156
    {
        if (F.beta < beta_thresh)  // :163-168
        {
            zero_vector(F, F.f.p);
            F.beta = 0.0;
            break;
        }
        const double c_im1 = F.h_red.p[i - 1], c_i = F.h_red.p[i];
        correct_vtf(F, F.f.p, F.f.p, i1);  // :171, :177, :179 — one pass over V
        F.Hat(i - 1, i) += c_im1;          // :173-175
        F.Hat(i, i - 1) = F.Hat(i - 1, i);
        F.Hat(i, i) += c_i;
        F.beta = F.h_red.p[kSlotBeta];
        ortho_err = F.h_red.p[kSlotErr];
        count++;
    }
}

// MISPEC_SMALL=device keeps the m x m work of a restart on the GPU (tested in both settings); the default is the host core.
bool small_on_device()
{
    const bool on = option_is("small", "device");
    return on;
}

// One-sweep steps: apply the correction that the last step of the sweep left pending, then continue the reference's loop
// (Lanczos.h:156-182) from "one correction applied".  H and beta already carry that correction (finish_lagged).
void corrections_after_fused_restart(mispec_fac& F, bool force);
// what the host learns about a sync-free fused restart once the state has come back (diagnostics and the sticky eager mode)
void absorb_restart_state(mispec_fac& F, const StepState& hs)
{
    F.beta = hs.beta;
    F.lag_chk_max = std::max(F.lag_chk_max, hs.rst_beta_corr > 0.0 ? hs.rst_err / hs.rst_beta_corr : 0.0);
    if (hs.rst_err > 0.5 * kEps * hs.rst_beta_corr)
        F.eager_sticky = true;  // the Ritz values of this restart were computed before the correction: do not repeat that
}
// A sync-free fused restart whose outcome some host code needs now (f, beta, H, a host-driven step): one synchronisation.
void resolve_restart(mispec_fac& F)
{
    if (!F.restart_unresolved)
        return;
    F.restart_unresolved = false;
    StepState& hs = *F.h_state.p;
    fetch_state(F);
    F.turn_open = false;
    absorb_restart_state(F, hs);
    if (hs.status == kStepRestartCheck)
    {
        F.fused_recorrected++;
        corrections_after_fused_restart(F, false);
    }
}

void finish_pending(mispec_fac& F)
{
    resolve_restart(F);
    if (!F.end_pending)
        return;
    F.end_pending = false;
    F.red_cur = F.end_rec;
    correct_vtf(F, F.f.p, F.f.p, F.m);  // :171, :177, :179
    F.beta = F.h_red.p[kSlotBeta];
    lanczos_corrections_host(F, F.m - 1, 1);
}

// One whole step of Lanczos.h:88-183 with every decision taken on the host (2-3 stream synchronisations).
// Used for user operators, for the rare restart / breakdown branches, and when MISPEC_HOST_STEPS=1.
void lanczos_step_host(mispec_fac& F, int i, int64_t* nmatop)
{
    const double eps_sqrt = std::sqrt(kEps);
    bool restart = (F.beta < kNear0);  // :99
    double* v = F.col(i);
    if (!restart)
    {
        {
            Timed t(F, FAM_SCALE);
            launch_scale(*F.ctx, F.f.p, v, F.ldv, F.beta);  // :106
        }
        if (F.beta < eps_sqrt)  // :107-113 (rare)
        {
            OrthArgs a = orth_args(F, 1);
            a.V = F.col(i - 1);
            a.src = v;
            int nrec;
            {
                Timed t(F, FAM_VTF);
                if (F.bmode())  // <V[:, i-1], v>_B
                {
                    b_apply(F, v);
                    a.src = F.bx.p;
                }
                nrec = launch_orth(*F.ctx, ORTH_VTF, a);
            }
            reduce_to_host(F, nrec, 1, 0);
            restart = (std::fabs(F.h_red.p[0]) > eps_sqrt);
        }
    }
    if (restart)
    {
        expand_basis(F, i, 2 * int64_t(i), nmatop);  // :117-118
        Timed t(F, FAM_SCALE);
        launch_scale(*F.ctx, F.f.p, v, F.ldv, F.beta);  // :119
    }
    F.Hat(i, i - 1) = restart ? 0.0 : F.beta;  // :127-128
    F.Hat(i - 1, i) = F.Hat(i, i - 1);

    // w = A v ; w -= H(i,i-1) V[:,i-1] ; alpha = <v, w>   (:131-142) — one kernel
    apply_op(F, v, F.w.p, true, restart ? nullptr : F.col(i - 1), F.Hat(i, i - 1));
    (*nmatop)++;

    // f = w - alpha v ; beta = |f| ; Vf = V[:, :i+1]' f   (:145-153) — one pass over V
    const int i1 = i + 1;
    const int nrec = resid_vtf(F, F.w.p, v, F.alpha_slot(), F.f.p, i1);
    MISPEC_HIP(hipMemcpyAsync(F.h_red.p + kPartialLd, F.alpha_slot(), sizeof(double), hipMemcpyDeviceToHost,
                              F.stream()));
    reduce_to_host(F, nrec, i1, 1);
    F.Hat(i, i) = F.h_red.p[kPartialLd];
    F.beta = F.h_red.p[kSlotBeta];
    lanczos_corrections_host(F, i, 0);
}

// One step of the device-driven path: every kernel is enqueued, nothing is read back.  The scalar decisions
// of the step (restart test, need for a correction, breakdown) are taken by the finish code on the device and
// recorded in F.d_state; a kernel that must not run any more turns itself into a no-op.
// Correction passes enqueued ahead of the decision.  The first one practically always runs (V'f of the plain
// three-term recurrence is above eps*beta), a second one almost never: enqueueing it blindly costs three no-op
// launches on one device, but a real all-reduce per step when sharded — so it is left to the host path there.
int speculative_corrections(const mispec_fac& F)
{
    const int knob = option_int("spec_corr", 0);
    if (knob >= 1 && knob <= 4)
        return knob;
    return (F.sharded() && F.ctx->world() > 1) ? 1 : 2;
}
void lanczos_step_device(mispec_fac& F, int i)
{
    const int kSpeculativeCorrections = speculative_corrections(F);
    StepState* st = F.d_state.p;
    double* v = F.col(i);
    {
        Timed t(F, FAM_SCALE);
        launch_scale_step(*F.ctx, F.f.p, v, F.ldv, st, i, std::sqrt(kEps));
    }
    apply_op(F, v, F.w.p, true, F.col(i - 1), 0.0, &st->subd[i - 1], &st->status);

    const int i1 = i + 1;
    FinishArgs fin;
    fin.st = st;
    fin.step = i;
    fin.eps = kEps;
    fin.beta_thresh = kEps * std::sqrt(double(F.n));
    fin.max_spec = kSpeculativeCorrections;
    {
        OrthArgs a = orth_args(F, i1);
        a.src = F.w.p;
        a.dst = F.f.p;
        a.vi = v;
        a.alpha_dev = F.alpha_slot();
        a.status = &st->status;
        Timed t(F, FAM_VTF);
        F.count_bytes(FAM_VTF, i1 + 2);  // i + 1 columns (v_i among them) and w read, f written
        const int nrec = launch_orth(*F.ctx, ORTH_RESID_VTF, a);
        fin.mode = kFinishStepFirst;
        fin.alpha_src = F.alpha_slot();
        reduce_record(F, nrec, i1, 1, fin);
    }
    for (int c = 0; c < kSpeculativeCorrections; c++)
    {
        OrthArgs a = orth_args(F, i1);
        a.src = F.f.p;
        a.dst = F.f.p;
        a.c_in = F.red_buf(F.red_cur);
        a.status = &st->status;
        a.need_corr = &st->need_corr;
        Timed t(F, FAM_GEMV);
        const int nrec = launch_orth(*F.ctx, ORTH_CORRECT_VTF, a);
        fin.mode = kFinishStepCorr;
        fin.prev_red = F.red_buf(F.red_cur);
        reduce_record(F, nrec, i1, F.red_cur ^ 1, fin);
    }
}

// One step of the opt-in one-sweep variant (CPU restatement: oracle/onesweep_variant.hpp).  The operator is
// applied to the not yet corrected column i; the pass that follows finishes column i with the correction measured in the
// previous step, forms the next residual and measures its V'f — one sweep over V per step instead of two.  Records
// alternate between the two halves of `red`: step i writes half (i & 1) and takes its coefficients from the other one.
// `last`: the sweep ends here, so the residual is finished the reference's way by the CORRECT_VTF launches that follow —
// unless `defer`: then the correction stays pending like that of any other step and the restart applies it (F.end_pending).
void lanczos_step_lagged(mispec_fac& F, int i, bool last, bool defer)
{
    StepState* st = F.d_state.p;
    double* v = F.col(i);
    // Column i is written by the pass below (from f, with the pending correction); until then only the product needs f / beta.
    // On diagonal storage the SpMV takes the un-normalised f and divides its row sums instead (csr.hpp post_scale_state): the
    // scaling pass and its copy of f disappear.  Other formats: k_scale_step, as in the reference flow.
    const bool post = F.A && !F.A2 && !F.Chol && !F.perm_mode && spmv_can_post_scale(*F.A);  // the plain product operator only
    // One reduction per step: the previous step of this run left its record unreduced (F.lag_def).  The product does not wait for
    // beta: u = A f~ with the partial sums of <f~, u>; then ONE reduction — that record and the sums — whose tail finishes step
    // i - 1, takes this step's small-beta stop and forms alpha~; the pass below turns u into w = u / beta - beta v_{i-1} itself.
    const bool onered = F.lag_def.have;
    if (onered)
    {
        F.skip_alpha_reduce = true;
        apply_op(F, F.f.p, F.w.p, true, nullptr, 0.0, nullptr, &st->status);
        F.skip_alpha_reduce = false;
        FinishArgs fin = F.lag_def.fin;
        fin.alpha_parts = F.alpha_partials.p;
        fin.alpha_count = (F.A && !F.Chol) ? spmv_num_blocks(F.nloc) : lanczos_epilogue_records(*F.ctx, F.nloc);
        fin.alpha_out = F.alpha_slot();
        {
            Timed t(F, FAM_VTF, nullptr, false);
            Timed t2(F, FAM_REDUCE);
            reduce_record(F, F.lag_def.nrec, F.lag_def.ncol, F.lag_def.half, fin);
        }
        F.lag_def.have = false;
    }
    else if (post)
    {
        F.post_scale_step = i;
        apply_op(F, F.f.p, F.w.p, true, F.col(i - 1), 0.0, &st->subd[i - 1], &st->status);
        F.post_scale_step = 0;
    }
    else
    {
        {
            Timed t(F, FAM_SCALE);
            launch_scale_step(*F.ctx, F.f.p, v, F.ldv, st, i, std::sqrt(kEps));
        }
        apply_op(F, v, F.w.p, true, F.col(i - 1), 0.0, &st->subd[i - 1], &st->status);
    }

    const int cur = i & 1;
    FinishArgs fin;
    fin.st = st;
    fin.step = i;
    fin.eps = kEps;
    fin.beta_thresh = kEps * std::sqrt(double(F.n));
    fin.eps_sqrt = std::sqrt(kEps);
    fin.lag_limit = F.lag_limit;
    fin.lag_last = (last && !defer) ? 1 : 0;
    fin.max_spec = speculative_corrections(F);
    {
        OrthArgs a = orth_args(F, i);
        a.src = F.w.p;
        a.vi = F.f.p;
        a.dst = F.f.p;
        a.vout = v;
        a.c_in = F.red_buf(cur ^ 1);
        a.alpha_dev = F.alpha_slot();
        a.beta_dev = &st->beta;
        a.pending = &st->lag_pending;
        a.status = &st->status;
        a.onered = onered ? 1 : 0;
        Timed t(F, FAM_VTF);
        F.count_bytes(FAM_VTF, i + 4);  // i columns, f and w read; column i and f written
        const int nrec = launch_orth(*F.ctx, ORTH_LAGGED, a);
        fin.mode = kFinishLagged;
        fin.alpha_src = F.alpha_slot();
        fin.prev_red = F.red_buf(cur ^ 1);
        // the plain matrix product on bases of one column panel; the last step of a sweep is reduced at once (what follows — the
        // reference's corrections or the restart — needs its record)
        // (every operator the library applies itself: matrices incl. the SVD product and the Cholesky mode of the generalized
        // problem, the banded / dense shift solve, dense matrices — on bases of up to 128 columns; user operators on device
        // pointers keep receiving the normalised vector)
        const bool defer_record = F.onered && !last && (F.A || F.S || F.D) && !F.dop;
        if (defer_record)
        {
            F.lag_def.have = true;
            F.lag_def.nrec = nrec;
            F.lag_def.ncol = 2 * i + 1;
            F.lag_def.half = cur;
            F.lag_def.fin = fin;
            F.red_cur = cur;
        }
        else
            reduce_record(F, nrec, 2 * i + 1, cur, fin);
    }
    if (!last || defer)
        return;
    const int i1 = i + 1;
    for (int c = 0; c < fin.max_spec; c++)
    {
        OrthArgs a = orth_args(F, i1);
        a.src = F.f.p;
        a.dst = F.f.p;
        a.c_in = F.red_buf(F.red_cur);
        a.status = &st->status;
        a.need_corr = &st->need_corr;
        Timed t(F, FAM_GEMV);
        const int nrec = launch_orth(*F.ctx, ORTH_CORRECT_VTF, a);
        fin.mode = kFinishStepCorr;
        fin.prev_red = F.red_buf(F.red_cur);
        reduce_record(F, nrec, i1, F.red_cur ^ 1, fin);
    }
}

// operators applied entirely by enqueued device work (apply_op never waits for the host)
// MISPEC_ONE_REDUCTION = 1 | 0: the default of the one-reduction form of the lagged steps (mispec_fac_set_orth_mode's
// MISPEC_ORTH_ONE_REDUCTION / MISPEC_ORTH_TWO_REDUCTIONS select it per factorisation)
bool default_one_reduction()
{
    const char* e = option("one_reduction");
    return e ? atoi(e) != 0 : true;  // the default since round 5 (C2: 0.986 -> 0.939 s per solve, same counters; profiles/r09l)
}

bool device_operator(const mispec_fac& F) { return F.A != nullptr || (F.S != nullptr && F.Bcsr == nullptr) || F.D != nullptr || F.dop != nullptr; }

// Lanczos.h:62-187
void factorize_lanczos(mispec_fac& F, int from_k, int to_m, int64_t* nmatop)
{
    zero_H_outside(F, from_k);
    // device-driven steps: every operator that works on device pointers without a host turn — device matrices (incl. the SVD
    // solver's product), and since round 4 the banded / dense shift-solve, dense matrices and user operators on device pointers
    // (their Lanczos epilogue is a kernel of its own that reads H(i,i-1) and the stop flag from device memory), and the Cholesky
    // mode of the generalized problem: L^{-1} A L^{-T} is a standard symmetric operator made of three enqueued products
    const bool fast = F.device_steps && device_operator(F) && !F.bmode();
    // standard problems (incl. the product operator of the SVD solver); bases of up to 128 columns (k_orth_lagged with 4 or 8 wavefronts)
    const bool lagged = fast && F.onesweep && F.m <= 2 * kPanelCols;
    // a sweep that completes the factorisation is followed by a restart (or by nothing that needs f): its last correction can wait
    // — for the fused restart (k_vq_fused: one column panel); wider bases finish every sweep the reference's way
    const bool defer = lagged && F.m <= kPanelCols && to_m == F.m && !F.eager_last && !F.eager_sticky && !small_on_device();
    F.end_pending = false;
    if (F.restart_unresolved && !(lagged && from_k == F.k))
        resolve_restart(F);  // (mispec_fac_factorize resolves through finish_pending unless this sweep can start from the device state)
    int i = from_k;
    while (i <= to_m - 1)
    {
        if (!fast)
        {
            lanczos_step_host(F, i, nmatop);
            i++;
            continue;
        }
        // ---- device-driven run of steps i .. to_m-1 -------------------------------------------------
        StepState& hs = *F.h_state.p;
        const bool from_restart = F.restart_unresolved;  // the start state is already in d_state (mispec_fac_restart_sym)
        F.restart_unresolved = false;
        if (!from_restart)
        {
            std::memset(&hs, 0, sizeof(StepState));
            hs.beta = F.beta;
            hs.status = kStepOk;
            for (int j = 0; j < F.m; j++)
            {
                hs.diag[j] = F.Hat(j, j);
                hs.subd[j] = (j + 1 < F.m) ? F.Hat(j + 1, j) : 0.0;
            }
            MISPEC_HIP(hipMemcpyAsync(F.d_state.p, &hs, sizeof(StepState), hipMemcpyHostToDevice, F.stream()));
        }
        F.lag_def.have = false;
        for (int s = i; s <= to_m - 1; s++)
        {
            if (lagged)
                lanczos_step_lagged(F, s, s == to_m - 1, defer);
            else
                lanczos_step_device(F, s);
        }
        fetch_state(F);

        const int status = hs.status;
        if (from_restart)
        {
            absorb_restart_state(F, hs);
            if (status == kStepRestartCheck)  // no step ran: the reference's loop on the compressed factorisation, then the sweep again
            {
                F.fused_recorrected++;
                corrections_after_fused_restart(F, false);
                continue;
            }
        }
        const int stop = (status == kStepOk) ? to_m : hs.stop_step;
        const int last_done = (status == kStepOk) ? to_m - 1 : ((status == kStepSmallBeta || status == kStepLagCheck) ? stop - 1 : stop);
        if (lagged)
        {
            F.lag_steps += hs.lag_steps;
            F.onered_steps += hs.onered_steps;
            F.lag_rel_c_max = std::max(F.lag_rel_c_max, hs.lag_rel_c_max);
            F.lag_chk_max = std::max(F.lag_chk_max, hs.lag_chk_max);
        }
        for (int j = i; j <= last_done; j++)  // bring H of the executed steps home
        {
            F.Hat(j, j) = hs.diag[j];
            F.Hat(j, j - 1) = F.Hat(j - 1, j) = hs.subd[j - 1];
        }
        *nmatop += (last_done - i + 1);
        F.beta = hs.beta;
        if (status == kStepOk)
        {
            if (defer && hs.lag_pending)
            {
                F.end_pending = true;
                F.end_rec = (to_m - 1) & 1;
            }
            break;
        }
        // ---- the rare branches continue on the host path, then the device path resumes ---------------
        if (status == kStepLagCheck)
        {
            // one-sweep variant: column `stop` (once corrected) fails the reference's test for a second correction.  Back to
            // the state of the reference's loop for step stop-1 after its first correction: f = beta * column, V'f = beta *
            // the measured V'v, count = 1; the step `stop` itself is then repeated (its speculative product is not counted)
            F.lag_check_stops++;
            const int rec = stop & 1;
            MISPEC_HIP(hipMemcpyAsync(F.h_red.p, F.red_buf(rec), kPartialLd * sizeof(double), hipMemcpyDeviceToHost, F.stream()));
            sync_stream(F);
            double err = 0.0;
            for (int j = 0; j < stop; j++)
            {
                F.h_red.p[j] = F.h_red.p[stop + 1 + j] * F.beta;
                err = std::max(err, std::fabs(F.h_red.p[j]));
            }
            F.h_red.p[kSlotErr] = err;
            F.h_red.p[kSlotBeta] = F.beta;
            MISPEC_HIP(hipMemcpyAsync(F.red_buf(rec), F.h_red.p, size_t(stop) * sizeof(double), hipMemcpyHostToDevice, F.stream()));
            {
                Timed t(F, FAM_SCALE);
                launch_scale(*F.ctx, F.col(stop), F.f.p, F.ldv, 1.0 / F.beta);
            }
            F.red_cur = rec;
            lanczos_corrections_host(F, stop - 1, 1);
            i = stop;
            continue;
        }
        if (status == kStepSmallBeta)
            lanczos_step_host(F, stop, nmatop);
        else
        {
            if (lagged)
                F.lag_state_stops++;
            F.red_cur = (hs.stop_count % 2 == 0) ? 1 : 0;  // RESID -> red[1], then the corrections alternate
            if (lagged)  // the lagged record of step `stop` sits in half (stop & 1), corrections alternate from there
                F.red_cur = (stop & 1) ^ (hs.stop_count & 1);
            MISPEC_HIP(hipMemcpyAsync(F.h_red.p, F.red_buf(F.red_cur), kPartialLd * sizeof(double), hipMemcpyDeviceToHost,
                                      F.stream()));
            sync_stream(F);
            lanczos_corrections_host(F, stop, hs.stop_count);  // kStepTinyF takes the clamp branch at once
        }
        i = stop + 1;
    }
    F.k = to_m;
}

// The correction loop of an Arnoldi step (Arnoldi.h:264-291), host decisions.  On entry F.h_red holds the
// record of the last f (V'f, beta, err) and red_buf(F.red_cur) the same on the device.
void arnoldi_corrections_host(mispec_fac& F, int i)
{
    const double beta_thresh = kEps * std::sqrt(double(F.n));
    const int i1 = i + 1;
    double* h = &F.Hat(0, i);
    double ortho_err = F.h_red.p[kSlotErr];
    int count = 0;
    while (count < 5 && ortho_err > kEps * F.beta)  // :266
    {
        if (F.beta < beta_thresh)
        {
            zero_vector(F, F.f.p);
            F.beta = 0.0;
            break;
        }
        double Vf[kMaxCols];
        for (int j = 0; j < i1; j++)
            Vf[j] = F.h_red.p[j];
        correct_vtf(F, F.f.p, F.f.p, i1);  // :281, :285, :287
        for (int j = 0; j < i1; j++)
            h[j] += Vf[j];  // :283
        F.beta = F.h_red.p[kSlotBeta];
        ortho_err = F.h_red.p[kSlotErr];
        count++;
    }
}

// One Arnoldi step with the host reading every scalar back (Arnoldi.h:225-292).
void arnoldi_step_host(mispec_fac& F, int i, int64_t* nmatop)
{
    bool restart = false;
    if (F.beta < kNear0)  // :228-233
    {
        expand_basis(F, i, 2 * int64_t(i), nmatop);
        restart = true;
    }
    double* v = F.col(i);
    {
        Timed t(F, FAM_SCALE);
        launch_scale(*F.ctx, F.f.p, v, F.ldv, F.beta);  // :236
    }
    F.Hat(i, i - 1) = restart ? 0.0 : F.beta;  // :239
    apply_op(F, v, F.w.p, false, nullptr, 0.0);  // :242
    (*nmatop)++;

    const int i1 = i + 1;
    vtf(F, F.w.p, i1, 0);  // h = V' w  (:251)
    double* h = &F.Hat(0, i);
    for (int j = 0; j < i1; j++)
        h[j] = F.h_red.p[j];
    // f = w - V h ; beta = |f| ; Vf = V' f   (:254-255, :262) — one pass over V
    correct_vtf(F, F.w.p, F.f.p, i1);
    F.beta = F.h_red.p[kSlotBeta];
    if (F.beta > 0.717 * host_norm(h, i1))  // :257
        return;
    arnoldi_corrections_host(F, i);
}

// The same step enqueued without reading anything back: h, |h|, beta and the 0.717 test live in device
// memory; a step that needs the correction loop (or the restart branch) stops the device run.
void arnoldi_step_device(mispec_fac& F, int i)
{
    StepState* st = F.d_state.p;
    double* v = F.col(i);
    {
        Timed t(F, FAM_SCALE);
        launch_scale_step(*F.ctx, F.f.p, v, F.ldv, st, i, kNear0);  // :228 (restart branch -> host), :236, :239
    }
    apply_op(F, v, F.w.p, false, nullptr, 0.0);  // :242

    const int i1 = i + 1;
    FinishArgs fin;
    fin.st = st;
    fin.step = i;
    fin.eps = kEps;
    fin.beta_thresh = kEps * std::sqrt(double(F.n));
    fin.hcol = F.d_H.p + size_t(i) * F.m;
    {
        OrthArgs a = orth_args(F, i1);
        a.src = F.w.p;
        a.status = &st->status;
        Timed t(F, FAM_VTF);
        const int nrec = launch_orth(*F.ctx, ORTH_VTF, a);
        fin.mode = kFinishArnoldiH;
        reduce_record(F, nrec, i1, 0, fin);
    }
    {
        OrthArgs a = orth_args(F, i1);
        a.src = F.w.p;
        a.dst = F.f.p;
        a.c_in = F.red_buf(0);
        a.status = &st->status;
        Timed t(F, FAM_GEMV);
        const int nrec = launch_orth(*F.ctx, ORTH_CORRECT_VTF, a);
        fin.mode = kFinishArnoldiF;
        reduce_record(F, nrec, i1, 1, fin);
    }
}

// Arnoldi.h:198-295
void factorize_arnoldi(mispec_fac& F, int from_k, int to_m, int64_t* nmatop)
{
    zero_H_outside(F, from_k);
    const bool fast = F.device_steps && device_operator(F);
    int i = from_k;
    while (i <= to_m - 1)
    {
        if (!fast)
        {
            arnoldi_step_host(F, i, nmatop);
            i++;
            continue;
        }
        // ---- device-driven run of steps i .. to_m-1 -------------------------------------------------
        StepState& hs = *F.h_state.p;
        std::memset(&hs, 0, sizeof(StepState));
        hs.beta = F.beta;
        hs.status = kStepOk;
        MISPEC_HIP(hipMemcpyAsync(F.d_state.p, &hs, sizeof(StepState), hipMemcpyHostToDevice, F.stream()));
        for (int s = i; s <= to_m - 1; s++)
            arnoldi_step_device(F, s);
        MISPEC_HIP(hipMemcpyAsync(&hs, F.d_state.p, sizeof(StepState), hipMemcpyDeviceToHost, F.stream()));
        MISPEC_HIP(hipMemcpyAsync(F.h_H.p, F.d_H.p, size_t(F.m) * F.m * sizeof(double), hipMemcpyDeviceToHost, F.stream()));
        sync_stream(F);

        const int status = hs.status;
        const int stop = (status == kStepOk) ? to_m : hs.stop_step;
        const int last_done = (status == kStepOk) ? to_m - 1 : (status == kStepSmallBeta ? stop - 1 : stop);
        for (int j = i; j <= last_done; j++)  // bring H of the executed steps home
        {
            F.Hat(j, j - 1) = hs.subd[j - 1];
            for (int r = 0; r <= j; r++)
                F.Hat(r, j) = F.h_H.p[size_t(j) * F.m + r];
        }
        *nmatop += (last_done - i + 1);
        F.beta = hs.beta;
        if (status == kStepOk)
            break;
        // ---- the rare branches continue on the host path, then the device path resumes ---------------
        if (status == kStepSmallBeta)
            arnoldi_step_host(F, stop, nmatop);
        else
        {
            F.red_cur = 1;  // VTF -> red[0], f = w - Vh -> red[1]
            MISPEC_HIP(hipMemcpyAsync(F.h_red.p, F.red_buf(1), kPartialLd * sizeof(double), hipMemcpyDeviceToHost, F.stream()));
            sync_stream(F);
            arnoldi_corrections_host(F, stop);  // kStepTinyF takes the clamp branch at once
        }
        i = stop + 1;
    }
    F.k = to_m;
}

void require_init(const mispec_fac* F, const char* who)
{
    if (!F)
        throw Error(MISPEC_EINVAL, std::string(who) + ": fac is NULL");
    if (F->k < 1)
        throw Error(MISPEC_ELOGIC, std::string(who) + ": need to call init first");
}

// V[:, :p] <- V Q with the m x m Q in F.Qdev (Arnoldi.h:326-335).  In place for ncv <= 64; a wider basis goes through
// the eigenvector workspace, because a panelled product cannot overwrite its own input.
void compress_basis(mispec_fac& F, int p)
{
    const int m = F.m;
    if (m <= kPanelCols)
    {
        F.count_bytes(FAM_COMPRESS, m + p);
        launch_vq(*F.ctx, F.V.p, F.ldv, m, F.Qdev.p, m, p, F.V.p, F.ldv, F.nloc);
        return;
    }
    if (F.X.n < size_t(F.ldv) * size_t(p))
        F.X.alloc(size_t(F.ldv) * size_t(p));
    F.x_cols = 0;  // whatever Ritz vectors were held there are gone
    launch_vq(*F.ctx, F.V.p, F.ldv, m, F.Qdev.p, m, p, F.X.p, F.ldv, F.nloc);
    MISPEC_HIP(hipMemcpyAsync(F.V.p, F.X.p, size_t(F.ldv) * size_t(p) * sizeof(double), hipMemcpyDeviceToDevice, F.stream()));
}

// f <- f*Q(m-1,k-1) + V[:,k]*H(k,k-1) ; beta = |f|   (Arnoldi.h:337-339)
void update_f_after_compress(mispec_fac& F, double q_last, double h_sub)
{
    int nrec;
    {
        Timed t(F, FAM_COMPRESS);
        nrec = launch_axpby(*F.ctx, F.f.p, q_last, F.col(F.k), h_sub, F.nloc, F.partials.p, F.pstride);
        if (F.bmode())  // beta = |f|_B  (Arnoldi.h:339 through ArnoldiOp::norm)
        {
            b_apply(F, F.f.p);
            b_norm_slots(F, F.f.p, nrec);
        }
    }
    reduce_to_host(F, nrec, 0, 0);
    F.beta = F.h_red.p[kSlotBeta];
}

// Device columns -> a pageable host matrix (the reference's eigenvectors() / matrix_V() return host matrices).  A direct
// hipMemcpy into pageable memory runs at a few GB/s (the runtime stages it in small pieces and the first touch of a fresh
// 1.6 GB allocation faults every page on one thread); here 64 MB pieces go D2H into two pinned buffers at PCIe rate while the host
// threads copy the piece before into the caller's memory (first touch spread over the threads).
void download_columns(mispec_fac& F, const double* src, int64_t ld_src, int64_t rows, int ncols, double* dst, int64_t ld_dst)
{
    if (rows <= 0 || ncols <= 0)
        return;
    constexpr int64_t kPiece = int64_t(8) << 20;        // doubles per piece (64 MB)
    constexpr int64_t kDirect = int64_t(4) << 20;       // up to 32 MB in total: one strided copy, no staging, no threads
    const int64_t total = rows * int64_t(ncols);
    if (total <= kDirect)
    {
        MISPEC_HIP(hipMemcpy2DAsync(dst, size_t(ld_dst) * sizeof(double), src, size_t(ld_src) * sizeof(double), size_t(rows) * sizeof(double),
                                    size_t(ncols), hipMemcpyDeviceToHost, F.stream()));
        MISPEC_HIP(hipStreamSynchronize(F.stream()));
        F.n_sync++;
        return;
    }
    // A destination this large is usually a fresh allocation (the reference's eigenvectors() returns a new matrix per call): its
    // first touch by the copy threads below would fault it in 4 KiB at a time — 1.6 GB at n = 1e7 cost 50-70 ms, more than the
    // transfer.  Ask for transparent huge pages on the 2 MiB-aligned interior (a hint; ignored where THP is off or the range is
    // already populated).
    {
        const uintptr_t lo = (reinterpret_cast<uintptr_t>(dst) + (uintptr_t(2) << 20) - 1) & ~((uintptr_t(2) << 20) - 1);
        const uintptr_t hi = reinterpret_cast<uintptr_t>(dst + int64_t(ncols - 1) * ld_dst + rows) & ~((uintptr_t(2) << 20) - 1);
        if (hi > lo)
            (void) madvise(reinterpret_cast<void*>(lo), size_t(hi - lo), MADV_HUGEPAGE);
    }
    // staging sized to the transfer (a solver object that only ever returns small matrices pins nothing)
    const int64_t piece = std::min(kPiece, total);
    for (int b = 0; b < 2; b++)
    {
        if (F.h_stage[b].n < size_t(piece))
            F.h_stage[b].alloc(size_t(piece));
        if (!F.ev_stage[b])
            MISPEC_HIP(hipEventCreateWithFlags(&F.ev_stage[b], hipEventDisableTiming));
    }
    struct Piece
    {
        double* dst;     // first destination column of the piece
        int64_t count;   // doubles per column
        int cols;        // columns in the piece (> 1 only when whole columns are packed)
        int64_t ld_dst;
    };
    Piece pending[2] = {{nullptr, 0, 0, 0}, {nullptr, 0, 0, 0}};
    auto drain = [&](int b) {
        if (!pending[b].dst)
            return;
        MISPEC_HIP(hipEventSynchronize(F.ev_stage[b]));
        const double* from = F.h_stage[b].p;
        const Piece pc = pending[b];
        const int64_t all = pc.count * pc.cols;
        // one host thread per MB or so: a small piece is copied inline
        const int nt = int(std::max<int64_t>(1, std::min<int64_t>(std::min(ingest_threads(), 32), all >> 17)));
        parallel_ranges(all, nt, [&](int, int64_t lo, int64_t hi) {
            while (lo < hi)  // [lo, hi) of the packed piece, column by column
            {
                const int64_t c = lo / pc.count, r = lo - c * pc.count;
                const int64_t len = std::min(hi - lo, pc.count - r);
                std::memcpy(pc.dst + c * pc.ld_dst + r, from + lo, size_t(len) * sizeof(double));
                lo += len;
            }
        });
        pending[b].dst = nullptr;
    };
    int b = 0;
    if (rows <= piece)
    {
        // whole columns, several per piece, packed densely in the staging buffer by one strided copy
        const int per = int(std::max<int64_t>(1, piece / rows));
        for (int j = 0; j < ncols; j += per)
        {
            const int nc = std::min(per, ncols - j);
            drain(b);  // this buffer's previous piece must be out before it is overwritten
            MISPEC_HIP(hipMemcpy2DAsync(F.h_stage[b].p, size_t(rows) * sizeof(double), src + int64_t(j) * ld_src, size_t(ld_src) * sizeof(double),
                                        size_t(rows) * sizeof(double), size_t(nc), hipMemcpyDeviceToHost, F.stream()));
            MISPEC_HIP(hipEventRecord(F.ev_stage[b], F.stream()));
            pending[b] = {dst + int64_t(j) * ld_dst, rows, nc, ld_dst};
            b ^= 1;
            drain(b);  // while the piece just enqueued travels, copy the other buffer's
        }
    }
    else
    {
        for (int j = 0; j < ncols; j++)
            for (int64_t r0 = 0; r0 < rows; r0 += piece)
            {
                const int64_t cnt = std::min(piece, rows - r0);
                drain(b);
                MISPEC_HIP(hipMemcpyAsync(F.h_stage[b].p, src + int64_t(j) * ld_src + r0, size_t(cnt) * sizeof(double), hipMemcpyDeviceToHost, F.stream()));
                MISPEC_HIP(hipEventRecord(F.ev_stage[b], F.stream()));
                pending[b] = {dst + int64_t(j) * ld_dst + r0, cnt, 1, ld_dst};
                b ^= 1;
                drain(b);
            }
    }
    drain(0);
    drain(1);
    F.n_sync++;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
namespace {
int fac_create_impl(mispec_ctx* ctx, const mispec_csr* A, const mispec_symshift* S, mispec_op_fn op, void* op_user, int64_t n,
                    int ncv, int symmetric, mispec_fac** out, const mispec_csr* A2 = nullptr, const mispec_reginv* Bop = nullptr,
                    const mispec_csr* Bcsr = nullptr, bool cayley = false, double cay_sigma = 0.0,
                    const mispec_cholesky* Chol = nullptr, const mispec_dense* D = nullptr, mispec_device_op_fn dop = nullptr,
                    void* dop_user = nullptr)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && out, "mispec_fac_create: NULL argument");
        MISPEC_REQUIRE(int(A != nullptr) + int(S != nullptr) + int(op != nullptr) + int(D != nullptr) + int(dop != nullptr) == 1,
                       "mispec_fac_create: give exactly one operator");
        if (D)
            MISPEC_REQUIRE(D->ctx == ctx && D->rows == n && D->cols == n, "mispec_fac_create_dense: matrix belongs to another context / is not n x n");
        MISPEC_REQUIRE(n >= 1, "mispec_fac_create: n must be positive");
        MISPEC_REQUIRE(ncv >= 1 && ncv <= n, "mispec_fac_create: need 1 <= ncv <= n");
        MISPEC_REQUIRE(ncv <= kMaxCols, "mispec_fac_create: the device factorisation holds at most 1024 basis vectors (ncv <= 1024)");
        if (Bop)
        {
            MISPEC_REQUIRE(A && !A2 && symmetric, "mispec_fac_create_geigs_reginv: needs a symmetric device matrix A");
            MISPEC_REQUIRE(Bop->ctx == ctx && Bop->n == n, "mispec_fac_create_geigs_reginv: B belongs to another context / size");
            MISPEC_REQUIRE(ctx->comm.allgather == nullptr, "mispec_fac_create_geigs_reginv: generalized problems cannot be row-sharded");
        }
        if (Chol)
        {
            MISPEC_REQUIRE(A && !A2 && !Bop && symmetric, "mispec_fac_create_geigs_cholesky: needs a symmetric device matrix A");
            MISPEC_REQUIRE(Chol->ctx == ctx && Chol->n == n, "mispec_fac_create_geigs_cholesky: B belongs to another context / size");
            MISPEC_REQUIRE(Chol->info == 0, "SymGEigsSolver: the Cholesky factorisation of B failed (B must be positive definite)");
            MISPEC_REQUIRE(ctx->comm.allgather == nullptr, "mispec_fac_create_geigs_cholesky: generalized problems cannot be row-sharded");
        }
        if (Bcsr)
        {
            MISPEC_REQUIRE(S && !A && symmetric, "mispec_fac_create_geigs_shift: needs a shift solver and the B matrix");
            MISPEC_REQUIRE(Bcsr->ctx == ctx && Bcsr->n_rows == n && Bcsr->n_cols == n,
                           "mispec_fac_create_geigs_shift: B belongs to another context / size");
        }
        if (A && A2)
        {
            MISPEC_REQUIRE(A->ctx == ctx && A2->ctx == ctx, "mispec_fac_create_product: matrix belongs to another context");
            MISPEC_REQUIRE(A->n_cols == n && A2->n_rows == n && A2->n_cols == A->n_rows,
                           "mispec_fac_create_product: need A (p x n) and A2 (n x p)");
            MISPEC_REQUIRE(ctx->comm.allgather == nullptr, "mispec_fac_create_product: product operators cannot be row-sharded");
        }
        else if (A)
        {
            MISPEC_REQUIRE(A->ctx == ctx, "mispec_fac_create: matrix belongs to another context");
            MISPEC_REQUIRE(A->n_rows == n && A->n_cols == n, "mispec_fac_create: operator must be square of size n");
        }
        else
            MISPEC_REQUIRE(ctx->comm.allgather == nullptr,
                           "mispec_fac_create: user operators and shift solvers cannot be row-sharded");
        if (S)
            MISPEC_REQUIRE(S->ctx == ctx && S->n == n, "mispec_fac_create: shift solver belongs to another context / size");
        ctx->make_current();
        auto* F = new mispec_fac();
        try
        {
            F->ctx = ctx;
            F->A = A;
            F->A2 = A2;
            F->Bop = Bop;
            if (A2 || Bop)
                F->mid.alloc(size_t(round_up(std::max<int64_t>(A->n_rows, 1), 2)) + 2);
            F->Chol = Chol;
            if (Chol)
            {
                F->mid.alloc(size_t(round_up(std::max<int64_t>(n, 1), 2)) + 2);
                F->bx.alloc(size_t(round_up(std::max<int64_t>(n, 1), 2)) + 2);
            }
            F->Bcsr = Bcsr;
            F->cayley = cayley;
            F->cay_sigma = cay_sigma;
            if (Bop || Bcsr)
            {
                F->bx.alloc(size_t(round_up(std::max<int64_t>(n, 1), 2)) + 2);
                MISPEC_HIP(hipMemsetAsync(F->bx.p, 0, F->bx.n * sizeof(double), ctx->stream));
            }
            F->S = S;
            F->op = op;
            F->op_user = op_user;
            F->D = D;
            F->dop = dop;
            F->dop_user = dop_user;
            F->n = n;
            F->m = ncv;
            F->symmetric = symmetric != 0;
            int64_t b, e;
            if (mispec_shard_range(n, ctx->world(), ctx->rank(), &b, &e) != MISPEC_OK)
                throw Error(MISPEC_EINVAL, mispec_last_error());
            F->row_begin = b;
            F->nloc = e - b;
            F->block = mispec_shard_block(n, ctx->world());
            // every n-vector is padded to an even length (16-byte vector accesses) and, when sharded, to the
            // all-gather block so that a column can be sent as one equal-sized block
            F->ldv = std::max<int64_t>(round_up(std::max<int64_t>(F->nloc, 1), 2), F->sharded() ? F->block : 0);
            F->H.assign(size_t(ncv) * ncv, 0.0);
            F->V.alloc(size_t(F->ldv) * ncv);
            F->f.alloc(size_t(F->ldv));
            F->w.alloc(size_t(F->ldv));
            F->tmp.alloc(size_t(F->ldv));
            MISPEC_HIP(hipMemsetAsync(F->V.p, 0, F->V.n * sizeof(double), ctx->stream));
            MISPEC_HIP(hipMemsetAsync(F->f.p, 0, F->f.n * sizeof(double), ctx->stream));
            MISPEC_HIP(hipMemsetAsync(F->w.p, 0, F->w.n * sizeof(double), ctx->stream));
            MISPEC_HIP(hipMemsetAsync(F->tmp.p, 0, F->tmp.n * sizeof(double), ctx->stream));
            if (F->sharded())
            {
                F->xfull.alloc(size_t(F->block) * ctx->world());
                MISPEC_HIP(hipMemsetAsync(F->xfull.p, 0, F->xfull.n * sizeof(double), ctx->stream));
                F->gmax.alloc(size_t(ctx->world()));
            }
            const int64_t max_rec = int64_t(ctx->num_cu) * 8 + 8;
            F->pstride = max_rec;
            F->partials.alloc(size_t(max_rec) * kPartialLd);
            const int64_t nparts = std::max<int64_t>(A ? spmv_num_blocks(F->nloc) : 0, lanczos_epilogue_records(*ctx, F->nloc));
            F->alpha_partials.alloc(size_t(std::max<int64_t>(nparts, 1)));
            F->red.alloc(3 * kPartialLd + 16);
            MISPEC_HIP(hipMemsetAsync(F->red.p, 0, F->red.n * sizeof(double), ctx->stream));
            F->Qdev.alloc(size_t(ncv) * ncv);
            F->d_diag.alloc(size_t(ncv));
            F->d_subd.alloc(size_t(ncv));
            F->d_evals.alloc(size_t(ncv));
            F->d_evecs.alloc(size_t(ncv) * ncv);
            F->d_Y.alloc(size_t(ncv) * ncv);
            F->d_info.alloc(1);
            F->d_state.alloc(1);
            if (!F->symmetric)
            {
                F->d_H.alloc(size_t(ncv) * ncv);
                F->h_H.alloc(size_t(ncv) * ncv);
            }
            F->h_state.alloc(1);
            F->h_flag.alloc(8);
            F->h_flag.p[0] = 0;
            F->h_up.alloc(size_t(ncv) * ncv + 2 * size_t(ncv));
            F->device_steps = option_int("host_steps", 0) == 0;
            {
                // default since round 4: the one-sweep steps (every gate of tests/test_gpu_onesweep.py and the reference's own test
                // programs hold in both modes); MISPEC_ORTH=reference restores the reference's two-pass control flow everywhere
                const char* o = option("orth");
                const std::string mode = o ? o : "onesweep";
                MISPEC_REQUIRE(mode == "onesweep" || mode == "reference" || mode == "onesweep-eager",
                               "MISPEC_ORTH: expected reference, onesweep or onesweep-eager");
                F->onesweep = mode != "reference";
                F->eager_last = mode == "onesweep-eager";
                F->onered = F->onesweep && default_one_reduction();
            }
            F->h_red.alloc(kPartialLd + 8);
            F->h_small.alloc(size_t(ncv) * ncv + 4 * size_t(ncv) + 8);
            if (op)
            {
                F->h_x.alloc(size_t(n));
                F->h_y.alloc(size_t(n));
            }
            F->perm_mode = A && A->reordered() && !A2 && !Bop && !Chol && !F->sharded();
            if (F->perm_mode)
                F->pscratch.alloc(size_t(F->ldv));
            MISPEC_HIP(hipStreamSynchronize(ctx->stream));
            plan_exchange(*F);
            plan_overlap(*F);
        }
        catch (...)
        {
            delete F;
            throw;
        }
        *out = F;
    });
}
}  // namespace

extern "C" int mispec_fac_create(mispec_ctx* ctx, const mispec_csr* A, mispec_op_fn op, void* op_user, int64_t n, int ncv,
                                 int symmetric, mispec_fac** out)
{
    return fac_create_impl(ctx, A, nullptr, op, op_user, n, ncv, symmetric, out);
}

extern "C" int mispec_fac_create_dense(mispec_ctx* ctx, const mispec_dense* D, int ncv, int symmetric, mispec_fac** out)
{
    if (!D)
    {
        set_last_error("mispec_fac_create_dense: NULL matrix");
        return MISPEC_EINVAL;
    }
    return fac_create_impl(ctx, nullptr, nullptr, nullptr, nullptr, D->rows, ncv, symmetric, out, nullptr, nullptr, nullptr, false, 0.0,
                           nullptr, D);
}

extern "C" int mispec_fac_create_device_op(mispec_ctx* ctx, mispec_device_op_fn op, void* op_user, int64_t n, int ncv, int symmetric,
                                           mispec_fac** out)
{
    if (!op)
    {
        set_last_error("mispec_fac_create_device_op: NULL callback");
        return MISPEC_EINVAL;
    }
    return fac_create_impl(ctx, nullptr, nullptr, nullptr, nullptr, n, ncv, symmetric, out, nullptr, nullptr, nullptr, false, 0.0,
                           nullptr, nullptr, op, op_user);
}

extern "C" int mispec_fac_create_product(mispec_ctx* ctx, const mispec_csr* A, const mispec_csr* A2, int ncv, mispec_fac** out)
{
    if (!A || !A2)
    {
        set_last_error("mispec_fac_create_product: NULL matrix");
        return MISPEC_EINVAL;
    }
    return fac_create_impl(ctx, A, nullptr, nullptr, nullptr, A->n_cols, ncv, 1, out, A2);
}

extern "C" int mispec_fac_create_geigs_reginv(mispec_ctx* ctx, const mispec_csr* A, const mispec_reginv* B, int ncv, mispec_fac** out)
{
    if (!A || !B)
    {
        set_last_error("mispec_fac_create_geigs_reginv: NULL operand");
        return MISPEC_EINVAL;
    }
    return fac_create_impl(ctx, A, nullptr, nullptr, nullptr, A->n_rows, ncv, 1, out, nullptr, B);
}

extern "C" int mispec_fac_create_geigs_shift(mispec_ctx* ctx, const mispec_symshift* S, const mispec_csr* B, int cayley, double sigma,
                                             int ncv, mispec_fac** out)
{
    if (!S || !B)
    {
        set_last_error("mispec_fac_create_geigs_shift: NULL operand");
        return MISPEC_EINVAL;
    }
    return fac_create_impl(ctx, nullptr, S, nullptr, nullptr, S->n, ncv, 1, out, nullptr, nullptr, B, cayley != 0, sigma);
}

extern "C" int mispec_fac_create_geigs_cholesky(mispec_ctx* ctx, const mispec_csr* A, const mispec_cholesky* B, int ncv, mispec_fac** out)
{
    if (!A || !B)
    {
        set_last_error("mispec_fac_create_geigs_cholesky: NULL operand");
        return MISPEC_EINVAL;
    }
    return fac_create_impl(ctx, A, nullptr, nullptr, nullptr, A->n_rows, ncv, 1, out, nullptr, nullptr, nullptr, false, 0.0, B);
}

extern "C" int mispec_fac_create_shiftsolve(mispec_ctx* ctx, const mispec_symshift* S, int ncv, int symmetric, mispec_fac** out)
{
    if (!S)
    {
        set_last_error("mispec_fac_create_shiftsolve: NULL solver");
        return MISPEC_EINVAL;
    }
    return fac_create_impl(ctx, nullptr, S, nullptr, nullptr, S->n, ncv, symmetric, out);
}

extern "C" int mispec_fac_destroy(mispec_fac* fac)
{
    return guarded([&] {
        if (fac)
        {
            fac->ctx->make_current();
            (void) hipStreamSynchronize(fac->ctx->stream);
            delete fac;
        }
    });
}

extern "C" int mispec_fac_init(mispec_fac* fac, const double* v0_host, int64_t* nmatop)
{
    return guarded([&] {
        MISPEC_REQUIRE(fac && v0_host && nmatop, "mispec_fac_init: NULL argument");
        mispec_fac& F = *fac;
        F.ctx->make_current();
        zero_vector(F, F.tmp.p);
        if (F.nloc)
            MISPEC_HIP(hipMemcpyAsync(F.tmp.p, v0_host + F.row_begin, size_t(F.nloc) * sizeof(double), hipMemcpyHostToDevice,
                                      F.stream()));
        sync_stream(F);  // v0_host may be pageable memory that the caller frees right after
        to_stored_order(F, F.tmp.p);
        init_from_tmp(F, nmatop);
    });
}

extern "C" int mispec_fac_init_random(mispec_fac* fac, uint64_t seed, int64_t* nmatop)
{
    return guarded([&] {
        MISPEC_REQUIRE(fac && nmatop, "mispec_fac_init_random: NULL argument");
        mispec_fac& F = *fac;
        F.ctx->make_current();
        zero_vector(F, F.tmp.p);
        launch_simple_random(*F.ctx, F.tmp.p, F.row_begin, F.nloc, seed);
        to_stored_order(F, F.tmp.p);  // the reference's start vector, entry i belonging to row i of the caller's matrix
        init_from_tmp(F, nmatop);
    });
}

extern "C" int mispec_fac_factorize(mispec_fac* fac, int from_k, int to_m, int64_t* nmatop)
{
    return guarded([&] {
        require_init(fac, "mispec_fac_factorize");
        MISPEC_REQUIRE(nmatop, "mispec_fac_factorize: nmatop is NULL");
        mispec_fac& F = *fac;
        F.ctx->make_current();
        if (to_m <= from_k)
            return;
        if (!(F.restart_unresolved && F.symmetric))  // (a sync-free restart is picked up by factorize_lanczos itself)
            finish_pending(F);
        MISPEC_REQUIRE(to_m <= F.m && from_k >= 1, "factorize_from: need 1 <= from_k < to_m <= ncv");
        if (from_k > F.k)  // Lanczos.h:70-75 / Arnoldi.h:206-211
            throw Error(MISPEC_EINVAL, std::string(F.symmetric ? "Lanczos" : "Arnoldi") + ": from_k (= " + std::to_string(from_k) +
                                           ") is larger than the current subspace dimension (= " + std::to_string(F.k) + ")");
        if (F.symmetric)
            factorize_lanczos(F, from_k, to_m, nmatop);
        else
            factorize_arnoldi(F, from_k, to_m, nmatop);
    });
}

extern "C" int mispec_fac_subspace_dim(const mispec_fac* fac) { return fac ? fac->k : 0; }
extern "C" int64_t mispec_fac_local_rows(const mispec_fac* fac) { return fac ? fac->nloc : 0; }

extern "C" int mispec_fac_overlap_info(const mispec_fac* fac, int* first_block, int* block_count, int* total_blocks)
{
    return guarded([&] {
        MISPEC_REQUIRE(fac, "mispec_fac_overlap_info: NULL argument");
        if (first_block)
            *first_block = fac->interior_first;
        if (block_count)
            *block_count = fac->interior_count;
        if (total_blocks)
            *total_blocks = fac->A ? spmv_num_blocks(fac->nloc) : 0;
    });
}

extern "C" int mispec_fac_set_orth_mode(mispec_fac* fac, int mode)
{
    return guarded([&] {
        MISPEC_REQUIRE(fac, "mispec_fac_set_orth_mode: NULL argument");
        const int base = mode & 0xff, flags = mode & ~0xff;
        MISPEC_REQUIRE((base == MISPEC_ORTH_REFERENCE && flags == 0) ||
                           (base == MISPEC_ORTH_ONESWEEP &&
                            (flags & ~(MISPEC_ORTH_EAGER_LAST | MISPEC_ORTH_TEST_RECORRECT | MISPEC_ORTH_TEST_RESTART_CHECK |
                                       MISPEC_ORTH_ONE_REDUCTION | MISPEC_ORTH_TWO_REDUCTIONS)) == 0),
                       "mispec_fac_set_orth_mode: unknown mode");
        fac->onesweep = (base == MISPEC_ORTH_ONESWEEP);
        fac->eager_last = (flags & MISPEC_ORTH_EAGER_LAST) != 0;
        fac->test_recorrect = (flags & MISPEC_ORTH_TEST_RECORRECT) != 0;
        fac->test_restart_check = (flags & MISPEC_ORTH_TEST_RESTART_CHECK) != 0;
        if (flags & MISPEC_ORTH_ONE_REDUCTION)
            fac->onered = true;
        else if (flags & MISPEC_ORTH_TWO_REDUCTIONS)
            fac->onered = false;
        else if (base == MISPEC_ORTH_ONESWEEP)
            fac->onered = default_one_reduction();
    });
}

extern "C" int mispec_fac_orth_info(const mispec_fac* fac, int* mode, int64_t* lagged_steps, int64_t* check_stops,
                                    int64_t* state_stops, double* max_rel_c, double* max_chk)
{
    return guarded([&] {
        MISPEC_REQUIRE(fac, "mispec_fac_orth_info: NULL argument");
        const bool active = fac->onesweep && fac->device_steps && fac->symmetric && device_operator(*fac) && !fac->bmode() && fac->m <= 2 * kPanelCols;
        if (mode)
            *mode = active ? (MISPEC_ORTH_ONESWEEP | ((fac->eager_last || fac->eager_sticky) ? MISPEC_ORTH_EAGER_LAST : 0) |
                              (fac->test_recorrect ? MISPEC_ORTH_TEST_RECORRECT : 0) |
                              (fac->test_restart_check ? MISPEC_ORTH_TEST_RESTART_CHECK : 0) |
                              (fac->onered ? MISPEC_ORTH_ONE_REDUCTION : 0))
                           : MISPEC_ORTH_REFERENCE;
        if (lagged_steps)
            *lagged_steps = fac->lag_steps;
        if (check_stops)
            *check_stops = fac->lag_check_stops;
        if (state_stops)
            *state_stops = fac->lag_state_stops;
        if (max_rel_c)
            *max_rel_c = fac->lag_rel_c_max;
        if (max_chk)
            *max_chk = fac->lag_chk_max;
    });
}

extern "C" int mispec_fac_onered_steps(const mispec_fac* fac, int64_t* steps)
{
    return guarded([&] {
        MISPEC_REQUIRE(fac && steps, "mispec_fac_onered_steps: NULL argument");
        *steps = fac->onered_steps;
    });
}

extern "C" int mispec_fac_restart_info(const mispec_fac* fac, int64_t* fused, int64_t* recorrected)
{
    return guarded([&] {
        MISPEC_REQUIRE(fac, "mispec_fac_restart_info: NULL argument");
        if (fused)
            *fused = fac->fused_restarts;
        if (recorrected)
            *recorrected = fac->fused_recorrected;
    });
}

extern "C" int mispec_fac_turn_info(const mispec_fac* fac, int64_t* turns, double* host_seconds, int64_t* fallbacks)
{
    return guarded([&] {
        MISPEC_REQUIRE(fac, "mispec_fac_turn_info: NULL argument");
        if (turns)
            *turns = fac->turn_count;
        if (host_seconds)
            *host_seconds = fac->turn_host_s;
        if (fallbacks)
            *fallbacks = fac->turn_fallbacks;
    });
}

extern "C" int mispec_fac_exchange_info(const mispec_fac* fac, int* halo, int64_t* recv_doubles)
{
    return guarded([&] {
        MISPEC_REQUIRE(fac && halo && recv_doubles, "mispec_fac_exchange_info: NULL argument");
        *halo = fac->halo ? 1 : 0;
        *recv_doubles = fac->halo ? fac->halo_recv : (fac->sharded() ? fac->block * (fac->ctx->world() - 1) : 0);
    });
}

extern "C" int mispec_fac_f_norm(const mispec_fac* fac, double* beta)
{
    return guarded([&] {
        MISPEC_REQUIRE(fac && beta, "mispec_fac_f_norm: NULL argument");
        if (fac->restart_unresolved)
        {
            fac->ctx->make_current();
            resolve_restart(*const_cast<mispec_fac*>(fac));
        }
        *beta = fac->beta;
    });
}

extern "C" int mispec_fac_get_H(const mispec_fac* fac, double* H_host)
{
    return guarded([&] {
        MISPEC_REQUIRE(fac && H_host, "mispec_fac_get_H: NULL argument");
        if (fac->end_pending || fac->restart_unresolved)  // a further correction, should the reference's loop take one, still changes H(m-1, m-2 : m-1)
        {
            fac->ctx->make_current();
            finish_pending(*const_cast<mispec_fac*>(fac));
        }
        std::memcpy(H_host, fac->H.data(), fac->H.size() * sizeof(double));
    });
}

extern "C" int mispec_fac_set_H(mispec_fac* fac, const double* H_host, int k)
{
    return guarded([&] {
        MISPEC_REQUIRE(fac && H_host && k >= 0 && k <= fac->m, "mispec_fac_set_H: bad argument");
        if (fac->end_pending || fac->restart_unresolved)
        {
            fac->ctx->make_current();
            finish_pending(*fac);
        }
        std::memcpy(fac->H.data(), H_host, fac->H.size() * sizeof(double));
        fac->k = k;
    });
}

extern "C" int mispec_fac_get_V(const mispec_fac* fac, int ncols, double* V_host)
{
    return guarded([&] {
        MISPEC_REQUIRE(fac && V_host && ncols >= 0 && ncols <= fac->m, "mispec_fac_get_V: bad argument");
        fac->ctx->make_current();
        MISPEC_HIP(hipStreamSynchronize(fac->ctx->stream));
        if (fac->perm_mode)
        {
            for (int j = 0; j < ncols; j++)  // column by column through the scratch vector, back in the caller's row order
            {
                launch_from_stored_order(*fac->A, fac->V.p + int64_t(j) * fac->ldv, fac->pscratch.p);
                MISPEC_HIP(hipMemcpyAsync(V_host + int64_t(j) * fac->nloc, fac->pscratch.p, size_t(fac->nloc) * sizeof(double),
                                          hipMemcpyDeviceToHost, fac->ctx->stream));
                MISPEC_HIP(hipStreamSynchronize(fac->ctx->stream));
            }
            return;
        }
        if (ncols && fac->nloc)
            MISPEC_HIP(hipMemcpy2D(V_host, size_t(fac->nloc) * sizeof(double), fac->V.p, size_t(fac->ldv) * sizeof(double),
                                   size_t(fac->nloc) * sizeof(double), size_t(ncols), hipMemcpyDeviceToHost));
    });
}

extern "C" int mispec_fac_get_f(const mispec_fac* fac, double* f_host)
{
    return guarded([&] {
        MISPEC_REQUIRE(fac && f_host, "mispec_fac_get_f: NULL argument");
        fac->ctx->make_current();
        finish_pending(*const_cast<mispec_fac*>(fac));  // one-sweep steps: the last correction may still be owed
        MISPEC_HIP(hipStreamSynchronize(fac->ctx->stream));
        if (fac->perm_mode)
        {
            launch_from_stored_order(*fac->A, fac->f.p, fac->pscratch.p);
            MISPEC_HIP(hipMemcpyAsync(f_host, fac->pscratch.p, size_t(fac->nloc) * sizeof(double), hipMemcpyDeviceToHost, fac->ctx->stream));
            MISPEC_HIP(hipStreamSynchronize(fac->ctx->stream));
            return;
        }
        if (fac->nloc)
            MISPEC_HIP(hipMemcpy(f_host, fac->f.p, size_t(fac->nloc) * sizeof(double), hipMemcpyDeviceToHost));
    });
}

extern "C" const double* mispec_fac_V_dev(const mispec_fac* fac, int64_t* ld)
{
    if (!fac)
        return nullptr;
    if (ld)
        *ld = fac->ldv;
    return fac->V.p;
}

extern "C" int mispec_fac_tridiag_eigen(mispec_fac* fac, double* evals_host, double* evecs_host)
{
    return guarded([&] {
        require_init(fac, "mispec_fac_tridiag_eigen");
        MISPEC_REQUIRE(evals_host, "mispec_fac_tridiag_eigen: evals is NULL");
        mispec_fac& F = *fac;
        MISPEC_REQUIRE(F.symmetric, "mispec_fac_tridiag_eigen: symmetric (Lanczos) factorisations only");
        F.ctx->make_current();
        const int m = F.m;
        double* hs = F.h_small.p;  // [diag m][subd m]
        for (int i = 0; i < m; i++)
            hs[i] = F.Hat(i, i);
        for (int i = 0; i < m; i++)
            hs[m + i] = (i < m - 1) ? F.Hat(i + 1, i) : 0.0;
        // H is on the host at this point (one read-back per factorisation sweep) and the Ritz pairs go back to the
        // host for the convergence test, so by default the m x m eigen-decomposition — a serial chain of rotations,
        // ~25 us on a host core against ~0.6 ms on one wavefront — runs where the data is.  MISPEC_SMALL=device
        // keeps it on the GPU (k_tridiag_eigen_w64 / k_tridiag_eigen; same routine, internal/SmallDense.h).
        const bool on_device_env = option_is("small", "device");
        const bool on_device = on_device_env && m <= kMaxSmallDim;
        if (!on_device)
        {
            F.counts[FAM_SMALL]++;
            std::vector<double> Q(evecs_host ? 0 : size_t(m) * m);
            double* q = evecs_host ? evecs_host : Q.data();
            std::fill(q, q + size_t(m) * m, 0.0);
            for (int i = 0; i < m; i++)
                q[size_t(i) * m + i] = 1.0;
            const int rc = small::tridiag_eigen(m, hs, hs + m, q, m, small::Lanes{0, 1});
            if (rc != 0)
                throw Error(MISPEC_ERUNTIME, "TridiagEigen: eigen decomposition failed");  // TridiagEigen.h:204
            std::copy(hs, hs + m, evals_host);
            return;
        }
        MISPEC_HIP(hipMemcpyAsync(F.d_diag.p, hs, size_t(m) * 8, hipMemcpyHostToDevice, F.stream()));
        MISPEC_HIP(hipMemcpyAsync(F.d_subd.p, hs + m, size_t(m) * 8, hipMemcpyHostToDevice, F.stream()));
        {
            Timed t(F, FAM_SMALL);
            launch_tridiag_eigen(*F.ctx, m, F.d_diag.p, F.d_subd.p, F.d_evals.p, F.d_evecs.p, F.d_info.p);
        }
        int info = 0;
        MISPEC_HIP(hipMemcpyAsync(evals_host, F.d_evals.p, size_t(m) * 8, hipMemcpyDeviceToHost, F.stream()));
        if (evecs_host)
            MISPEC_HIP(hipMemcpyAsync(evecs_host, F.d_evecs.p, size_t(m) * m * 8, hipMemcpyDeviceToHost, F.stream()));
        MISPEC_HIP(hipMemcpyAsync(&info, F.d_info.p, sizeof(int), hipMemcpyDeviceToHost, F.stream()));
        sync_stream(F);
        if (info != 0)
            throw Error(MISPEC_ERUNTIME, "TridiagEigen: eigen decomposition failed");  // TridiagEigen.h:204
    });
}

// Ritz values and the LAST ROW of the eigenvector matrix of H only — what the convergence test of an iteration needs
// (HermEigsBase.h:158-175: |last component| * |f|); the full matrix costs O(m) per rotation instead of O(1) and is needed once,
// after the last iteration.  Same routine, same rotations: tridiag_eigen updates the rows its `lanes` name, here row m - 1 alone,
// so the values and that row are bit-identical to mispec_fac_tridiag_eigen's (62 -> ~12 us per restart at m = 40 on a host core).
extern "C" int mispec_fac_ritz_values(mispec_fac* fac, double* evals_host, double* last_row_host)
{
    return guarded([&] {
        require_init(fac, "mispec_fac_ritz_values");
        MISPEC_REQUIRE(evals_host && last_row_host, "mispec_fac_ritz_values: NULL argument");
        mispec_fac& F = *fac;
        MISPEC_REQUIRE(F.symmetric, "mispec_fac_ritz_values: symmetric (Lanczos) factorisations only");
        const int m = F.m;
        const bool on_device_env = option_is("small", "device");
        if (on_device_env && m <= kMaxSmallDim)  // the device kernel forms the whole matrix: take its last row
        {
            std::vector<double> U(size_t(m) * m);
            if (mispec_fac_tridiag_eigen(fac, evals_host, U.data()) != MISPEC_OK)
                throw Error(MISPEC_ERUNTIME, mispec_last_error());
            for (int j = 0; j < m; j++)
                last_row_host[j] = U[size_t(j) * m + (m - 1)];
            return;
        }
        F.ctx->make_current();
        double* hs = F.h_small.p;  // [diag m][subd m][Q m*m]
        for (int i = 0; i < m; i++)
            hs[i] = F.Hat(i, i);
        for (int i = 0; i < m; i++)
            hs[m + i] = (i < m - 1) ? F.Hat(i + 1, i) : 0.0;
        F.counts[FAM_SMALL]++;
        double* q = hs + 2 * m;
        std::fill(q, q + size_t(m) * m, 0.0);
        for (int i = 0; i < m; i++)
            q[size_t(i) * m + i] = 1.0;
        const int rc = small::tridiag_eigen(m, hs, hs + m, q, m, small::Lanes{m - 1, 1});
        if (rc != 0)
            throw Error(MISPEC_ERUNTIME, "TridiagEigen: eigen decomposition failed");  // TridiagEigen.h:204
        std::copy(hs, hs + m, evals_host);
        for (int j = 0; j < m; j++)
            last_row_host[j] = q[size_t(j) * m + (m - 1)];
    });
}

namespace {

// One-sweep steps: the restart went ahead on a residual whose single correction left max |V'f| above eps |f| (Lanczos.h:156, the
// case in which the reference's loop corrects once more).  The loop continues on the COMPRESSED factorisation: V[:, :k]'f is
// measured again and, while the test still fails, f -= V c with c[k-2], c[k-1] absorbed by H(k-2 : k-1, k-1) (Lanczos.h:171-180
// for the residual of a k-step factorisation; count continues at 1).  `force`: apply one correction in any case (test hook).
void corrections_after_fused_restart(mispec_fac& F, bool force)
{
    const int k = F.k;
    const double beta_thresh = kEps * std::sqrt(double(F.n));
    vtf(F, F.f.p, k, 0);
    F.beta = F.h_red.p[kSlotBeta];
    double ortho_err = F.h_red.p[kSlotErr];
    int count = 1;
    while (count < 5 && (ortho_err > kEps * F.beta || force))
    {
        force = false;
        if (F.beta < beta_thresh)
        {
            zero_vector(F, F.f.p);
            F.beta = 0.0;
            break;
        }
        const double c_km2 = k >= 2 ? F.h_red.p[k - 2] : 0.0, c_km1 = F.h_red.p[k - 1];
        correct_vtf(F, F.f.p, F.f.p, k);
        if (k >= 2)
        {
            F.Hat(k - 2, k - 1) += c_km2;
            F.Hat(k - 1, k - 2) = F.Hat(k - 2, k - 1);
        }
        F.Hat(k - 1, k - 1) += c_km1;
        F.beta = F.h_red.p[kSlotBeta];
        ortho_err = F.h_red.p[kSlotErr];
        count++;
    }
}

}  // namespace

extern "C" int mispec_fac_restart_sym(mispec_fac* fac, const double* shifts_host, int nshift)
{
    return guarded([&] {
        require_init(fac, "mispec_fac_restart_sym");
        mispec_fac& F = *fac;
        MISPEC_REQUIRE(F.symmetric, "mispec_fac_restart_sym: symmetric (Lanczos) factorisations only");
        MISPEC_REQUIRE(shifts_host && nshift >= 1 && nshift < F.m, "mispec_fac_restart_sym: need 1 <= nshift < ncv");
        MISPEC_REQUIRE(F.k == F.m, "mispec_fac_restart_sym: the factorisation must be complete (k == ncv)");
        F.ctx->make_current();
        const int m = F.m;
        const int k = m - nshift;  // compress_H decrements k once per shift (Lanczos.h:198-202)
        double* hs = F.h_small.p;  // [diag m][subd m][Q m*m]
        for (int i = 0; i < m; i++)
            hs[i] = F.Hat(i, i);
        for (int i = 0; i < m; i++)
            hs[m + i] = (i < m - 1) ? F.Hat(i + 1, i) : 0.0;
        // Where the (m-k) shifted QR sweeps run: on the host core by default (same routine as the kernel, internal/SmallDense.h:
        // ~40 us + a 12.8 KB upload of Q, against 0.24 ms for the one-wavefront kernel k_restart_sym* — a serial chain the GPU
        // cannot speed up and that every rank of a sharded run would repeat).  Measured on C2, one GPU, end of round 3: 18.35 ->
        // 18.72 eigenpairs/s (profiles/r05m_*; -2.3 % already in round 2, when the default was still the device).
        // MISPEC_SMALL=device keeps the sweeps on the GPU (m <= 128; tested in both settings).
        const bool on_host = m > kMaxSmallDim || !small_on_device();
        if (on_host)
        {
            F.counts[FAM_SMALL]++;
            double* Q = hs + 2 * m;
            // The sweeps as a skewed pipeline (internal/SmallDensePipelined.h): the serial order's T and Q bit for bit (the sweeps in
            // flight overlap on the out-of-order core, the rows of Q go through SIMD registers) in 40 % of its time — ~20 instead of
            // ~50 us at m = 40, 18 shifts, on the critical path of every restart.  Option small=host-serial: the reference's order.
            if (option_is("small", "host-serial"))
            {
                std::fill(Q, Q + size_t(m) * m, 0.0);
                for (int i = 0; i < m; i++)
                    Q[size_t(i) * m + i] = 1.0;
                std::vector<double> work(size_t(4) * m);
                for (int sft = 0; sft < nshift; sft++)
                    small::tridiag_shifted_qr(m, hs, hs + m, shifts_host[sft], Q, m, m, work.data(), small::Lanes{0, 1});
            }
            else
            {
                const int ld = (m + 7) / 8 * 8;
                F.sweep_work.resize(size_t(2 * nshift + 2) * m);
                F.sweep_q.assign(size_t(ld) * m, 0.0);
                F.sweep_lanes.resize(size_t(nshift));
                for (int i = 0; i < m; i++)
                    F.sweep_q[size_t(i) * ld + i] = 1.0;
                small::restart_rotations_pipelined(m, hs, hs + m, shifts_host, nshift, F.sweep_work.data(), F.sweep_lanes.data());
                small::apply_sweeps_to_Q(F.sweep_q.data(), ld, ld, m, F.sweep_work.data(), F.sweep_work.data() + size_t(nshift) * m,
                                         nshift);
                for (int c = 0; c < m; c++)
                    std::memcpy(Q + size_t(c) * m, F.sweep_q.data() + size_t(c) * ld, size_t(m) * sizeof(double));
            }
            const double q_last = Q[size_t(k - 1) * m + (m - 1)], h_sub = hs[m + k - 1];  // Q(m-1, k-1), the new H(k, k-1)
            const bool fused = F.end_pending;
            // Without a host turn (the default): the record's scalar tail runs on the device (kFinishFusedRestart) and leaves the
            // start state of the next sweep in d_state — H after compress_H is known here, beta = |f_new| follows from the
            // record —, so factorize_lanczos enqueues that sweep at once; should the corrected residual fail the reference's
            // test (Lanczos.h:156), none of its steps runs and the host continues the reference's loop at the sweep's end.
            const bool no_sync = fused && !option_is("restart_sync", "1") && !F.test_recorrect && F.device_steps && device_operator(F) &&
                                 !F.bmode();
            const bool fast = fast_host_turn();
            if (fast)
            {
                // Q and (no_sync) the next sweep's start state through ONE small kernel reading pinned host memory
                double* up = F.h_up.p;
                std::memcpy(up, Q, size_t(m) * m * 8);
                for (int j = 0; j < m; j++)
                {
                    up[size_t(m) * m + j] = hs[j];
                    up[size_t(m) * m + m + j] = (j + 1 < m) ? hs[m + j] : 0.0;
                }
                launch_fetch_restart(*F.ctx, up, m, F.Qdev.p, F.d_state.p, no_sync ? 1 : 0);
            }
            else
                MISPEC_HIP(hipMemcpyAsync(F.Qdev.p, Q, size_t(m) * m * 8, hipMemcpyHostToDevice, F.stream()));
            bool test_failed = false;
            if (fused)
            {
                // One-sweep steps: the last step's correction f = ftilde - V c and the reference's test of the result (Lanczos.h:
                // 156: max |V'f| <= eps |f|) ride on the V*Q pass (k_vq_fused) — one sweep over the basis instead of two.
                F.end_pending = false;
                VqFusedArgs fa;
                fa.c = F.red_buf(F.end_rec);
                fa.ftilde = F.f.p;
                fa.fnew = F.tmp.p;
                fa.q_last = q_last;
                fa.h_sub = h_sub;
                fa.kcol = k;
                fa.partials = F.partials.p;
                fa.pstride = F.pstride;
                if (no_sync && !fast)
                {
                    StepState& st0 = *F.h_state.p;
                    std::memset(&st0, 0, sizeof(StepState));
                    st0.status = kStepOk;
                    for (int j = 0; j < m; j++)
                    {
                        st0.diag[j] = hs[j];
                        st0.subd[j] = (j + 1 < m) ? hs[m + j] : 0.0;
                    }
                    MISPEC_HIP(hipMemcpyAsync(F.d_state.p, &st0, sizeof(StepState), hipMemcpyHostToDevice, F.stream()));
                }
                int nrec;
                {
                    Timed t(F, FAM_COMPRESS);
                    F.count_bytes(FAM_COMPRESS, m + k + 1 + 2);  // m columns and ftilde read, k + 1 columns and the new f written
                    nrec = launch_vq_fused(*F.ctx, F.V.p, F.ldv, m, F.Qdev.p, m, k + 1, F.V.p, F.ldv, F.nloc, fa);
                }
                if (no_sync)
                {
                    FinishArgs fin;
                    fin.mode = kFinishFusedRestart;
                    fin.st = F.d_state.p;
                    fin.step = k;
                    fin.eps = F.test_restart_check ? -1.0 : kEps;  // (the hook: max |V'f| > -|f| always holds)
                    reduce_record(F, nrec, m + 1, F.end_rec ^ 1, fin);
                    if (F.turn_open)
                    {
                        F.turn_host_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - F.turn_t0).count();
                        F.turn_count++;
                        F.turn_open = false;
                    }
                    F.fused_restarts++;
                    F.f.swap(F.tmp);
                    F.beta = std::numeric_limits<double>::quiet_NaN();  // unknown on the host until the state comes back
                    std::fill(F.H.begin(), F.H.end(), 0.0);
                    for (int i = 0; i < m; i++)
                        F.Hat(i, i) = hs[i];
                    for (int i = 0; i < m - 1; i++)
                        F.Hat(i + 1, i) = F.Hat(i, i + 1) = hs[m + i];
                    F.k = k;
                    F.restart_unresolved = true;
                    return;
                }
                reduce_to_host(F, nrec, m + 1, F.end_rec ^ 1);  // slots [0, m) V'f, m |f_new|^2, kSlotBeta2 |f|^2
                double err = 0.0;
                for (int j = 0; j < m; j++)
                    err = std::max(err, std::fabs(F.h_red.p[j]));
                const double beta_corr = F.h_red.p[kSlotBeta];
                F.lag_chk_max = std::max(F.lag_chk_max, beta_corr > 0.0 ? err / beta_corr : 0.0);
                test_failed = err > kEps * beta_corr;  // Lanczos.h:156 with count = 1
                if (!F.test_recorrect && err > 0.5 * kEps * beta_corr)
                    F.eager_sticky = true;  // the Ritz values of this restart were computed before the correction: do not repeat that
                F.fused_restarts++;
                F.f.swap(F.tmp);
                F.beta = std::sqrt(F.h_red.p[m]);  // Arnoldi.h:339
            }
            else
            {
                Timed t(F, FAM_COMPRESS);
                compress_basis(F, k + 1);
            }
            std::fill(F.H.begin(), F.H.end(), 0.0);
            for (int i = 0; i < m; i++)
                F.Hat(i, i) = hs[i];
            for (int i = 0; i < m - 1; i++)
                F.Hat(i + 1, i) = F.Hat(i, i + 1) = hs[m + i];
            F.k = k;
            if (!fused)
                update_f_after_compress(F, q_last, h_sub);  // syncs: Q has been consumed by then
            else if (test_failed || F.test_recorrect)
            {
                F.fused_recorrected++;
                corrections_after_fused_restart(F, F.test_recorrect);
            }
            return;
        }
        finish_pending(F);
        MISPEC_HIP(hipMemcpyAsync(F.d_diag.p, hs, size_t(m) * 8, hipMemcpyHostToDevice, F.stream()));
        MISPEC_HIP(hipMemcpyAsync(F.d_subd.p, hs + m, size_t(m) * 8, hipMemcpyHostToDevice, F.stream()));
        {
            Timed t(F, FAM_SMALL);
            launch_restart_sym(*F.ctx, m, F.d_diag.p, F.d_subd.p, shifts_host, nshift, F.Qdev.p);
        }
        // V[:, :k+1] <- V Q  (Arnoldi.h:326-335), in place, straight from the device Q
        {
            Timed t(F, FAM_COMPRESS);
            compress_basis(F, k + 1);
        }
        MISPEC_HIP(hipMemcpyAsync(hs, F.d_diag.p, size_t(m) * 8, hipMemcpyDeviceToHost, F.stream()));
        MISPEC_HIP(hipMemcpyAsync(hs + m, F.d_subd.p, size_t(m) * 8, hipMemcpyDeviceToHost, F.stream()));
        MISPEC_HIP(hipMemcpyAsync(hs + 2 * m, F.Qdev.p + size_t(k - 1) * m + (m - 1), sizeof(double), hipMemcpyDeviceToHost,
                                  F.stream()));
        sync_stream(F);
        std::fill(F.H.begin(), F.H.end(), 0.0);
        for (int i = 0; i < m; i++)
            F.Hat(i, i) = hs[i];
        for (int i = 0; i < m - 1; i++)
            F.Hat(i + 1, i) = F.Hat(i, i + 1) = hs[m + i];
        F.k = k;
        update_f_after_compress(F, hs[2 * m], F.Hat(k, k - 1));
    });
}

extern "C" int mispec_fac_compress_V(mispec_fac* fac, const double* Q_host, const double* H_host, int new_k)
{
    return guarded([&] {
        require_init(fac, "mispec_fac_compress_V");
        mispec_fac& F = *fac;
        MISPEC_REQUIRE(Q_host && H_host && new_k >= 1 && new_k < F.m, "mispec_fac_compress_V: bad argument");
        F.ctx->make_current();
        finish_pending(F);
        const int m = F.m;
        std::memcpy(F.h_small.p, Q_host, size_t(m) * m * sizeof(double));
        MISPEC_HIP(hipMemcpyAsync(F.Qdev.p, F.h_small.p, size_t(m) * m * 8, hipMemcpyHostToDevice, F.stream()));
        std::memcpy(F.H.data(), H_host, F.H.size() * sizeof(double));
        F.k = new_k;
        {
            Timed t(F, FAM_COMPRESS);
            compress_basis(F, new_k + 1);
        }
        update_f_after_compress(F, Q_host[size_t(new_k - 1) * m + (m - 1)], F.Hat(new_k, new_k - 1));
    });
}

// The whole shift list of one general (Arnoldi) restart on the device: H and Q stay there, Q is consumed by V <- V Q
// straight from HBM; only H comes back (it is authoritative on the host).  GenEigsBase.h:204-222 / RestartArnoldi.
extern "C" int mispec_fac_restart_gen(mispec_fac* fac, const int* kind, const double* a, const double* b, int nshift, int new_k)
{
    return guarded([&] {
        require_init(fac, "mispec_fac_restart_gen");
        mispec_fac& F = *fac;
        MISPEC_REQUIRE(!F.symmetric, "mispec_fac_restart_gen: general (Arnoldi) factorisations only");
        MISPEC_REQUIRE(kind && a && b && nshift >= 1 && nshift <= kMaxShifts && new_k >= 1 && new_k < F.m, "mispec_fac_restart_gen: bad argument");
        MISPEC_REQUIRE(F.m <= kMaxGenDim, "mispec_fac_restart_gen: the device kernel holds at most 96 basis columns");
        MISPEC_REQUIRE(F.k == F.m, "mispec_fac_restart_gen: the factorisation must be complete (k == ncv)");
        F.ctx->make_current();
        const int m = F.m;
        GenShiftList sl;
        sl.count = nshift;
        for (int i = 0; i < nshift; i++)
        {
            MISPEC_REQUIRE(kind[i] == 0 || kind[i] == 1, "mispec_fac_restart_gen: kind must be 0 (real shift) or 1 (double shift)");
            sl.kind[i] = kind[i];
            sl.a[i] = a[i];
            sl.b[i] = b[i];
        }
        std::memcpy(F.h_H.p, F.H.data(), size_t(m) * m * sizeof(double));
        MISPEC_HIP(hipMemcpyAsync(F.d_H.p, F.h_H.p, size_t(m) * m * 8, hipMemcpyHostToDevice, F.stream()));
        {
            Timed t(F, FAM_SMALL);
            launch_restart_gen(*F.ctx, m, F.d_H.p, sl, F.Qdev.p);
        }
        F.k = new_k;
        {
            Timed t(F, FAM_COMPRESS);
            compress_basis(F, new_k + 1);
        }
        double* hs = F.h_small.p;
        MISPEC_HIP(hipMemcpyAsync(F.h_H.p, F.d_H.p, size_t(m) * m * 8, hipMemcpyDeviceToHost, F.stream()));
        MISPEC_HIP(hipMemcpyAsync(hs, F.Qdev.p + size_t(new_k - 1) * m + (m - 1), sizeof(double), hipMemcpyDeviceToHost, F.stream()));
        sync_stream(F);
        std::memcpy(F.H.data(), F.h_H.p, size_t(m) * m * sizeof(double));
        update_f_after_compress(F, hs[0], F.Hat(new_k, new_k - 1));
    });
}

extern "C" int mispec_fac_ritz_vectors(mispec_fac* fac, const double* Y_host, int ncols, double* X_host, const double** X_dev)
{
    return guarded([&] {
        require_init(fac, "mispec_fac_ritz_vectors");
        mispec_fac& F = *fac;
        MISPEC_REQUIRE(Y_host && ncols >= 1 && ncols <= F.m, "mispec_fac_ritz_vectors: bad argument");
        F.ctx->make_current();
        const int m = F.m;
        if (F.X.n < size_t(F.ldv) * size_t(ncols))
            F.X.alloc(size_t(F.ldv) * size_t(ncols));
        MISPEC_HIP(hipMemsetAsync(F.X.p, 0, size_t(F.ldv) * size_t(ncols) * sizeof(double), F.stream()));
        std::memcpy(F.h_small.p, Y_host, size_t(m) * size_t(ncols) * sizeof(double));
        MISPEC_HIP(hipMemcpyAsync(F.d_Y.p, F.h_small.p, size_t(m) * size_t(ncols) * 8, hipMemcpyHostToDevice, F.stream()));
        {
            Timed t(F, FAM_COMPRESS);
            F.count_bytes(FAM_COMPRESS, m + ncols);
            launch_vq(*F.ctx, F.V.p, F.ldv, m, F.d_Y.p, m, ncols, F.X.p, F.ldv, F.nloc);  // HermEigsBase.h:467
        }
        F.x_cols = ncols;
        F.x_original = F.perm_mode;
        if (F.perm_mode)
            for (int j = 0; j < ncols; j++)
                from_stored_order(F, F.X.p + int64_t(j) * F.ldv);  // rows back in the caller's order
        if (X_host && F.nloc)
            download_columns(F, F.X.p, F.ldv, F.nloc, ncols, X_host, F.nloc);
        sync_stream(F);
        if (X_dev)
            *X_dev = F.X.p;
    });
}

extern "C" int mispec_fac_residuals(mispec_fac* fac, const double* lambda_host, int ncols, double* resid_host)
{
    return guarded([&] {
        require_init(fac, "mispec_fac_residuals");
        mispec_fac& F = *fac;
        MISPEC_REQUIRE(lambda_host && resid_host && ncols >= 0 && ncols <= F.x_cols,
                       "mispec_fac_residuals: call mispec_fac_ritz_vectors first");
        MISPEC_REQUIRE(F.A, "mispec_fac_residuals: needs a device-resident matrix");
        F.ctx->make_current();
        for (int j = 0; j < ncols; j++)
        {
            const double* x = F.X.p + int64_t(j) * F.ldv;
            if (F.Bop)  // generalized problem: || A x - lambda B x || / || B x ||
            {
                launch_spmv(*F.A, x, F.tmp.p, nullptr);
                b_apply(F, x);
                x = F.bx.p;
            }
            else if (F.perm_mode && F.x_original)
                launch_spmv(*F.A, x, F.tmp.p, nullptr);  // X is in the caller's order: the order-preserving product
            else
                apply_op(F, x, F.tmp.p, false, nullptr, 0.0);
            const int nrec = launch_resid_norms(*F.ctx, F.tmp.p, x, lambda_host[j], F.nloc, F.partials.p, F.pstride);
            reduce_to_host(F, nrec, 1, 0);
            resid_host[j] = std::sqrt(F.h_red.p[kSlotBeta2]) / std::sqrt(F.h_red.p[0]);
        }
    });
}

extern "C" int mispec_fac_residuals_complex(mispec_fac* fac, const double* Yre_host, const double* Yim_host,
                                           const double* lambda_host, int ncols, double* resid_host)
{
    return guarded([&] {
        require_init(fac, "mispec_fac_residuals_complex");
        mispec_fac& F = *fac;
        MISPEC_REQUIRE(Yre_host && Yim_host && lambda_host && resid_host && ncols >= 0 && ncols <= F.m,
                       "mispec_fac_residuals_complex: bad argument");
        MISPEC_REQUIRE(F.A, "mispec_fac_residuals_complex: needs a device-resident matrix");
        F.ctx->make_current();
        const int m = F.m;
        if (F.X.n < size_t(F.ldv) * 2)
            F.X.alloc(size_t(F.ldv) * 2);
        F.x_cols = 0;
        double* xr = F.X.p;
        double* xi = F.X.p + F.ldv;
        for (int j = 0; j < ncols; j++)
        {
            // [x_r | x_i] = V * [Re y_j | Im y_j]
            std::memcpy(F.h_small.p, Yre_host + size_t(j) * m, size_t(m) * sizeof(double));
            std::memcpy(F.h_small.p + m, Yim_host + size_t(j) * m, size_t(m) * sizeof(double));
            MISPEC_HIP(hipMemcpyAsync(F.d_Y.p, F.h_small.p, size_t(2 * m) * 8, hipMemcpyHostToDevice, F.stream()));
            MISPEC_HIP(hipMemsetAsync(F.X.p, 0, size_t(F.ldv) * 2 * sizeof(double), F.stream()));
            launch_vq(*F.ctx, F.V.p, F.ldv, m, F.d_Y.p, m, 2, F.X.p, F.ldv, F.nloc);
            apply_op(F, xr, F.tmp.p, false, nullptr, 0.0);
            apply_op(F, xi, F.w.p, false, nullptr, 0.0);
            const int nrec = launch_resid_norms_complex(*F.ctx, F.tmp.p, F.w.p, xr, xi, lambda_host[2 * j], lambda_host[2 * j + 1],
                                                        F.nloc, F.partials.p, F.pstride);
            reduce_to_host(F, nrec, 1, 0);
            resid_host[j] = std::sqrt(F.h_red.p[kSlotBeta2]) / std::sqrt(F.h_red.p[0]);
        }
    });
}

extern "C" int mispec_fac_profile(mispec_fac* fac, int enable)
{
    return guarded([&] {
        MISPEC_REQUIRE(fac, "mispec_fac_profile: NULL argument");
        fac->ctx->make_current();
        if (!enable && fac->prof)
            drain_profile(*fac);
        fac->prof = (enable >= 2 && enable <= 4) ? enable : (enable != 0 ? 1 : 0);
    });
}

extern "C" int mispec_fac_get_profile(const mispec_fac* fac_c, mispec_profile* out)
{
    return guarded([&] {
        MISPEC_REQUIRE(fac_c && out, "mispec_fac_get_profile: NULL argument");
        mispec_fac& F = *const_cast<mispec_fac*>(fac_c);
        F.ctx->make_current();
        drain_profile(F);
        out->n_spmv = F.counts[FAM_SPMV];
        out->n_vtf = F.counts[FAM_VTF];
        out->n_gemv = F.counts[FAM_GEMV];
        out->n_scale = F.counts[FAM_SCALE];
        out->n_compress = F.counts[FAM_COMPRESS];
        out->n_small = F.counts[FAM_SMALL];
        out->n_host_sync = F.n_sync;
        out->ms_spmv = F.ms_acc[FAM_SPMV];
        out->ms_vtf = F.ms_acc[FAM_VTF];
        out->ms_gemv = F.ms_acc[FAM_GEMV];
        out->ms_scale = F.ms_acc[FAM_SCALE];
        out->ms_compress = F.ms_acc[FAM_COMPRESS];
        out->ms_small = F.ms_acc[FAM_SMALL];
        out->spmv_bytes = (F.A ? F.A->algorithmic_bytes() : 0.0) + (F.A2 ? F.A2->algorithmic_bytes() : 0.0) +
                          (F.D ? F.D->algorithmic_bytes() : 0.0);
        out->bytes_vtf = F.bytes_acc[FAM_VTF];
        out->bytes_gemv = F.bytes_acc[FAM_GEMV];
        out->bytes_compress = F.bytes_acc[FAM_COMPRESS];
        out->n_reduce = F.counts[FAM_REDUCE];
        out->n_exchange = F.counts[FAM_EXCH];
        out->n_exchange_wait = F.counts[FAM_XWAIT];
        out->n_allreduce = F.counts[FAM_ALLRED];
        out->ms_reduce = F.ms_acc[FAM_REDUCE];
        out->ms_exchange = F.ms_acc[FAM_EXCH];
        out->ms_exchange_wait = F.ms_acc[FAM_XWAIT];
        out->ms_allreduce = F.ms_acc[FAM_ALLRED];
    });
}
