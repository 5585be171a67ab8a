// =============================================================================
//  TEST INFRASTRUCTURE — C interface (ctypes-friendly) over spectra_oracle.hpp.
//  Built by oracle/Makefile into oracle/liboracle.so.  See the header of
//  spectra_oracle.hpp for scope and pinning status.
// =============================================================================
#include "spectra_oracle.hpp"
#include "spectra_oracle_gen.hpp"
#include "synth_matrix.h"
#include "onesweep_variant.hpp"

#include <chrono>
#include <cstring>
#include <random>

using namespace oracle;

namespace {
thread_local std::string g_err;
template <typename F>
int guarded(F&& f)
{
    try
    {
        f();
        return 0;
    }
    catch (const std::invalid_argument& e)
    {
        g_err = e.what();
        return -1;
    }
    catch (const std::logic_error& e)
    {
        g_err = e.what();
        return -2;
    }
    catch (const std::exception& e)
    {
        g_err = e.what();
        return -3;
    }
}
}  // namespace

extern "C" {

const char* oracle_last_error() { return g_err.c_str(); }

// ---- Util/SimpleRandom.h ----------------------------------------------------
void oracle_simple_random(unsigned long seed, long n, double* out)
{
    SimpleRandom rng(seed);
    rng.fill(out, n);
}
void oracle_lcg_states(long seed, long count, long* out)
{
    long x = seed;
    for (long i = 0; i < count; i++)
    {
        x = lcg_next(x);
        out[i] = x;
    }
}
// the same LCG through libstdc++ (std::minstd_rand0), as a cross-check of lcg_next
void oracle_minstd_states(long seed, long count, long* out)
{
    std::minstd_rand0 g(static_cast<unsigned long>(seed));
    for (long i = 0; i < count; i++)
        out[i] = static_cast<long>(g());
}

// ---- test/SymEigs.cpp:25-42 gen_sparse_data (libstdc++ RNG, bit-reproducible) --
// COO triplets in insertion (row-major) order. Call with rows==NULL to count.
long oracle_gen_sparse_data(int n, double prob, int* rows, int* cols, double* vals)
{
    std::default_random_engine gen;
    gen.seed(0);
    std::uniform_real_distribution<double> distr(0.0, 1.0);
    long cnt = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            if (distr(gen) < prob)
            {
                const double v = distr(gen) - 0.5;
                if (rows)
                {
                    rows[cnt] = i;
                    cols[cnt] = j;
                    vals[cnt] = v;
                }
                cnt++;
            }
    return cnt;
}

// ---- test/SVD.cpp:17-33 gen_sparse_data(m, n, prob): the rectangular fixture of the partial SVD tests --
long oracle_gen_sparse_data_rect(int m, int n, double prob, int* rows, int* cols, double* vals)
{
    std::default_random_engine gen;
    gen.seed(0);
    std::uniform_real_distribution<double> distr(0.0, 1.0);
    long cnt = 0;
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++)
            if (distr(gen) < prob)
            {
                const double v = distr(gen) - 0.5;
                if (rows)
                {
                    rows[cnt] = i;
                    cols[cnt] = j;
                    vals[cnt] = v;
                }
                cnt++;
            }
    return cnt;
}

// ---- test/DavidsonSymEigs.cpp:46-67 gen_sym_data_sparse(n): prob 0.5, entries 0.1*(u - 0.5), diagonal forced to i+1.
// (The matrix is not symmetric; SparseSymMatProd reads its lower triangle.)  COO, row-major order, diagonal included once.
long oracle_gen_davidson_sparse(int n, int* rows, int* cols, double* vals)
{
    const double prob = 0.5;
    std::default_random_engine gen;
    gen.seed(0);
    std::uniform_real_distribution<double> distr(0.0, 1.0);
    long cnt = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
        {
            bool have = false;
            double v = 0.0;
            if (distr(gen) < prob)
            {
                v = 0.1 * (distr(gen) - 0.5);
                have = true;
            }
            if (i == j)
            {
                v = i + 1;
                have = true;
            }
            if (have)
            {
                if (rows)
                {
                    rows[cnt] = i;
                    cols[cnt] = j;
                    vals[cnt] = v;
                }
                cnt++;
            }
        }
    return cnt;
}

// ---- synthetic benchmark matrices ------------------------------
// Row i holds columns i+off for every signed offset in {0} U {+-offsets[k]} that lands in [0,n),
// ascending.  rowptr may be NULL to just count.
long oracle_synth_band_csr(long n, unsigned long long seed, const long* offsets, int noff, int symmetric,
                           int* rowptr, int* colind, double* val)
{
    std::vector<long> offs;
    offs.push_back(0);
    for (int k = 0; k < noff; k++)
    {
        offs.push_back(offsets[k]);
        offs.push_back(-offsets[k]);
    }
    std::sort(offs.begin(), offs.end());
    offs.erase(std::unique(offs.begin(), offs.end()), offs.end());
    long cnt = 0;
    for (long i = 0; i < n; i++)
    {
        if (rowptr)
            rowptr[i] = static_cast<int>(cnt);
        for (long o : offs)
        {
            const long j = i + o;
            if (j < 0 || j >= n)
                continue;
            if (rowptr)
            {
                colind[cnt] = static_cast<int>(j);
                const uint64_t a = symmetric ? static_cast<uint64_t>(std::min(i, j)) : static_cast<uint64_t>(i);
                const uint64_t b = symmetric ? static_cast<uint64_t>(std::max(i, j)) : static_cast<uint64_t>(j);
                val[cnt] = synth_value(seed, a, b);
            }
            cnt++;
        }
    }
    if (rowptr)
        rowptr[n] = static_cast<int>(cnt);
    return cnt;
}
double oracle_synth_value(unsigned long long seed, unsigned long long a, unsigned long long b)
{
    return synth_value(seed, a, b);
}

// ---- LinAlg/Givens.h ----------------------------------------------------------
void oracle_givens(double x, double y, double* r, double* c, double* s) { givens_rotation(x, y, *r, *c, *s); }
void oracle_eigen_make_givens(double p, double q, double* c, double* s) { eigen_make_givens(p, q, *c, *s); }

// ---- LinAlg/UpperHessenbergQR.h TridiagQR --------------------------------------
// T: n x n column-major. Outputs (each may be NULL): R, QtHQ, Q (= I * G1 * G2 ...), all n x n col-major.
int oracle_tridiag_qr(long n, const double* T, double shift, double* R, double* QtHQ, double* Q)
{
    return guarded([&] {
        Mat M(n, n);
        std::memcpy(M.a.data(), T, sizeof(double) * n * n);
        TridiagQR qr;
        qr.compute(M, shift);
        if (R)
        {
            Mat r = qr.matrix_R();
            std::memcpy(R, r.a.data(), sizeof(double) * n * n);
        }
        if (QtHQ)
        {
            Mat d;
            qr.matrix_QtHQ(d);
            std::memcpy(QtHQ, d.a.data(), sizeof(double) * n * n);
        }
        if (Q)
        {
            Mat q(n, n);
            q.set_identity();
            qr.apply_YQ(q);
            std::memcpy(Q, q.a.data(), sizeof(double) * n * n);
        }
    });
}

// ---- LinAlg/TridiagEigen.h ------------------------------------------------------
int oracle_tridiag_eigen(long n, const double* T, double* evals, double* evecs)
{
    return guarded([&] {
        Mat M(n, n);
        std::memcpy(M.a.data(), T, sizeof(double) * n * n);
        TridiagEigen te;
        te.compute(M);
        std::memcpy(evals, te.main_diag.data(), sizeof(double) * n);
        if (evecs)
            std::memcpy(evecs, te.evecs.a.data(), sizeof(double) * n * n);
    });
}

// ---- Util/SelectionRule.h argsort --------------------------------------------------
int oracle_argsort(int rule, const double* values, long len, long* out)
{
    return guarded([&] {
        std::vector<Index> ind = argsort(static_cast<SortRule>(rule), values, len);
        for (long i = 0; i < len; i++)
            out[i] = ind[i];
    });
}

// ---- operators ---------------------------------------------------------------------
void* oracle_op_csc_sym(long n, const int* colptr, const int* rowind, const double* val, int lower)
{
    return new SparseSymCsc(n, colptr, rowind, val, lower != 0);
}
void* oracle_op_csr(long nr, long nc, const int* rowptr, const int* colind, const double* val)
{
    return new SparseCsr(nr, nc, rowptr, colind, val);
}
void* oracle_op_csc(long nr, long nc, const int* colptr, const int* rowind, const double* val)
{
    return new SparseCsc(nr, nc, colptr, rowind, val);
}
void* oracle_op_dense_sym(long n, const double* a) { return new DenseSym(n, a); }
void* oracle_op_dense_gen(long n, const double* a) { return new DenseGen(n, a); }
void* oracle_op_diag(long n, const double* d)
{
    std::vector<double> dv(d, d + n);
    return new CallbackOp(n, [dv, n](const double* x, double* y) {
        for (long i = 0; i < n; i++)
            y[i] = x[i] * dv[i];
    });
}
// operator given as a C callback (tests: a scipy factorisation playing the role of Eigen::SparseLU in
// MatOp/SparseSymShiftSolve.h:104-109)
typedef void (*oracle_op_cb)(const double* x, double* y);
void* oracle_op_callback(long n, oracle_op_cb cb)
{
    return new CallbackOp(n, [cb](const double* x, double* y) { cb(x, y); });
}
void oracle_op_free(void* op) { delete static_cast<Op*>(op); }
long oracle_op_rows(void* op) { return static_cast<Op*>(op)->rows(); }
void oracle_op_apply(void* op, const double* x, double* y) { static_cast<Op*>(op)->perform_op(x, y); }
// mean seconds per perform_op over `reps` applications (cpu_baseline helper)
double oracle_op_time(void* op, const double* x, double* y, int reps)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; i++)
        static_cast<Op*>(op)->perform_op(x, y);
    const auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count() / reps;
}

// ---- LinAlg/Arnoldi.h + Lanczos.h ------------------------------------------------------
struct FacHandle
{
    Factorization fac;
    bool symmetric;
    Index nmatop = 0;
    FacHandle(const Op& op, Index m, bool sym) : fac(op, m), symmetric(sym) {}
};
void* oracle_fac_create(void* op, long m, int symmetric)
{
    return new FacHandle(*static_cast<Op*>(op), m, symmetric != 0);
}
void oracle_fac_free(void* h) { delete static_cast<FacHandle*>(h); }
int oracle_fac_init(void* h, const double* v0)
{
    auto* F = static_cast<FacHandle*>(h);
    return guarded([&] { F->fac.init(v0, F->nmatop); });
}
int oracle_fac_factorize(void* h, long from_k, long to_m)
{
    auto* F = static_cast<FacHandle*>(h);
    return guarded([&] {
        if (F->symmetric)
            F->fac.factorize_from_lanczos(from_k, to_m, F->nmatop);
        else
            F->fac.factorize_from_arnoldi(from_k, to_m, F->nmatop);
    });
}
long oracle_fac_k(void* h) { return static_cast<FacHandle*>(h)->fac.k; }
long oracle_fac_nmatop(void* h) { return static_cast<FacHandle*>(h)->nmatop; }
double oracle_fac_beta(void* h) { return static_cast<FacHandle*>(h)->fac.beta; }
void oracle_fac_get(void* h, double* V, double* H, double* f)
{
    auto* F = static_cast<FacHandle*>(h);
    if (V)
        std::memcpy(V, F->fac.V.a.data(), sizeof(double) * F->fac.V.a.size());
    if (H)
        std::memcpy(H, F->fac.H.a.data(), sizeof(double) * F->fac.H.a.size());
    if (f)
        std::memcpy(f, F->fac.f.data(), sizeof(double) * F->fac.f.size());
}

// ---- HermEigsBase.h / SymEigsSolver.h / SymEigsShiftSolver.h -----------------------------
void* oracle_symeigs_create(void* op, long nev, long ncv)
{
    SymEigs* s = nullptr;
    int rc = guarded([&] { s = new SymEigs(*static_cast<Op*>(op), nev, ncv); });
    return rc == 0 ? s : nullptr;
}
void oracle_symeigs_free(void* s) { delete static_cast<SymEigs*>(s); }
// generalized problems with an explicit B-inner product (HermEigsBase<ModeMatOp, BOpType>): op is the Krylov operator,
// bop the operator y = B x; transform: 1 shift-invert, 2 buckling, 3 Cayley (SymGEigsShiftSolver.h), 0 none
void* oracle_symeigs_create_b(void* op, void* bop, long nev, long ncv, int transform, double sigma)
{
    SymEigs* s = nullptr;
    int rc = guarded([&] {
        s = new SymEigs(*static_cast<Op*>(op), nev, ncv, static_cast<Op*>(bop));
        s->sigma = sigma;
        s->shift_invert = (transform == 1);
        s->transform = transform;
    });
    return rc == 0 ? s : nullptr;
}

// ---- generalized problem, regular-inverse mode (SymGEigsSolver.h:224-238 + SparseRegularInverse.h) ----------
// A and B as CSC of which the lower triangle is used (the reference's defaults).  The returned holder owns
// everything; oracle_geigs_symeigs() gives the inner solver for the oracle_symeigs_* calls (do not free it).
struct GEigsHolder
{
    SparseSymCsc A, B;
    RegularInverse binv;
    RegInvOp op;
    SymEigs eigs;
    GEigsHolder(long n, const int* acp, const int* ari, const double* av, const int* bcp, const int* bri, const double* bv, long nev,
                long ncv) :
        A(n, acp, ari, av, true), B(n, bcp, bri, bv, true), binv(B), op(A, binv), eigs(op, nev, ncv, &B)
    {}
};
void* oracle_geigs_reginv_create(long n, const int* acp, const int* ari, const double* av, const int* bcp, const int* bri,
                                 const double* bv, long nev, long ncv)
{
    GEigsHolder* h = nullptr;
    int rc = guarded([&] { h = new GEigsHolder(n, acp, ari, av, bcp, bri, bv, nev, ncv); });
    return rc == 0 ? h : nullptr;
}
void oracle_geigs_free(void* h) { delete static_cast<GEigsHolder*>(h); }
void* oracle_geigs_symeigs(void* h) { return &static_cast<GEigsHolder*>(h)->eigs; }
// x = B^{-1} rhs by the restated ConjugateGradient; returns the iteration count, or -1 if it did not converge
long oracle_geigs_cg_solve(void* h, const double* rhs, double* x)
{
    auto* H = static_cast<GEigsHolder*>(h);
    const bool ok = H->binv.solve(rhs, x);
    return ok ? long(H->binv.last_iterations) : -1;
}
void oracle_geigs_bprod(void* h, const double* x, double* y) { static_cast<GEigsHolder*>(h)->B.perform_op(x, y); }
// NON-reference variant (oracle/onesweep_variant.hpp): the repository's opt-in one-sweep Lanczos factorisation under the
// reference's driver.  on == 0 restores the reference's algorithm.
// on: bit 0 the variant, bit 1 the last correction of a sweep rides on the restart (fused restart), bit 2 test hook: one more
// correction after every fused restart, bit 3 one reduction per step (product on the un-normalised residual)
void oracle_symeigs_set_onesweep(void* s, int on)
{
    auto* S = static_cast<SymEigs*>(s);
    if (!(on & 1))
    {
        S->fac.lanczos_variant = nullptr;
        S->fac.compress_variant = nullptr;
        S->fac.variant_user.reset();
        return;
    }
    auto st = std::make_shared<OneSweepStats>();
    st->defer_last = (on & 2) != 0;
    st->force_recorrect = (on & 4) != 0;
    st->one_reduction = (on & 8) != 0;
    S->fac.variant_user = st;
    S->fac.lanczos_variant = [](Factorization& F, Index from_k, Index to_m, Index& ops, void* user) {
        factorize_from_lanczos_onesweep(F, from_k, to_m, ops, static_cast<OneSweepStats*>(user));
    };
    S->fac.compress_variant = [](Factorization& F, const Mat& Q, void* user) { compress_onesweep(F, Q, static_cast<OneSweepStats*>(user)); };
}
// out[0..10): lagged steps, faithful steps, check fallbacks, state fallbacks, final passes, max |c|/|f~|, max |V'v| after a lagged
// correction, fused restarts, fused restarts followed by further corrections, steps taken with one reduction
void oracle_symeigs_onesweep_stats(void* s, double* out)
{
    auto* S = static_cast<SymEigs*>(s);
    const auto* st = static_cast<const OneSweepStats*>(S->fac.variant_user.get());
    for (int i = 0; i < 10; i++)
        out[i] = 0.0;
    if (!st)
        return;
    out[0] = double(st->lagged_steps);
    out[1] = double(st->faithful_steps);
    out[2] = double(st->fallbacks_check);
    out[3] = double(st->fallbacks_state);
    out[4] = double(st->final_passes);
    out[5] = st->max_rel_c;
    out[6] = st->max_chk;
    out[7] = double(st->fused_restarts);
    out[8] = double(st->fused_recorrected);
    out[9] = double(st->one_reduction_steps);
}
void oracle_symeigs_set_shift_invert(void* s, double sigma)
{
    static_cast<SymEigs*>(s)->shift_invert = true;
    static_cast<SymEigs*>(s)->sigma = sigma;
}
int oracle_symeigs_init(void* s, const double* v0)
{
    auto* S = static_cast<SymEigs*>(s);
    return guarded([&] {
        if (v0)
            S->init(v0);
        else
            S->init();
    });
}
// returns nconv (>= 0) or a negative error code
long oracle_symeigs_compute(void* s, int selection, long maxit, double tol, int sorting)
{
    auto* S = static_cast<SymEigs*>(s);
    long nconv = 0;
    int rc = guarded(
        [&] { nconv = S->compute(static_cast<SortRule>(selection), maxit, tol, static_cast<SortRule>(sorting)); });
    return rc == 0 ? nconv : rc;
}
int oracle_symeigs_info(void* s) { return static_cast<int>(static_cast<SymEigs*>(s)->info); }
long oracle_symeigs_num_iterations(void* s) { return static_cast<SymEigs*>(s)->niter; }
long oracle_symeigs_num_operations(void* s) { return static_cast<SymEigs*>(s)->nmatop; }
long oracle_symeigs_eigenvalues(void* s, double* out)
{
    std::vector<double> ev = static_cast<SymEigs*>(s)->eigenvalues();
    std::memcpy(out, ev.data(), sizeof(double) * ev.size());
    return static_cast<long>(ev.size());
}
long oracle_symeigs_eigenvectors(void* s, long nvec, double* out)
{
    Mat X = static_cast<SymEigs*>(s)->eigenvectors(nvec);
    std::memcpy(out, X.a.data(), sizeof(double) * X.a.size());
    return X.cols;
}
// Bounded timing sample for bench.py's cpu_baseline: init() + factorize_from(1, nsteps+1)
// on the given op; returns wall seconds and the number of perform_op calls made.
double oracle_symeigs_time_steps(void* op, long ncv, long nsteps, long* nops_out)
{
    Op& O = *static_cast<Op*>(op);
    const Index n = O.rows();
    std::vector<double> v0(n);
    SimpleRandom rng(0);
    rng.fill(v0.data(), n);
    Factorization fac(O, ncv);
    Index nops = 0;
    const auto t0 = std::chrono::steady_clock::now();
    fac.init(v0.data(), nops);
    fac.factorize_from_lanczos(1, std::min<Index>(ncv, nsteps + 1), nops);
    const auto t1 = std::chrono::steady_clock::now();
    *nops_out = nops;
    return std::chrono::duration<double>(t1 - t0).count();
}


// The sample SURVEY.md 8(d) prescribes for the CPU baseline: init() + factorize_from(1, ncv) [the first sweep], then `cycles`
// restart cycles of the IRLM driver (HermEigsBase.h:366-390: num_converged, nev_adjusted, restart = shifts + compress_V +
// factorize_from(k, ncv)).  out = {seconds of the first sweep, its perform_op count, seconds of the restart cycles,
// their perform_op count, number of cycles run (fewer if the solve converged first)}.
int oracle_symeigs_time_cycles(void* op, long nev, long ncv, int selection, double tol, long cycles, double* out)
{
    return guarded([&] {
        Op& O = *static_cast<Op*>(op);
        SymEigs s(O, nev, ncv);
        const auto t0 = std::chrono::steady_clock::now();
        s.init();
        s.fac.factorize_from_lanczos(1, ncv, s.nmatop);
        s.retrieve_ritzpair(static_cast<SortRule>(selection));
        const auto t1 = std::chrono::steady_clock::now();
        const Index ops_first = s.nmatop;
        long done = 0;
        for (; done < cycles; done++)
        {
            const Index nconv = s.num_converged(tol);
            if (nconv >= nev)
                break;
            s.restart(s.nev_adjusted(nconv), static_cast<SortRule>(selection));
        }
        const auto t2 = std::chrono::steady_clock::now();
        out[0] = std::chrono::duration<double>(t1 - t0).count();
        out[1] = double(ops_first);
        out[2] = std::chrono::duration<double>(t2 - t1).count();
        out[3] = double(s.nmatop - ops_first);
        out[4] = double(done);
    });
}

// ---- general (non-symmetric) path: LinAlg/UpperHessenbergQR.h, DoubleShiftQR.h, UpperHessenbergSchur.h,
// ---- UpperHessenbergEigen.h, GenEigsBase.h ------------------------------------------------------------
int oracle_hess_qr(long n, const double* Hm, double shift, double* Q, double* QtHQ)
{
    return guarded([&] {
        Mat M(n, n);
        std::memcpy(M.a.data(), Hm, sizeof(double) * n * n);
        UpperHessenbergQR qr;
        qr.compute(M, shift);
        if (Q)
        {
            Mat q(n, n);
            q.set_identity();
            qr.apply_YQ(q);
            std::memcpy(Q, q.a.data(), sizeof(double) * n * n);
        }
        if (QtHQ)
        {
            Mat d;
            qr.matrix_QtHQ(d);
            std::memcpy(QtHQ, d.a.data(), sizeof(double) * n * n);
        }
    });
}
int oracle_double_shift_qr(long n, const double* Hm, double s, double t, double* Q, double* QtHQ)
{
    return guarded([&] {
        Mat M(n, n);
        std::memcpy(M.a.data(), Hm, sizeof(double) * n * n);
        DoubleShiftQR qr;
        qr.compute(M, s, t);
        if (Q)
        {
            Mat q(n, n);
            q.set_identity();
            qr.apply_YQ(q);
            std::memcpy(Q, q.a.data(), sizeof(double) * n * n);
        }
        if (QtHQ)
        {
            Mat d;
            qr.matrix_QtHQ(d);
            std::memcpy(QtHQ, d.a.data(), sizeof(double) * n * n);
        }
    });
}
int oracle_hess_schur(long n, const double* Hm, double* T, double* U)
{
    return guarded([&] {
        Mat M(n, n);
        std::memcpy(M.a.data(), Hm, sizeof(double) * n * n);
        UpperHessenbergSchur sc;
        sc.compute(M);
        std::memcpy(T, sc.T.a.data(), sizeof(double) * n * n);
        std::memcpy(U, sc.U.a.data(), sizeof(double) * n * n);
    });
}
// evals: n complex (interleaved re,im); evecs: n x n complex column-major interleaved
int oracle_hess_eigen(long n, const double* Hm, double* evals, double* evecs)
{
    return guarded([&] {
        Mat M(n, n);
        std::memcpy(M.a.data(), Hm, sizeof(double) * n * n);
        UpperHessenbergEigen eg;
        eg.compute(M);
        std::memcpy(evals, eg.eivalues.data(), sizeof(Complex) * n);
        if (evecs)
        {
            std::vector<Complex> V = eg.eigenvectors();
            std::memcpy(evecs, V.data(), sizeof(Complex) * n * n);
        }
    });
}

void* oracle_geneigs_create(void* op, long nev, long ncv)
{
    GenEigs* s = nullptr;
    int rc = guarded([&] { s = new GenEigs(*static_cast<Op*>(op), nev, ncv); });
    return rc == 0 ? s : nullptr;
}
void oracle_geneigs_free(void* s) { delete static_cast<GenEigs*>(s); }
void oracle_geneigs_set_shift_invert(void* s, double sigma)
{
    static_cast<GenEigs*>(s)->shift_invert = true;
    static_cast<GenEigs*>(s)->sigma = sigma;
}
// GenEigsComplexShiftSolver: op (given at creation) is Re((A - sigma I)^{-1} .), op_probe the operator at the real probe shift
void oracle_geneigs_set_complex_shift(void* s, double sigmar, double sigmai, void* op_probe)
{
    auto* S = static_cast<GenEigs*>(s);
    S->complex_shift = true;
    S->sigmar = sigmar;
    S->sigmai = sigmai;
    S->op_probe = static_cast<Op*>(op_probe);
}
double oracle_complex_shift_probe(double sigmar) { return GenEigs::probe_shift(sigmar); }
int oracle_geneigs_init(void* s, const double* v0)
{
    auto* S = static_cast<GenEigs*>(s);
    return guarded([&] {
        if (v0)
            S->init(v0);
        else
            S->init();
    });
}
long oracle_geneigs_compute(void* s, int selection, long maxit, double tol, int sorting)
{
    auto* S = static_cast<GenEigs*>(s);
    long nconv = 0;
    int rc = guarded(
        [&] { nconv = S->compute(static_cast<SortRule>(selection), maxit, tol, static_cast<SortRule>(sorting)); });
    return rc == 0 ? nconv : rc;
}
int oracle_geneigs_info(void* s) { return static_cast<int>(static_cast<GenEigs*>(s)->info); }
long oracle_geneigs_num_iterations(void* s) { return static_cast<GenEigs*>(s)->niter; }
long oracle_geneigs_num_operations(void* s) { return static_cast<GenEigs*>(s)->nmatop; }
long oracle_geneigs_eigenvalues(void* s, double* out)  // interleaved complex
{
    std::vector<Complex> ev = static_cast<GenEigs*>(s)->eigenvalues();
    std::memcpy(out, ev.data(), sizeof(Complex) * ev.size());
    return static_cast<long>(ev.size());
}
long oracle_geneigs_eigenvectors(void* s, long nvec, double* out)  // n x ncols complex column-major interleaved
{
    Index ncols = 0;
    std::vector<Complex> X = static_cast<GenEigs*>(s)->eigenvectors(nvec, ncols);
    std::memcpy(out, X.data(), sizeof(Complex) * X.size());
    return ncols;
}

}  // extern "C"
