// Length-n dense kernels of the Lanczos / Arnoldi factorisation for gfx950 (all HBM-bound).
//
// Common shape ("orth" kernels): a 256-thread workgroup walks 128-row tiles of the column-major
// basis V (grid-stride over tiles).  Lane l of every wave owns rows (2l, 2l+1) of the tile, so
// each load is 16 B/lane and 1 KiB contiguous per wave; wave w owns basis columns j = w (mod 4),
// "slot" jj <-> column w + 4*jj.  All of a tile's loads are issued before the first use (no
// branches between them), giving up to 16 x 1 KiB in flight per wave.  Each wave keeps its V
// values in registers, which is what lets   f <- f - V c   and   c' <- V' f   share ONE pass over
// V (the reference makes two, Lanczos.h:171 and :179): the per-row sums that couple the four
// waves go through an 8 KiB double-buffered LDS exchange, one barrier per tile.
// Reductions are two-stage and atomic-free: per-workgroup records -> one fixed-order summing
// kernel, so results are reproducible run to run.
//
// Algorithmic bytes (8*n per vector pass): VTF (ncol+1), RESID_VTF (ncol+2 reads... w, v_i is
// one of the ncol columns, so ncol+1 reads + 1 write), CORRECT_VTF (ncol+1 reads + 1 write).
#include "krylov.hpp"

#include <cstdlib>
#include <string>

using namespace mispec;

namespace {

typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int kThreads = 256;
constexpr int kTileRows = 128;

// 16-byte load of two rows of a basis column with the non-temporal hint (streamed once per pass)
__device__ __forceinline__ double2 load_streamed(const double* p)
{
    const v2d t = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p));  // (plain loads: round 4, profiles/r07b — the solve 1.7 % slower)
    double2 r;
    r.x = t.x;
    r.y = t.y;
    return r;
}

__device__ __forceinline__ double wave_reduce_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_reduce_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v = fmax(v, __shfl_down(v, off, 64));
    return v;
}
__device__ __forceinline__ double block_reduce_sum(double v, double* red)
{
    v = wave_reduce_sum(v);
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// R = row pairs per lane and tile: a tile is 128*R rows, and each wave reads R consecutive KiB of every
// column it owns (longer contiguous runs per column stream).
template <int MODE, int MAXS, int R>
__global__ __launch_bounds__(kThreads) void k_orth(OrthArgs a)
{
    constexpr bool kCorrect = (MODE == ORTH_CORRECT_VTF || MODE == ORTH_CORRECT_ONLY);
    constexpr bool kVtf = (MODE != ORTH_CORRECT_ONLY);
    constexpr int kRows = kTileRows * R;
    __shared__ double cs[kPanelCols];
    __shared__ __attribute__((aligned(16))) double psum[2][4][kRows];

    if (a.status && *a.status != kStepOk)
        return;
    if (a.need_corr && *a.need_corr == 0)
        return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    if (kCorrect)
    {
        if (tid < kPanelCols)
            cs[tid] = (tid < a.ncol) ? a.c_in[a.col0 + tid] : 0.0;
        __syncthreads();
    }
    double alpha = 0.0;
    if (MODE == ORTH_RESID_VTF)
        alpha = *a.alpha_dev;

    // column base pointers and coefficients of this wave's slots; slots past ncol alias column 0
    // with coefficient 0 (their loads hit L1/L2, their results are dropped).
    const double* colp[MAXS];
    double cw[MAXS];
    double acc[MAXS];
#pragma unroll
    for (int jj = 0; jj < MAXS; jj++)
    {
        const int j = w + 4 * jj;
        colp[jj] = a.V + int64_t(a.col0 + (j < a.ncol ? j : 0)) * a.ldv;
        cw[jj] = (kCorrect && j < kPanelCols) ? cs[j] : 0.0;
        acc[jj] = 0.0;
    }
    double b2 = 0.0, mx = 0.0;

    const int64_t ntiles = (a.n + kRows - 1) / kRows;
    int buf = 0;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x)
    {
        int64_t r[R], rc[R];
        bool valid[R];
#pragma unroll
        for (int q = 0; q < R; q++)
        {
            r[q] = t * kRows + q * kTileRows + 2 * lane;
            valid[q] = r[q] < a.n;  // rows come in even pairs; vectors are zero-padded to an even length
            rc[q] = valid[q] ? r[q] : 0;
        }

        // The basis is streamed once per pass and is far larger than any cache (3.2 GB at C2); loaded with the non-temporal hint
        // it does not push the step's vectors (f, w, the newest columns: 80 MB each) out of the 256 MiB Infinity Cache, from which
        // the next kernels then read them.  Measured on C2 (profiles/r05g_*): lagged pass 667 -> 592 ms per solve, and the SpMV
        // behind it 301 -> 291 ms.
        double2 vv[MAXS][R];
#pragma unroll
        for (int jj = 0; jj < MAXS; jj++)
#pragma unroll
            for (int q = 0; q < R; q++)
                vv[jj][q] = load_streamed(colp[jj] + rc[q]);

        double2 fv[R];
#pragma unroll
        for (int q = 0; q < R; q++)
        {
            if (MODE == ORTH_RESID_VTF)
            {
                const double2 wv = *reinterpret_cast<const double2*>(a.src + rc[q]);
                const double2 vi = *reinterpret_cast<const double2*>(a.vi + rc[q]);
                fv[q].x = wv.x - alpha * vi.x;  // Lanczos.h:145
                fv[q].y = wv.y - alpha * vi.y;
            }
            else
                fv[q] = *reinterpret_cast<const double2*>(a.src + rc[q]);
            if (!valid[q])
            {
                fv[q].x = 0.0;
                fv[q].y = 0.0;
            }
        }

        if (kCorrect)
        {
#pragma unroll
            for (int q = 0; q < R; q++)
            {
                double2 p;
                p.x = 0.0;
                p.y = 0.0;
#pragma unroll
                for (int jj = 0; jj < MAXS; jj++)
                {
                    p.x += vv[jj][q].x * cw[jj];
                    p.y += vv[jj][q].y * cw[jj];
                }
                *reinterpret_cast<double2*>(&psum[buf][w][q * kTileRows + 2 * lane]) = p;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < R; q++)
            {
                const int o = q * kTileRows + 2 * lane;
                const double2 p0 = *reinterpret_cast<const double2*>(&psum[buf][0][o]);
                const double2 p1 = *reinterpret_cast<const double2*>(&psum[buf][1][o]);
                const double2 p2 = *reinterpret_cast<const double2*>(&psum[buf][2][o]);
                const double2 p3 = *reinterpret_cast<const double2*>(&psum[buf][3][o]);
                if (valid[q])
                {
                    fv[q].x -= (p0.x + p1.x) + (p2.x + p3.x);  // Lanczos.h:171 / Arnoldi.h:254
                    fv[q].y -= (p0.y + p1.y) + (p2.y + p3.y);
                }
            }
            buf ^= 1;
        }

        if (w == 0)
        {
#pragma unroll
            for (int q = 0; q < R; q++)
            {
                if (MODE != ORTH_VTF && valid[q])
                    *reinterpret_cast<double2*>(a.dst + r[q]) = fv[q];
                b2 += fv[q].x * fv[q].x + fv[q].y * fv[q].y;
                mx = fmax(mx, fmax(fabs(fv[q].x), fabs(fv[q].y)));
            }
        }
        if (kVtf)
        {
#pragma unroll
            for (int jj = 0; jj < MAXS; jj++)
#pragma unroll
                for (int q = 0; q < R; q++)
                    acc[jj] += vv[jj][q].x * fv[q].x + vv[jj][q].y * fv[q].y;
        }
    }

    double* rec = a.partials + blockIdx.x;  // slot j of this workgroup is rec[j * pstride]
    if (kVtf)
    {
#pragma unroll
        for (int jj = 0; jj < MAXS; jj++)
        {
            const double s = wave_reduce_sum(acc[jj]);
            const int j = w + 4 * jj;
            if (lane == 0 && j < a.ncol)
                rec[int64_t(a.col0 + j) * a.pstride] = s;
        }
    }
    if (w == 0 && a.norms)
    {
        b2 = wave_reduce_sum(b2);
        mx = wave_reduce_max(mx);
        if (lane == 0)
        {
            rec[kSlotBeta2 * a.pstride] = b2;
            rec[kSlotMaxAbs * a.pstride] = mx;
        }
    }
}

// ---- one-sweep variant (ORTH_LAGGED; opt-in, see krylov.hpp and DESIGN.md 3.2.1) -------------------------------------
// The correction of step i-1 and the projection of step i share one pass over V[:, :i]:
//   p = V c_in (cross-wave row sums through LDS, as in CORRECT) ;  v_i = (f - p) / beta -> column i
//   chk_j = <V_j, v_i> ;  dst = w - alpha v_i -> f ;  c_j = <V_j, dst>, c_i = <v_i, dst>, |dst|^2
// Same tiling, load pattern and record layout as k_orth; with c_in = 0 (no pending correction) column i is rewritten
// with the bits k_scale_step gave it.
// NW wavefronts per workgroup: 4 (up to 63 finished columns) or 8 (64..127: bases of up to 128 columns, round 4 — the same
// 16 columns per wavefront, twice the row sums through LDS).
// ONERED (one reduction per step, krylov.hpp FinishArgs::alpha_parts): src is u = A f~ on the UN-normalised residual; the pass
// forms w = u / beta - beta v_{i-1} itself (Lanczos.h:106 and :139 applied after the product; column i-1 is among the columns the
// pass reads: its owner wavefront hands the tile's entries to the others through LDS, next to the row sums).
template <int MAXS, int R, int NW, bool ONERED = false>
__global__ __launch_bounds__(64 * NW) void k_orth_lagged(OrthArgs a)
{
    constexpr int kRows = kTileRows * R;
    constexpr int kCols = 16 * NW;
    __shared__ double cs[kCols];
    __shared__ __attribute__((aligned(16))) double psum[2][NW][kRows];
    __shared__ __attribute__((aligned(16))) double vprev_s[ONERED ? 2 : 1][ONERED ? kRows : 2];

    if (a.status && *a.status != kStepOk)
        return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool pending = *a.pending != 0;
    if (tid < kCols)
        cs[tid] = (pending && tid < a.ncol) ? a.c_in[tid] : 0.0;
    __syncthreads();
    const double alpha = *a.alpha_dev;
    const double beta = *a.beta_dev;

    const double* colp[MAXS];
    double cw[MAXS], acc[MAXS], chk[MAXS];
#pragma unroll
    for (int jj = 0; jj < MAXS; jj++)
    {
        const int j = w + NW * jj;
        colp[jj] = a.V + int64_t(j < a.ncol ? j : 0) * a.ldv;
        cw[jj] = (j < kCols) ? cs[j] : 0.0;
        acc[jj] = 0.0;
        chk[jj] = 0.0;
    }
    double b2 = 0.0, mx = 0.0, dvi = 0.0;

    const int64_t ntiles = (a.n + kRows - 1) / kRows;
    int buf = 0;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x)
    {
        int64_t r[R], rc[R];
        bool valid[R];
#pragma unroll
        for (int q = 0; q < R; q++)
        {
            r[q] = t * kRows + q * kTileRows + 2 * lane;
            valid[q] = r[q] < a.n;
            rc[q] = valid[q] ? r[q] : 0;
        }
        double2 vv[MAXS][R];
#pragma unroll
        for (int jj = 0; jj < MAXS; jj++)
#pragma unroll
            for (int q = 0; q < R; q++)
                vv[jj][q] = load_streamed(colp[jj] + rc[q]);  // non-temporal: see k_orth
        double2 fv[R], wv[R];
#pragma unroll
        for (int q = 0; q < R; q++)
        {
            fv[q] = *reinterpret_cast<const double2*>(a.vi + rc[q]);
            wv[q] = *reinterpret_cast<const double2*>(a.src + rc[q]);
            if (!valid[q])
            {
                fv[q].x = fv[q].y = 0.0;
                wv[q].x = wv[q].y = 0.0;
            }
        }
#pragma unroll
        for (int q = 0; q < R; q++)
        {
            double2 p;
            p.x = 0.0;
            p.y = 0.0;
#pragma unroll
            for (int jj = 0; jj < MAXS; jj++)
            {
                p.x += vv[jj][q].x * cw[jj];
                p.y += vv[jj][q].y * cw[jj];
            }
            *reinterpret_cast<double2*>(&psum[buf][w][q * kTileRows + 2 * lane]) = p;
        }
        if (ONERED)
        {
            // column i-1 = slot (i-1) / NW of wavefront (i-1) % NW (a.ncol = i >= 1)
            const int jp = a.ncol - 1;
            if (w == jp % NW)
            {
#pragma unroll
                for (int jj = 0; jj < MAXS; jj++)
                    if (jj == jp / NW)
                    {
#pragma unroll
                        for (int q = 0; q < R; q++)
                            *reinterpret_cast<double2*>(&vprev_s[buf][q * kTileRows + 2 * lane]) = vv[jj][q];
                    }
            }
        }
        __syncthreads();
        if (ONERED)
        {
#pragma unroll
            for (int q = 0; q < R; q++)
            {
                const double2 vp = *reinterpret_cast<const double2*>(&vprev_s[buf][q * kTileRows + 2 * lane]);
                wv[q].x = wv[q].x / beta - beta * vp.x;
                wv[q].y = wv[q].y / beta - beta * vp.y;
                if (!valid[q])
                {
                    wv[q].x = 0.0;
                    wv[q].y = 0.0;
                }
            }
        }
        double2 vi[R], fn[R];
#pragma unroll
        for (int q = 0; q < R; q++)
        {
            const int o = q * kTileRows + 2 * lane;
            const double2 p0 = *reinterpret_cast<const double2*>(&psum[buf][0][o]);
            const double2 p1 = *reinterpret_cast<const double2*>(&psum[buf][1][o]);
            const double2 p2 = *reinterpret_cast<const double2*>(&psum[buf][2][o]);
            const double2 p3 = *reinterpret_cast<const double2*>(&psum[buf][3][o]);
            double px = (p0.x + p1.x) + (p2.x + p3.x), py = (p0.y + p1.y) + (p2.y + p3.y);
            if (NW == 8)
            {
                const double2 p4 = *reinterpret_cast<const double2*>(&psum[buf][NW - 4][o]);
                const double2 p5 = *reinterpret_cast<const double2*>(&psum[buf][NW - 3][o]);
                const double2 p6 = *reinterpret_cast<const double2*>(&psum[buf][NW - 2][o]);
                const double2 p7 = *reinterpret_cast<const double2*>(&psum[buf][NW - 1][o]);
                px += (p4.x + p5.x) + (p6.x + p7.x);
                py += (p4.y + p5.y) + (p6.y + p7.y);
            }
            vi[q].x = (fv[q].x - px) / beta;  // Lanczos.h:171 then :106 (true division)
            vi[q].y = (fv[q].y - py) / beta;
            if (!valid[q])  // rows past the end were loaded from row 0 (clamped address): they must not reach the sums
            {
                vi[q].x = 0.0;
                vi[q].y = 0.0;
            }
            fn[q].x = wv[q].x - alpha * vi[q].x;  // Lanczos.h:145
            fn[q].y = wv[q].y - alpha * vi[q].y;
        }
        buf ^= 1;
        if (w == 0)
        {
#pragma unroll
            for (int q = 0; q < R; q++)
            {
                if (valid[q])
                {
                    // column i and f stay cacheable: the next product reads both (a non-temporal store of the column measured
                    // 8 ms per solve slower for the SpMV, profiles/r05i_*)
                    *reinterpret_cast<double2*>(a.vout + r[q]) = vi[q];
                    *reinterpret_cast<double2*>(a.dst + r[q]) = fn[q];
                }
                b2 += fn[q].x * fn[q].x + fn[q].y * fn[q].y;
                dvi += vi[q].x * fn[q].x + vi[q].y * fn[q].y;
                mx = fmax(mx, fmax(fabs(fn[q].x), fabs(fn[q].y)));
            }
        }
#pragma unroll
        for (int jj = 0; jj < MAXS; jj++)
#pragma unroll
            for (int q = 0; q < R; q++)
            {
                acc[jj] += vv[jj][q].x * fn[q].x + vv[jj][q].y * fn[q].y;
                chk[jj] += vv[jj][q].x * vi[q].x + vv[jj][q].y * vi[q].y;
            }
    }

    double* rec = a.partials + blockIdx.x;
#pragma unroll
    for (int jj = 0; jj < MAXS; jj++)
    {
        const double s = wave_reduce_sum(acc[jj]);
        const double c = wave_reduce_sum(chk[jj]);
        const int j = w + NW * jj;
        if (lane == 0 && j < a.ncol)
        {
            rec[int64_t(j) * a.pstride] = s;
            rec[int64_t(a.ncol + 1 + j) * a.pstride] = c;
        }
    }
    if (w == 0)
    {
        b2 = wave_reduce_sum(b2);
        dvi = wave_reduce_sum(dvi);
        mx = wave_reduce_max(mx);
        if (lane == 0)
        {
            rec[int64_t(a.ncol) * a.pstride] = dvi;
            rec[kSlotBeta2 * a.pstride] = b2;
            rec[kSlotMaxAbs * a.pstride] = mx;
        }
    }
}

// What the scalar tail of a lagged step reads from global memory, loaded in ONE round at the start of the reducing kernel (the
// loads then overlap the trip to the partial records; read one after the other behind the sums they cost a dependent round trip
// each — 10 us of a 10.6 us k_finish, profiles/r11d).  Every value is written by earlier kernels only.
struct LagPre
{
    int status, lag_pending;
    double beta, lag_chk_max, lag_rel_c_max, alpha, cp1, cp2, subd2, diag1;
    long long lag_steps, onered_steps;
};
__device__ __forceinline__ LagPre load_lag_pre(const FinishArgs& fa)
{
    const StepState* st = fa.st;
    const int i = fa.step;
    LagPre p;
    p.status = st->status;
    p.lag_pending = st->lag_pending;
    p.beta = st->beta;
    p.lag_chk_max = st->lag_chk_max;
    p.lag_rel_c_max = st->lag_rel_c_max;
    p.lag_steps = st->lag_steps;
    p.onered_steps = st->onered_steps;
    p.alpha = *fa.alpha_src;
    p.cp1 = fa.prev_red[i - 1];
    p.cp2 = i >= 2 ? fa.prev_red[i - 2] : 0.0;
    p.subd2 = i >= 2 ? st->subd[i - 2] : 0.0;
    p.diag1 = st->diag[i - 1];
    return p;
}

// max |v[j]| and the running sum of v[j]^2 over j < n, in index order; eight loads in flight (one thread walking LDS or global
// memory pays a full load latency per element otherwise)
__device__ __forceinline__ void scan_abs_sq(const double* v, int n, double& mx, double& sq)
{
    int j = 0;
    for (; j + 8 <= n; j += 8)
    {
        double x[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            x[k] = v[j + k];
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            mx = fmax(mx, fabs(x[k]));
            sq += x[k] * x[k];
        }
    }
    for (; j < n; j++)
    {
        mx = fmax(mx, fabs(v[j]));
        sq += v[j] * v[j];
    }
}

// Executed by ONE thread after the reduction of an ORTH_LAGGED record (layout above, i = fa.step columns were final).
// H of step i follows from the Lanczos relation of step i-1; the decisions are the reference's
// (Lanczos.h:156-168) with "apply the correction now" replaced by "carry it into the next sweep" whenever that is safe.
// `pre`: the state as loaded by load_lag_pre; status / beta: the state this step leaves (what start_next_onered continues from).
__device__ void finish_lagged(double* red, const FinishArgs& fa, const LagPre& pre, int& status, double& beta)
{
    StepState* st = fa.st;
    const int i = fa.step;
    status = pre.status;
    beta = pre.beta;
    if (pre.status != kStepOk)
        return;
    const bool pend = pre.lag_pending != 0;
    const double bprev = pre.beta;  // the divisor column i was formed with
    if (pend)
    {
        double cerr = 0.0, unused = 0.0;
        scan_abs_sq(red + i + 1, i, cerr, unused);
        st->lag_chk_max = fmax(pre.lag_chk_max, cerr);
        if (cerr > fa.eps)  // Lanczos.h:156 after the first correction, in units of beta
        {
            st->status = status = kStepLagCheck;
            st->stop_step = i;
            st->stop_count = 1;
            return;
        }
    }
    const double alpha = pre.alpha;
    double d = alpha, sub = bprev;
    if (pend)  // cp = the accepted V'f of step i-1
    {
        d -= pre.cp1;
        sub -= ((i >= 2 ? pre.subd2 * pre.cp2 : 0.0) + pre.diag1 * pre.cp1) / bprev;
    }
    const double gamma2 = red[kSlotBeta2];
    const double gamma = sqrt(gamma2);
    double err = 0.0, c2 = 0.0;
    scan_abs_sq(red, i + 1, err, c2);
    red[kSlotBeta] = gamma;
    red[kSlotErr] = err;
    st->alpha = alpha;
    st->err = err;
    st->need_corr = 0;
    st->lag_steps = pre.lag_steps + 1;
    double d_out = d, sub_out = sub;
    int count = 0, lag_pending = 0;
    beta = gamma;
    const int need = err > fa.eps * gamma;  // Lanczos.h:156
    if (need && gamma < fa.beta_thresh)     // Lanczos.h:163
    {
        status = kStepTinyF;
        st->status = status;
        st->stop_step = i;
        st->stop_count = 0;
    }
    else if (fa.lag_last)  // the CORRECT_VTF launches that follow finish f the reference's way
        st->need_corr = need;
    else if (need)
    {
        const double b2 = gamma2 - c2;
        if (c2 <= fa.lag_limit * gamma2 && b2 > 0.0 && sqrt(b2) >= fa.eps_sqrt)
        {
            sub_out = sub + red[i - 1];  // Lanczos.h:173-175
            d_out = d + red[i];
            beta = sqrt(b2);             // |f - V c| for V'V = I, f'V = c'
            lag_pending = 1;
            count = 1;
            st->lag_rel_c_max = fmax(pre.lag_rel_c_max, sqrt(c2) / gamma);
        }
        else  // the reference's loop on the host, from the uncorrected state
        {
            status = kStepMoreCorr;
            st->status = status;
            st->stop_step = i;
            st->stop_count = 0;
        }
    }
    st->diag[i] = d_out;
    st->subd[i - 1] = sub_out;
    st->beta = beta;
    st->count = count;
    st->lag_pending = lag_pending;
}
__device__ void finish_lagged(double* red, const FinishArgs& fa)
{
    const LagPre pre = load_lag_pre(fa);
    int status;
    double beta;
    finish_lagged(red, fa, pre, status, beta);
}

// One reduction per step: after the bookkeeping of step i the same thread starts step i + 1 — its beta < sqrt(eps) stop
// (Lanczos.h:107; the restart heuristics need a finished f: host) and alpha~ = <f~, A f~> / beta^2 - <f~, v_i> from the sum s1 that
// travelled with the record (oracle/onesweep_variant.hpp, one_reduction).  status / beta: what finish_lagged left.
__device__ void start_next_onered(const double* red, const FinishArgs& fa, double s1, int status, double beta, long long onered_steps)
{
    StepState* st = fa.st;
    if (status != kStepOk)
        return;
    if (beta < fa.eps_sqrt)
    {
        st->status = kStepSmallBeta;
        st->stop_step = fa.step + 1;
        st->stop_count = 0;
        return;
    }
    *fa.alpha_out = s1 / (beta * beta) - red[fa.step];
    st->onered_steps = onered_steps + 1;
}

// Executed by ONE thread after the reduction of a k_vq_fused record (krylov.hpp VqFusedArgs): what mispec_fac_restart_sym used to
// do on the host after a stream synchronisation — now the sweep that follows the restart is enqueued at once and starts from
// this state.
__device__ void finish_fused_restart(double* red, int ncol, const FinishArgs& fa)
{
    const int m = ncol - 1;
    double err = 0.0;
    for (int j = 0; j < m; j++)
        err = fmax(err, fabs(red[j]));
    const double beta_corr = sqrt(red[kSlotBeta2]);
    red[kSlotBeta] = beta_corr;
    red[kSlotErr] = err;
    StepState* st = fa.st;
    if (st->status != kStepOk)
        return;
    st->beta = sqrt(red[m]);  // Arnoldi.h:339
    st->rst_err = err;
    st->rst_beta_corr = beta_corr;
    if (err > fa.eps * beta_corr)  // Lanczos.h:156 with count = 1
    {
        st->status = kStepRestartCheck;
        st->stop_step = fa.step;
        st->stop_count = 1;
    }
}

// Executed by ONE thread after a reduction: the scalar tail of a Lanczos step.
__device__ void finish_record(double* red, int ncol, const FinishArgs& fa)
{
    if (fa.mode == kFinishNone)
        return;
    if (fa.mode == kFinishFusedRestart)
    {
        finish_fused_restart(red, ncol, fa);
        return;
    }
    if (fa.mode == kFinishLagged)
    {
        finish_lagged(red, fa);
        return;
    }
    const double beta = sqrt(red[kSlotBeta2]);  // ArnoldiOp.h:152-155: plain sqrt(sum x^2)
    double err = 0.0;
    for (int j = 0; j < ncol; j++)
        err = fmax(err, fabs(red[j]));  // Lanczos.h:153 cwiseAbs().maxCoeff()
    red[kSlotBeta] = beta;
    red[kSlotErr] = err;
    if (fa.mode == kFinishNorms)
        return;
    StepState* st = fa.st;
    if (st->status != kStepOk)
        return;
    if (fa.mode == kFinishArnoldiH)
    {
        double s = 0.0;
        for (int j = 0; j < ncol; j++)
        {
            fa.hcol[j] = red[j];                            // Arnoldi.h:251
            s = __dadd_rn(s, __dmul_rn(red[j], red[j]));    // the host path's plain loop (no contraction)
        }
        st->alpha = sqrt(s);
        return;
    }
    if (fa.mode == kFinishArnoldiF)
    {
        st->beta = beta;
        st->err = err;
        st->count = 0;
        st->need_corr = 0;
        if (beta > 0.717 * st->alpha)  // Arnoldi.h:257: no re-orthogonalisation needed
            return;
        if (err > fa.eps * beta)       // Arnoldi.h:266: corrections run on the host path
        {
            st->status = (beta < fa.beta_thresh) ? kStepTinyF : kStepMoreCorr;
            st->stop_step = fa.step;
            st->stop_count = 0;
        }
        return;
    }
    if (fa.mode == kFinishStepFirst)
    {
        st->alpha = *fa.alpha_src;
        st->diag[fa.step] = st->alpha;  // Lanczos.h:142
        st->count = 0;
    }
    else
    {
        if (!st->need_corr)
            return;  // this correction was not executed
        st->subd[fa.step - 1] += fa.prev_red[fa.step - 1];  // Lanczos.h:173-175
        st->diag[fa.step] += fa.prev_red[fa.step];
        st->count++;
    }
    st->beta = beta;
    st->err = err;
    int need = (st->count < 5) && (err > fa.eps * beta);  // Lanczos.h:156
    if (need && beta < fa.beta_thresh)                    // Lanczos.h:163
    {
        st->status = kStepTinyF;
        st->stop_step = fa.step;
        st->stop_count = st->count;
        need = 0;
    }
    else if (need && st->count >= fa.max_spec)
    {
        st->status = kStepMoreCorr;
        st->stop_step = fa.step;
        st->stop_count = st->count;
    }
    st->need_corr = need;
}

// One workgroup sums the records in a fixed order.  A wave owns up to three slots per pass; its lane l adds
// the records l, l+64, l+128, ... of each.  All loads of a pass are issued before the first add: the records were
// written by other XCDs, so they come from beyond the L2 and one load latency is several microseconds — the kernel
// must pay it once, not once per slot.  One shuffle tree per slot then finishes the sum.
template <int NS>
__device__ __forceinline__ void reduce_slots(const double* __restrict__ partials, int64_t pstride, int nrec, const int (&slot)[NS],
                                             int lane, double* sh)
{
    constexpr int kBatch = 16;  // records per lane and round: 1024 records per round
    double acc[NS];
#pragma unroll
    for (int s = 0; s < NS; s++)
        acc[s] = 0.0;
    for (int base = 0; base < nrec; base += 64 * kBatch)
    {
        double x[NS][kBatch];
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int k = 0; k < kBatch; k++)
            {
                const int b = base + lane + 64 * k;
                x[s][k] = (slot[s] >= 0 && b < nrec) ? partials[int64_t(slot[s]) * pstride + b] : 0.0;
            }
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int k = 0; k < kBatch; k++)
                acc[s] = (slot[s] == kSlotMaxAbs) ? fmax(acc[s], x[s][k]) : acc[s] + x[s][k];
    }
#pragma unroll
    for (int s = 0; s < NS; s++)
    {
        if (slot[s] < 0)
            continue;  // wave-uniform
        const double r = (slot[s] == kSlotMaxAbs) ? wave_reduce_max(acc[s]) : wave_reduce_sum(acc[s]);
        if (lane == 0)
            sh[slot[s]] = r;
    }
}

// Every slot of a record of at most 64 * RPL workgroups in ONE round of loads (k_reduce_partials, few records): wave g owns the
// slots g, g + 16, ..., g + 80 (the two scalar slots ride behind the columns); with `alpha`, thread t also loads the product's
// partial sums t, t + 1024, ... (at most eight) in the same round — k_reduce_sum's order.
template <int RPL>
__device__ __forceinline__ void reduce_one_round(const double* __restrict__ partials, int64_t pstride, int nrec, int ncol,
                                                 const FinishArgs& fin, bool alpha, double* sh, double* sh_a)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
    int sl[6];
    double xv[6][RPL];
#pragma unroll
    for (int u = 0; u < 6; u++)
    {
        const int s = g + 16 * u;
        sl[u] = s < ncol ? s : (s == ncol ? kSlotBeta2 : (s == ncol + 1 ? kSlotMaxAbs : -1));
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int b = lane + 64 * k;
            xv[u][k] = (sl[u] >= 0 && b < nrec) ? partials[int64_t(sl[u]) * pstride + b] : 0.0;
        }
    }
    double av[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
    {
        const int64_t i = tid + int64_t(k) * 1024;
        av[k] = (alpha && i < fin.alpha_count) ? fin.alpha_parts[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 6; u++)
    {
        if (sl[u] < 0)
            continue;  // wave-uniform
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < RPL; k++)
            acc = (sl[u] == kSlotMaxAbs) ? fmax(acc, xv[u][k]) : acc + xv[u][k];
        const double r = (sl[u] == kSlotMaxAbs) ? wave_reduce_max(acc) : wave_reduce_sum(acc);
        if (lane == 0)
            sh[sl[u]] = r;
    }
    if (alpha)
    {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 8; k++)
            v += av[k];
        v = wave_reduce_sum(v);
        if (lane == 0)
            sh_a[g] = v;
    }
}

__global__ __launch_bounds__(1024) void k_reduce_partials(const double* __restrict__ partials, int64_t pstride, int nrec,
                                                           int ncol, double* __restrict__ red, FinishArgs fin)
{
    __shared__ double sh[kPartialLd];
    // A device-driven run that has stopped: the launches still in the queue are no-ops, and so is this one — the two record
    // halves must stay what the stopping pass and the one before it left there, the host continues from them.  The flag is
    // LOADED here and looked at behind the sums (nothing is written before that): its latency overlaps the trip to the records.
    const int run_status = (fin.st != nullptr) ? fin.st->status : int(kStepOk);
    const bool lagged_tail = fin.mode == kFinishLagged && fin.st != nullptr;
    LagPre pre = {};
    if (lagged_tail && threadIdx.x == 0)
        pre = load_lag_pre(fin);
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int t = tid; t < kPartialLd; t += 1024)
        sh[t] = 0.0;
    __syncthreads();
    __shared__ double sh_a[16];
    // Few records (at most 256: every run of the LDS-DMA pass, whose grid is one workgroup per CU, and small problems): a lane
    // holds RPL = 1, 2 or 4 records of each of its wave's six slots, so ALL the slots of the record and (up to 8192 of) the
    // product's partial sums fit ONE round of loads — one trip to the other XCDs' data instead of three, on the critical path of
    // every step.  Same sums as the passes below: a lane adds its records l, l + 64, ... in that order, then the shuffle tree.
    const bool small = nrec <= 256 && ncol + 2 <= 96;
    const bool alpha_in_round = small && fin.alpha_parts && fin.alpha_count <= 8192;
    if (small)
    {
        if (nrec <= 64)
            reduce_one_round<1>(partials, pstride, nrec, ncol, fin, alpha_in_round, sh, sh_a);
        else if (nrec <= 128)
            reduce_one_round<2>(partials, pstride, nrec, ncol, fin, alpha_in_round, sh, sh_a);
        else
            reduce_one_round<4>(partials, pstride, nrec, ncol, fin, alpha_in_round, sh, sh_a);
    }
    // column passes of 48 (wave g: c0+g, c0+g+16, c0+g+32); the two scalar slots ride in the last pass when its
    // positions 46/47 are free, otherwise in a pass of their own
    bool scalars_done = small;
    for (int c0 = 0; !small && (c0 == 0 || c0 < ncol); c0 += 48)
    {
        int slot[3];
        slot[0] = (c0 + g < ncol) ? c0 + g : -1;
        slot[1] = (c0 + g + 16 < ncol) ? c0 + g + 16 : -1;
        slot[2] = (c0 + g + 32 < ncol) ? c0 + g + 32 : -1;
        if (ncol <= c0 + 46)  // block-uniform: only true in the last pass
        {
            if (g == 14)
                slot[2] = kSlotMaxAbs;
            if (g == 15)
                slot[2] = kSlotBeta2;
            scalars_done = true;
        }
        reduce_slots<3>(partials, pstride, nrec, slot, lane, sh);
    }
    if (!scalars_done)
    {
        int slot[1];
        slot[0] = (g == 0) ? kSlotBeta2 : (g == 1 ? kSlotMaxAbs : -1);
        reduce_slots<1>(partials, pstride, nrec, slot, lane, sh);
    }
    if (fin.alpha_parts && !alpha_in_round)
    {
        // k_reduce_sum's order: thread t adds in[t], in[t + 1024], ... ; wave sums; the 16 wave sums one after the other
        constexpr int kBatch = 40;
        double v = 0.0;
        for (int64_t base = tid; base < fin.alpha_count; base += int64_t(kBatch) * 1024)
        {
            double x[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; k++)
            {
                const int64_t i = base + int64_t(k) * 1024;
                x[k] = (i < fin.alpha_count) ? fin.alpha_parts[i] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < kBatch; k++)
                v += x[k];
        }
        v = wave_reduce_sum(v);
        if (lane == 0)
            sh_a[g] = v;
    }
    __syncthreads();
    if (run_status != kStepOk)  // (uniform: every thread loaded the same word)
        return;
    if (tid == 0)
    {
        double s1 = 0.0;
        if (fin.alpha_parts)
            for (int k = 0; k < 16; k++)
                s1 += sh_a[k];
        if (fin.packed)
        {
            sh[ncol] = sh[kSlotBeta2];
            if (fin.alpha_parts)
                sh[ncol + 1] = s1;  // sharded: the local sum travels behind sum f^2
        }
        if (lagged_tail)
        {
            int status;
            double beta;
            finish_lagged(sh, fin, pre, status, beta);
            if (fin.alpha_parts && !fin.packed)
                start_next_onered(sh, fin, s1, status, beta, pre.onered_steps);
        }
        else
            finish_record(sh, ncol, fin);
    }
    __syncthreads();
    for (int t = tid; t < kPartialLd; t += 1024)
        red[t] = sh[t];
}

// Sharded runs: the all-reduced record sits in a staging area; it becomes the content of a record half only while the run is live.
__global__ __launch_bounds__(256) void k_finish(const double* __restrict__ stage, double* __restrict__ red, int ncol, FinishArgs fin)
{
    __shared__ double sh[kPartialLd];
    const int run_status = (fin.st != nullptr) ? fin.st->status : int(kStepOk);
    const bool lagged_tail = fin.mode == kFinishLagged && fin.st != nullptr;
    LagPre pre = {};
    if (lagged_tail && threadIdx.x == 0)
        pre = load_lag_pre(fin);
    // the record goes through LDS: the tail's walks over it then cost LDS latencies, and `red` is written once, complete
    for (int t = threadIdx.x; t < kPartialLd; t += 256)
        sh[t] = stage[t];
    __syncthreads();
    if (run_status != kStepOk)
        return;
    if (threadIdx.x == 0)
    {
        double s1 = 0.0;
        if (fin.packed)
        {
            sh[kSlotBeta2] = sh[ncol];
            if (ncol != kSlotBeta2)
                sh[ncol] = 0.0;
            if (fin.alpha_parts)
            {
                s1 = sh[ncol + 1];
                sh[ncol + 1] = 0.0;
            }
        }
        if (lagged_tail)
        {
            int status;
            double beta;
            finish_lagged(sh, fin, pre, status, beta);
            if (fin.alpha_parts)
                start_next_onered(sh, fin, s1, status, beta, pre.onered_steps);
        }
        else
            finish_record(sh, ncol, fin);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < kPartialLd; t += 256)
        red[t] = sh[t];
}

__global__ __launch_bounds__(256) void k_publish_state(const StepState* __restrict__ st, int m, StepState* host_dst,
                                                        unsigned long long* host_flag, unsigned long long seq)
{
    const int tid = threadIdx.x;
    if (tid == 0)
    {
        host_dst->beta = st->beta;
        host_dst->alpha = st->alpha;
        host_dst->err = st->err;
        host_dst->count = st->count;
        host_dst->need_corr = st->need_corr;
        host_dst->status = st->status;
        host_dst->stop_step = st->stop_step;
        host_dst->stop_count = st->stop_count;
        host_dst->lag_pending = st->lag_pending;
        host_dst->lag_rel_c_max = st->lag_rel_c_max;
        host_dst->lag_chk_max = st->lag_chk_max;
        host_dst->lag_steps = st->lag_steps;
        host_dst->rst_err = st->rst_err;
        host_dst->rst_beta_corr = st->rst_beta_corr;
        host_dst->onered_steps = st->onered_steps;
    }
    for (int j = tid; j < m; j += 256)
    {
        host_dst->diag[j] = st->diag[j];
        host_dst->subd[j] = st->subd[j];
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0)
        __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Start of a restart: Q (m x m) and the start state of the next sweep (diag / subd after compress_H) are read from a pinned
// host buffer [Q m*m][diag m][subd m] by ONE small kernel in front of the V*Q pass (instead of two hipMemcpyAsync); one element
// per thread: every read crosses PCIe, so all of them have to be in flight at once (one round trip, not four).
__global__ __launch_bounds__(256) void k_fetch_restart(const double* __restrict__ host_src, int m, double* __restrict__ Qdev,
                                                        StepState* st, int with_state)
{
    const int tid = threadIdx.x + blockIdx.x * 256;
    const int nq = m * m;
    for (int i = tid; i < nq; i += 256 * gridDim.x)
        Qdev[i] = host_src[i];
    if (!with_state || blockIdx.x != 0)
        return;
    for (int j = threadIdx.x; j < m; j += 256)
    {
        st->diag[j] = host_src[nq + j];
        st->subd[j] = host_src[nq + m + j];
    }
    if (threadIdx.x == 0)
    {
        st->beta = 0.0;
        st->alpha = 0.0;
        st->err = 0.0;
        st->count = 0;
        st->need_corr = 0;
        st->status = kStepOk;
        st->stop_step = 0;
        st->stop_count = 0;
        st->lag_pending = 0;
        st->lag_rel_c_max = 0.0;
        st->lag_chk_max = 0.0;
        st->lag_steps = 0;
        st->rst_err = 0.0;
        st->rst_beta_corr = 0.0;
        st->onered_steps = 0;
    }
}

__global__ __launch_bounds__(1024) void k_reduce_sum(const double* __restrict__ in, int64_t count, double* __restrict__ out)
{
    __shared__ double sh[16];
    double v = 0.0;
    // thread t adds in[t], in[t+1024], ... in that order; the loads of a batch are all issued before the first add
    // (the partials were written by other XCDs: one load latency per batch instead of one per element)
    constexpr int kBatch = 40;
    for (int64_t base = threadIdx.x; base < count; base += int64_t(kBatch) * 1024)
    {
        double x[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; k++)
        {
            const int64_t i = base + int64_t(k) * 1024;
            x[k] = (i < count) ? in[i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < kBatch; k++)
            v += x[k];
    }
    v = wave_reduce_sum(v);
    if ((threadIdx.x & 63) == 0)
        sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double s = 0.0;
        for (int k = 0; k < 16; k++)
            s += sh[k];
        out[0] = s;
    }
}

__global__ __launch_bounds__(kThreads) void k_scale(const double* __restrict__ src, double* __restrict__ dst, int64_t npairs,
                                                     double divisor)
{
    for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < npairs; i += int64_t(gridDim.x) * kThreads)
    {
        double2 v = reinterpret_cast<const double2*>(src)[i];
        v.x = v.x / divisor;  // the reference divides (Lanczos.h:106 `f / beta`), it does not multiply by 1/beta
        v.y = v.y / divisor;
        reinterpret_cast<double2*>(dst)[i] = v;
    }
}

__global__ __launch_bounds__(kThreads) void k_scale_step(const double* __restrict__ src, double* __restrict__ dst,
                                                          int64_t npairs, StepState* st, int step, double eps_sqrt)
{
    if (st->status != kStepOk)
        return;
    const double beta = st->beta;
    if (beta < eps_sqrt)  // includes beta < near_0: Lanczos.h:99 and :107-113 are resolved on the host path
    {
        if (blockIdx.x == 0 && threadIdx.x == 0)
        {
            st->status = kStepSmallBeta;
            st->stop_step = step;
            st->stop_count = 0;
        }
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        st->subd[step - 1] = beta;  // Lanczos.h:127-128
    for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < npairs; i += int64_t(gridDim.x) * kThreads)
    {
        double2 v = reinterpret_cast<const double2*>(src)[i];
        v.x = v.x / beta;
        v.y = v.y / beta;
        reinterpret_cast<double2*>(dst)[i] = v;
    }
}

__global__ __launch_bounds__(kThreads) void k_axpby(double* __restrict__ f, double a, const double* __restrict__ v, double b,
                                                     int64_t npairs, double* __restrict__ partials, int64_t pstride)
{
    __shared__ double red[4];
    double b2 = 0.0;
    for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < npairs; i += int64_t(gridDim.x) * kThreads)
    {
        double2 x = reinterpret_cast<double2*>(f)[i];
        const double2 y = reinterpret_cast<const double2*>(v)[i];
        x.x = x.x * a + y.x * b;  // Arnoldi.h:337
        x.y = x.y * a + y.y * b;
        reinterpret_cast<double2*>(f)[i] = x;
        b2 += x.x * x.x + x.y * x.y;
    }
    const double tot = block_reduce_sum(b2, red);
    if (threadIdx.x == 0)
    {
        partials[kSlotBeta2 * pstride + blockIdx.x] = tot;
        partials[kSlotMaxAbs * pstride + blockIdx.x] = 0.0;
    }
}

__global__ __launch_bounds__(kThreads) void k_lanczos_epilogue(double* __restrict__ w, const double* __restrict__ v,
                                                                const double* __restrict__ v_prev, double h_prev,
                                                                int64_t npairs, double* __restrict__ partials,
                                                                const double* __restrict__ h_prev_dev, const int* __restrict__ status)
{
    __shared__ double red[4];
    if (status && *status != 0)
        return;
    if (h_prev_dev)
        h_prev = *h_prev_dev;
    double acc = 0.0;
    for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < npairs; i += int64_t(gridDim.x) * kThreads)
    {
        double2 wv = reinterpret_cast<double2*>(w)[i];
        const double2 vv = reinterpret_cast<const double2*>(v)[i];
        if (v_prev)
        {
            const double2 pv = reinterpret_cast<const double2*>(v_prev)[i];
            wv.x -= h_prev * pv.x;
            wv.y -= h_prev * pv.y;
            reinterpret_cast<double2*>(w)[i] = wv;
        }
        acc += vv.x * wv.x + vv.y * wv.y;
    }
    const double tot = block_reduce_sum(acc, red);
    if (threadIdx.x == 0)
        partials[blockIdx.x] = tot;
}

// r = y - lambda x : records get |r|^2 (kSlotBeta2) and |x|^2 (slot 0)
__global__ __launch_bounds__(kThreads) void k_resid_norms(const double* __restrict__ y, const double* __restrict__ x,
                                                           double lambda, int64_t npairs, double* __restrict__ partials,
                                                           int64_t pstride)
{
    __shared__ double red[4];
    __shared__ double red2[4];
    double r2 = 0.0, x2 = 0.0;
    for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < npairs; i += int64_t(gridDim.x) * kThreads)
    {
        const double2 yv = reinterpret_cast<const double2*>(y)[i];
        const double2 xv = reinterpret_cast<const double2*>(x)[i];
        const double r0 = yv.x - lambda * xv.x, r1 = yv.y - lambda * xv.y;
        r2 += r0 * r0 + r1 * r1;
        x2 += xv.x * xv.x + xv.y * xv.y;
    }
    const double t1 = block_reduce_sum(r2, red);
    const double t2 = block_reduce_sum(x2, red2);
    if (threadIdx.x == 0)
    {
        partials[kSlotBeta2 * pstride + blockIdx.x] = t1;
        partials[0 * pstride + blockIdx.x] = t2;
        partials[kSlotMaxAbs * pstride + blockIdx.x] = 0.0;
    }
}

__global__ __launch_bounds__(kThreads) void k_resid_norms_complex(const double* __restrict__ yr, const double* __restrict__ yi,
                                                                   const double* __restrict__ xr, const double* __restrict__ xi,
                                                                   double a, double b, int64_t npairs,
                                                                   double* __restrict__ partials, int64_t pstride)
{
    __shared__ double red[4];
    __shared__ double red2[4];
    double r2 = 0.0, x2 = 0.0;
    for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < npairs; i += int64_t(gridDim.x) * kThreads)
    {
        const double2 pr = reinterpret_cast<const double2*>(yr)[i], pi = reinterpret_cast<const double2*>(yi)[i];
        const double2 qr = reinterpret_cast<const double2*>(xr)[i], qi = reinterpret_cast<const double2*>(xi)[i];
        // (y_r + i y_i) - (a + i b)(x_r + i x_i)
        const double e0 = pr.x - (a * qr.x - b * qi.x), f0 = pi.x - (b * qr.x + a * qi.x);
        const double e1 = pr.y - (a * qr.y - b * qi.y), f1 = pi.y - (b * qr.y + a * qi.y);
        r2 += e0 * e0 + f0 * f0 + e1 * e1 + f1 * f1;
        x2 += qr.x * qr.x + qi.x * qi.x + qr.y * qr.y + qi.y * qi.y;
    }
    const double t1 = block_reduce_sum(r2, red);
    const double t2 = block_reduce_sum(x2, red2);
    if (threadIdx.x == 0)
    {
        partials[kSlotBeta2 * pstride + blockIdx.x] = t1;
        partials[0 * pstride + blockIdx.x] = t2;
        partials[kSlotMaxAbs * pstride + blockIdx.x] = 0.0;
    }
}

// X[:, 0:p] = V[:, 0:m] * Q.  128-row tiles of V are staged in LDS (all m columns), so the product
// may overwrite V in place (compress_V): a tile's rows are private to its workgroup and fully read
// before the first write.  Wave w produces output columns i = w (mod 4).  Q is re-laid out in LDS
// so that a wave's coefficients for one j are contiguous (broadcast ds_read_b128).
// this wave's columns (w, w+4, ...) of one 128-row tile -> registers; surplus slots re-read column w (column 0 for a wave that
// has no column at all: m < 4 — column w would lie behind the basis)
template <int NJ>
__device__ __forceinline__ void vq_fetch(v2d (&pre)[NJ], const double* __restrict__ V, int64_t ldv, int w, int nj,
                                         int64_t r, int64_t n)
{
    const int64_t rc = (r < n) ? r : 0;
#pragma unroll
    for (int jj = 0; jj < NJ; jj++)
    {
        const int jc = (jj < nj) ? (w + 4 * jj) : (nj > 0 ? w : 0);
        pre[jj] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(V + int64_t(jc) * ldv + rc));  // streamed once: see k_orth
    }
}

template <int MAXS>
__global__ __launch_bounds__(kThreads) void k_vq(const double* __restrict__ V, int64_t ldv, int m,
                                                  const double* __restrict__ Q, int ldq, int p, double* X, int64_t ldx,
                                                  int64_t n, int accumulate)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Vt = smem;                          // [m][128]
    double* Qs = smem + int64_t(m) * kTileRows;  // [m][4][MAXS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int kMaxJ = kPanelCols / 4;  // input columns per wave
    const int nj = (m - w + 3) / 4;          // this wave stages columns w, w+4, ...

    for (int idx = tid; idx < m * 4 * MAXS; idx += kThreads)
    {
        const int jj = idx % MAXS, ww = (idx / MAXS) & 3, j = idx / (4 * MAXS);
        const int i = ww + 4 * jj;
        Qs[idx] = (i < p) ? Q[j + int64_t(i) * ldq] : 0.0;
    }

    // Software pipeline: the next tile's rows travel HBM -> registers while the current tile is multiplied
    // out of LDS, so the load latency hides behind the FMAs and stores of the tile before.
    const int64_t ntiles = (n + kTileRows - 1) / kTileRows;
    v2d pre[kMaxJ];
    int64_t t = blockIdx.x;
    if (t < ntiles)
        vq_fetch<kMaxJ>(pre, V, ldv, w, nj, t * kTileRows + 2 * lane, n);
    for (; t < ntiles; t += gridDim.x)
    {
        const int64_t r = t * kTileRows + 2 * lane;
        const bool valid = r < n;
        __syncthreads();  // previous tile fully consumed (and Qs written, first time round)
#pragma unroll
        for (int jj = 0; jj < kMaxJ; jj++)
            if (jj < nj)
                *reinterpret_cast<v2d*>(&Vt[(w + 4 * jj) * kTileRows + 2 * lane]) = pre[jj];
        __syncthreads();
        if (t + gridDim.x < ntiles)
            vq_fetch<kMaxJ>(pre, V, ldv, w, nj, (t + gridDim.x) * kTileRows + 2 * lane, n);

        double2 acc[MAXS];
#pragma unroll
        for (int jj = 0; jj < MAXS; jj++)
        {
            acc[jj].x = 0.0;
            acc[jj].y = 0.0;
            if (accumulate && valid && w + 4 * jj < p)  // a later panel of input columns adds to the earlier ones' result
                acc[jj] = *reinterpret_cast<const double2*>(X + int64_t(w + 4 * jj) * ldx + r);
        }
        for (int j = 0; j < m; j++)
        {
            const double2 v = *reinterpret_cast<const double2*>(&Vt[j * kTileRows + 2 * lane]);
            const double* q = &Qs[(j * 4 + w) * MAXS];
#pragma unroll
            for (int jj = 0; jj < MAXS; jj++)
            {
                acc[jj].x += v.x * q[jj];  // Arnoldi.h:332-334 (dense form; structural zeros of Q multiply to 0)
                acc[jj].y += v.y * q[jj];
            }
        }
        if (valid)
        {
#pragma unroll
            for (int jj = 0; jj < MAXS; jj++)
            {
                const int i = w + 4 * jj;
                if (i < p)  // 2 GB of output per restart: streamed past the caches like the input
                    __builtin_nontemporal_store(v2d{acc[jj].x, acc[jj].y}, reinterpret_cast<v2d*>(X + int64_t(i) * ldx + r));
            }
        }
    }
}

// ---- V*Q with the pending correction of the last one-sweep step riding on it (krylov.hpp launch_vq_fused) ----------------------
// k_vq's tiling (128-row tiles of all m columns staged in LDS, wave w = output columns w, w+4, ...), in place like k_vq (a tile is
// read completely before its rows are written; writing the lines just read measured faster than a second buffer).  Every wave
// additionally forms the corrected residual of its rows from the staged tile (p = V c: m LDS reads, the same for all four
// waves — cheaper than a second barrier), accumulates chk_j = <V_j, f_corr> for the input columns j = w (mod 4), and the wave
// that holds output column kcol writes fnew.  Rows past the end are staged from row 0 (clamped address) and masked out of every sum.
template <int MAXS, int NJ>
__global__ __launch_bounds__(kThreads) void k_vq_fused(const double* V, int64_t ldv, int m, const double* __restrict__ Q, int ldq, int p,
                                                        double* X, int64_t ldx, int64_t n, VqFusedArgs fa)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Vt = smem;                           // [m][128]
    double* Qs = smem + int64_t(m) * kTileRows;  // [m][4][MAXS]
    double* cs = Qs + int64_t(m) * 4 * MAXS;     // [m]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int kMaxJ = NJ;  // input columns per wave: ceil(m / 4) <= NJ
    const int nj = (m - w + 3) / 4;
    for (int idx = tid; idx < m * 4 * MAXS; idx += kThreads)
    {
        const int jj = idx % MAXS, ww = (idx / MAXS) & 3, j = idx / (4 * MAXS);
        const int i = ww + 4 * jj;
        Qs[idx] = (i < p) ? Q[j + int64_t(i) * ldq] : 0.0;
    }
    for (int j = tid; j < m; j += kThreads)
        cs[j] = fa.c[j];
    double chk[kMaxJ];
#pragma unroll
    for (int jj = 0; jj < kMaxJ; jj++)
        chk[jj] = 0.0;
    double b2c = 0.0, b2n = 0.0;
    const int64_t ntiles = (n + kTileRows - 1) / kTileRows;
    v2d pre[kMaxJ];
    int64_t t = blockIdx.x;
    if (t < ntiles)
        vq_fetch<kMaxJ>(pre, V, ldv, w, nj, t * kTileRows + 2 * lane, n);
    for (; t < ntiles; t += gridDim.x)
    {
        const int64_t r = t * kTileRows + 2 * lane;
        const bool valid = r < n;
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < kMaxJ; jj++)
            if (jj < nj)
                *reinterpret_cast<v2d*>(&Vt[(w + 4 * jj) * kTileRows + 2 * lane]) = pre[jj];
        double2 ft;
        ft.x = ft.y = 0.0;
        if (valid)
            ft = *reinterpret_cast<const double2*>(fa.ftilde + r);
        __syncthreads();
        if (t + gridDim.x < ntiles)
            vq_fetch<kMaxJ>(pre, V, ldv, w, nj, (t + gridDim.x) * kTileRows + 2 * lane, n);

        double2 acc[MAXS];
#pragma unroll
        for (int jj = 0; jj < MAXS; jj++)
        {
            acc[jj].x = 0.0;
            acc[jj].y = 0.0;
        }
        double2 pc;  // (V c) of this lane's two rows
        pc.x = pc.y = 0.0;
        for (int j = 0; j < m; j++)
        {
            const double2 v = *reinterpret_cast<const double2*>(&Vt[j * kTileRows + 2 * lane]);
            const double* q = &Qs[(j * 4 + w) * MAXS];
            const double cj = cs[j];
            pc.x += v.x * cj;  // Lanczos.h:171
            pc.y += v.y * cj;
#pragma unroll
            for (int jj = 0; jj < MAXS; jj++)
            {
                acc[jj].x += v.x * q[jj];  // Arnoldi.h:332-334
                acc[jj].y += v.y * q[jj];
            }
        }
        double2 fc;
        fc.x = valid ? ft.x - pc.x : 0.0;
        fc.y = valid ? ft.y - pc.y : 0.0;
#pragma unroll
        for (int jj = 0; jj < kMaxJ; jj++)
            if (jj < nj)
            {
                const double2 v = *reinterpret_cast<const double2*>(&Vt[(w + 4 * jj) * kTileRows + 2 * lane]);
                chk[jj] += v.x * fc.x + v.y * fc.y;  // Lanczos.h:179
            }
        if (w == 0)
            b2c += fc.x * fc.x + fc.y * fc.y;
        if (valid)
        {
#pragma unroll
            for (int jj = 0; jj < MAXS; jj++)
            {
                const int i = w + 4 * jj;
                if (i < p)
                    __builtin_nontemporal_store(v2d{acc[jj].x, acc[jj].y}, reinterpret_cast<v2d*>(X + int64_t(i) * ldx + r));
                if (i == fa.kcol)  // wave-uniform: this wave holds the column that enters the new residual
                {
                    double2 fn;
                    fn.x = fc.x * fa.q_last + acc[jj].x * fa.h_sub;  // Arnoldi.h:337
                    fn.y = fc.y * fa.q_last + acc[jj].y * fa.h_sub;
                    *reinterpret_cast<double2*>(fa.fnew + r) = fn;
                    b2n += fn.x * fn.x + fn.y * fn.y;
                }
            }
        }
    }
    double* rec = fa.partials + blockIdx.x;
#pragma unroll
    for (int jj = 0; jj < kMaxJ; jj++)
    {
        const double sum = wave_reduce_sum(chk[jj]);
        const int j = w + 4 * jj;
        if (lane == 0 && j < m)
            rec[int64_t(j) * fa.pstride] = sum;
    }
    if (w == (fa.kcol & 3))
    {
        const double sum = wave_reduce_sum(b2n);
        if (lane == 0)
            rec[int64_t(m) * fa.pstride] = sum;
    }
    if (w == 0)
    {
        const double sum = wave_reduce_sum(b2c);
        if (lane == 0)
        {
            rec[kSlotBeta2 * fa.pstride] = sum;
            rec[kSlotMaxAbs * fa.pstride] = 0.0;
        }
    }
}

// ---- V*Q on the f64 matrix cores ------------------------------------------------------------------------------
// X' = Q' V' in 16x16x4 MFMA steps (v_mfma_f64_16x16x4_f64): A = Q' (16 output columns x 4 input columns), B = V'
// (4 input columns x 16 rows), D = 16 output columns x 16 rows.  Operand layout (one f64 per lane): A[i][k] and
// B[k][j] live in lane (i or j) + 16 k; D[i][j]: j = lane & 15, i = (lane >> 4) + 4 reg.  So a lane's B operand is
// one element of V — 16 consecutive rows of one column per quarter-wave, straight from HBM, no LDS staging — and
// a store of one D register covers 16 consecutive rows of four columns.  A wave owns 16*NB rows of the tile and
// reads all m columns of them before it writes, which keeps the in-place update (X aliasing V) legal.
typedef double v4d __attribute__((ext_vector_type(4)));
// Lane (j, k) = (lane & 15, lane >> 4) loads the row PAIR (2j, 2j+1) of input column 4kb + k as one 16-byte
// access; the even rows feed one MFMA, the odd rows a second one, and the two D registers of a lane go back as one
// 16-byte store.  A wave therefore covers 32*NB rows per tile with 1 KiB per load instruction.
template <int KB, int MB, int NB>  // ceil(m/4), ceil(p/16) upper bounds; 32-row blocks per wave
__global__ __launch_bounds__(kThreads) void k_vq_mfma(const double* __restrict__ V, int64_t ldv, int m,
                                                       const double* __restrict__ Q, int ldq, int p, double* X, int64_t ldx,
                                                       int64_t n, int accumulate)
{
    extern __shared__ __attribute__((aligned(16))) double qfrag[];  // [kb][mb][64]: A fragments of Q'
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kbn = (m + 3) / 4, mbn = (p + 15) / 16;
    for (int idx = tid; idx < kbn * mbn * 64; idx += kThreads)
    {
        const int l = idx & 63, mb = (idx >> 6) % mbn, kb = (idx >> 6) / mbn;
        const int k = 4 * kb + (l >> 4), i = 16 * mb + (l & 15);
        qfrag[idx] = (k < m && i < p) ? Q[k + int64_t(i) * ldq] : 0.0;
    }
    __syncthreads();
    constexpr int kRowsPerWave = 32 * NB;
    constexpr int kRowsPerBlock = 4 * kRowsPerWave;
    const int64_t ntiles = (n + kRowsPerBlock - 1) / kRowsPerBlock;
    const int jrow = lane & 15, kq = lane >> 4;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x)
    {
        const int64_t row0 = t * kRowsPerBlock + int64_t(w) * kRowsPerWave;
        v2d b[KB][NB];
#pragma unroll
        for (int kb = 0; kb < KB; kb++)
#pragma unroll
            for (int nb = 0; nb < NB; nb++)
            {
                const int col = 4 * kb + kq;
                const int64_t r = row0 + 32 * nb + 2 * jrow;  // rows come in even pairs (vectors are padded to an even length)
                const bool ok = (kb < kbn) && (col < m) && (r < n);
                const v2d v = *reinterpret_cast<const v2d*>(V + int64_t(ok ? col : 0) * ldv + (ok ? r : 0));
                b[kb][nb] = ok ? v : v2d{0.0, 0.0};
            }
#pragma unroll
        for (int mb = 0; mb < MB; mb++)
        {
            if (mb >= mbn)  // block-uniform; no `break`, so that the loop unrolls and b[][] stays in registers
                continue;
            v4d even[NB], odd[NB];
#pragma unroll
            for (int nb = 0; nb < NB; nb++)
            {
                even[nb] = v4d{0.0, 0.0, 0.0, 0.0};
                odd[nb] = v4d{0.0, 0.0, 0.0, 0.0};
                if (accumulate)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++)
                    {
                        const int i = 16 * mb + kq + 4 * reg;
                        const int64_t r = row0 + 32 * nb + 2 * jrow;
                        if (i < p && r < n)
                        {
                            const v2d x = *reinterpret_cast<const v2d*>(X + int64_t(i) * ldx + r);
                            even[nb][reg] = x.x;
                            odd[nb][reg] = x.y;
                        }
                    }
            }
#pragma unroll
            for (int kb = 0; kb < KB; kb++)
            {
                if (kb >= kbn)
                    continue;
                const double a = qfrag[(kb * mbn + mb) * 64 + lane];
#pragma unroll
                for (int nb = 0; nb < NB; nb++)
                {
                    even[nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[kb][nb].x, even[nb], 0, 0, 0);
                    odd[nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[kb][nb].y, odd[nb], 0, 0, 0);
                }
            }
#pragma unroll
            for (int nb = 0; nb < NB; nb++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++)
                {
                    const int i = 16 * mb + kq + 4 * reg;
                    const int64_t r = row0 + 32 * nb + 2 * jrow;
                    if (i < p && r < n)
                        *reinterpret_cast<v2d*>(X + int64_t(i) * ldx + r) = v2d{even[nb][reg], odd[nb][reg]};
                }
        }
    }
}

// ---- SimpleRandom stream by jump-ahead (Util/SimpleRandom.h:30-52, :56-66, :92-96) -----------------
__device__ __forceinline__ uint64_t mod_m31(uint64_t x)  // x < 2^62  ->  x mod (2^31 - 1)
{
    const uint64_t p = 2147483647ULL;
    x = (x & p) + (x >> 31);
    x = (x & p) + (x >> 31);
    return x >= p ? x - p : x;
}
__global__ __launch_bounds__(kThreads) void k_simple_random(double* __restrict__ v, int64_t row_begin, int64_t nloc,
                                                             uint64_t seed)
{
    constexpr int kRun = 16;
    const int64_t first = (int64_t(blockIdx.x) * kThreads + threadIdx.x) * kRun;
    if (first >= nloc)
        return;
    // state after k steps = s0 * 16807^k mod p ; element g of the stream is the state after g+1 steps
    uint64_t s0 = seed ? (seed & 2147483647ULL) : 1ULL;
    uint64_t e = uint64_t(row_begin + first);  // steps already taken before this run
    uint64_t base = 16807ULL, pw = 1ULL;
    while (e)
    {
        if (e & 1)
            pw = mod_m31(pw * base);
        base = mod_m31(base * base);
        e >>= 1;
    }
    uint64_t s = mod_m31(s0 * pw);
    const int64_t last = (first + kRun < nloc) ? first + kRun : nloc;
    if (s0 == 2147483647ULL)  // a fixed point of the reference's folded product (the stream is the constant 0.5)
    {
        for (int64_t i = first; i < last; i++)
            v[i] = double(int64_t(s0)) / double(2147483647L) - 0.5;
        return;
    }
    for (int64_t i = first; i < last; i++)
    {
        s = mod_m31(s * 16807ULL);
        v[i] = double(int64_t(s)) / double(2147483647L) - 0.5;
    }
}

int persistent_grid(const mispec_ctx& ctx, int64_t work_items, int per_cu)
{
    int64_t g = int64_t(ctx.num_cu) * per_cu;
    if (g > work_items)
        g = work_items;
    if (g < 1)
        g = 1;
    return int(g);
}

template <int MODE>
void launch_orth_mode(const mispec_ctx& ctx, const OrthArgs& a, int grid)
{
    // One instantiation per slot count (columns per wave, ceil(ncol / 4)): a wave then issues exactly the
    // loads it needs — with a coarser set of sizes the surplus slots re-read column 0 and spend L1/L2 bandwidth.
    const int slots = (a.ncol + 3) / 4;
    const bool two = slots <= 10;
    const dim3 g(static_cast<unsigned>(grid)), b(kThreads);
    switch (slots)
    {
#define MISPEC_ORTH_CASE(S)                                                      \
    case S:                                                                      \
        if (two)                                                                 \
            hipLaunchKernelGGL((k_orth<MODE, S, 2>), g, b, 0, ctx.stream, a);    \
        else                                                                     \
            hipLaunchKernelGGL((k_orth<MODE, S, 1>), g, b, 0, ctx.stream, a);    \
        break;
        case 0:
            MISPEC_ORTH_CASE(1)
            MISPEC_ORTH_CASE(2)
            MISPEC_ORTH_CASE(3)
            MISPEC_ORTH_CASE(4)
            MISPEC_ORTH_CASE(5)
            MISPEC_ORTH_CASE(6)
            MISPEC_ORTH_CASE(7)
            MISPEC_ORTH_CASE(8)
            MISPEC_ORTH_CASE(9)
            MISPEC_ORTH_CASE(10)
            MISPEC_ORTH_CASE(11)
            MISPEC_ORTH_CASE(12)
            MISPEC_ORTH_CASE(13)
            MISPEC_ORTH_CASE(14)
            MISPEC_ORTH_CASE(15)
        default:
            hipLaunchKernelGGL((k_orth<MODE, 16, 1>), g, b, 0, ctx.stream, a);
            break;
#undef MISPEC_ORTH_CASE
    }
}

void launch_orth_lagged(const mispec_ctx& ctx, const OrthArgs& a, int grid)
{
    const dim3 g(static_cast<unsigned>(grid));
    if (a.ncol >= kPanelCols)  // 64..127 finished columns: eight wavefronts of 16 columns
    {
        const dim3 b8(512);
        switch ((a.ncol + 7) / 8)
        {
#define MISPEC_LAG_CASE8(S)                                                     \
    case S:                                                                     \
        if (a.onered)                                                           \
            hipLaunchKernelGGL((k_orth_lagged<S, 1, 8, true>), g, b8, 0, ctx.stream, a); \
        else                                                                    \
            hipLaunchKernelGGL((k_orth_lagged<S, 1, 8>), g, b8, 0, ctx.stream, a); \
        break;
            MISPEC_LAG_CASE8(8)
            MISPEC_LAG_CASE8(9)
            MISPEC_LAG_CASE8(10)
            MISPEC_LAG_CASE8(11)
            MISPEC_LAG_CASE8(12)
            MISPEC_LAG_CASE8(13)
            MISPEC_LAG_CASE8(14)
            MISPEC_LAG_CASE8(15)
            default:
                if (a.onered)
                    hipLaunchKernelGGL((k_orth_lagged<16, 1, 8, true>), g, b8, 0, ctx.stream, a);
                else
                    hipLaunchKernelGGL((k_orth_lagged<16, 1, 8>), g, b8, 0, ctx.stream, a);
                break;
#undef MISPEC_LAG_CASE8
        }
        return;
    }
    const int slots = (a.ncol + 3) / 4;
    const bool two = slots <= 10;
    const dim3 b(kThreads);
    switch (slots)
    {
#define MISPEC_LAG_CASE(S)                                                  \
    case S:                                                                 \
        if (a.onered && two)                                                \
            hipLaunchKernelGGL((k_orth_lagged<S, 2, 4, true>), g, b, 0, ctx.stream, a); \
        else if (a.onered)                                                  \
            hipLaunchKernelGGL((k_orth_lagged<S, 1, 4, true>), g, b, 0, ctx.stream, a); \
        else if (two)                                                       \
            hipLaunchKernelGGL((k_orth_lagged<S, 2, 4>), g, b, 0, ctx.stream, a); \
        else                                                                \
            hipLaunchKernelGGL((k_orth_lagged<S, 1, 4>), g, b, 0, ctx.stream, a); \
        break;
        case 0:
            MISPEC_LAG_CASE(1)
            MISPEC_LAG_CASE(2)
            MISPEC_LAG_CASE(3)
            MISPEC_LAG_CASE(4)
            MISPEC_LAG_CASE(5)
            MISPEC_LAG_CASE(6)
            MISPEC_LAG_CASE(7)
            MISPEC_LAG_CASE(8)
            MISPEC_LAG_CASE(9)
            MISPEC_LAG_CASE(10)
            MISPEC_LAG_CASE(11)
            MISPEC_LAG_CASE(12)
            MISPEC_LAG_CASE(13)
            MISPEC_LAG_CASE(14)
            MISPEC_LAG_CASE(15)
        default:
            if (a.onered)
                hipLaunchKernelGGL((k_orth_lagged<16, 1, 4, true>), g, b, 0, ctx.stream, a);
            else
                hipLaunchKernelGGL((k_orth_lagged<16, 1, 4>), g, b, 0, ctx.stream, a);
            break;
#undef MISPEC_LAG_CASE
    }
}

}  // namespace

namespace mispec {

namespace {
constexpr bool kOrthDmaDefault = true;  // round 6: -3.5 % per C2 solve, -4...7 % per operation at 1.25 M rows (profiles/r11d)
int orth_tile_rows(int ncol)
{
    return kTileRows * (((ncol + 3) / 4 <= 10) ? 2 : 1);
}

// one launch over at most kPanelCols columns; grid == 0: choose it from the tile count
int launch_orth_panel(const mispec_ctx& ctx, OrthMode mode, const OrthArgs& a, int grid)
{
    // option orth_kernel = dma | dma2 | reg: the LDS-DMA ring version of the one-sweep pass (orth_dma.hip; three or two ring slots)
    // or the register version below; unset: the LDS-DMA version on vectors of at least 1024 tiles of 128 rows
    if (mode == ORTH_LAGGED && grid == 0 && orth_lagged_dma_eligible(a))
    {
        const char* k = option("orth_kernel");
        int depth = 0, flags = 0;
        if (k)
        {
            const std::string ks(k);  // dma, dma2 (two slots), dmac (contiguous tile runs), dmap (plain loads), dmacp
            depth = ks.rfind("dma", 0) == 0 ? (ks == "dma2" ? 2 : 3) : 0;
            if (depth == 3 && ks.size() > 3)
                flags = (ks.find('c', 3) != std::string::npos ? 1 : 0) | (ks.find('p', 3) != std::string::npos ? 2 : 0);
        }
        else if (kOrthDmaDefault && a.n >= int64_t(1024) * 128)
            depth = 3;
        if (depth)
            return launch_orth_lagged_dma(ctx, a, depth, flags);
    }
    // the passes of the reference flow and of the Arnoldi process through the same LDS ring (orth_dma_modes.hip; option
    // orth_kernel=reg: the register kernels below for every mode)
    if (mode != ORTH_LAGGED && grid == 0 && kOrthDmaDefault && orth_dma_modes_eligible(mode, a) && !option_is("orth_kernel", "reg"))
        return launch_orth_dma_mode(ctx, mode, a);
    if (grid == 0)
    {
        const int rows = orth_tile_rows(a.ncol);
        const int64_t ntiles = (a.n + rows - 1) / rows;
        // 3 and 6 workgroups per CU measured slower (profiles/r05b_ab_*); the 512-thread one-sweep kernel of wide bases: 2 (1 the
        // same, 3 slower: profiles/r07aa)
        grid = persistent_grid(ctx, ntiles, (mode == ORTH_LAGGED && a.ncol >= kPanelCols) ? 2 : 4);
    }
    MISPEC_REQUIRE(a.pstride >= grid, "orth kernel: partial-record stride smaller than the grid");
    switch (mode)
    {
        case ORTH_VTF:
            launch_orth_mode<ORTH_VTF>(ctx, a, grid);
            break;
        case ORTH_RESID_VTF:
            launch_orth_mode<ORTH_RESID_VTF>(ctx, a, grid);
            break;
        case ORTH_CORRECT_VTF:
            launch_orth_mode<ORTH_CORRECT_VTF>(ctx, a, grid);
            break;
        case ORTH_CORRECT_ONLY:
            launch_orth_mode<ORTH_CORRECT_ONLY>(ctx, a, grid);
            break;
        case ORTH_LAGGED:
            launch_orth_lagged(ctx, a, grid);
            break;
    }
    MISPEC_HIP(hipGetLastError());
    return grid;
}
}  // namespace

// Up to kPanelCols columns: one launch.  Wider bases (ncv up to kMaxCols) are processed in column panels that all
// write into the same partial records (slot = global column index), so one reduction serves the whole step:
//   VTF          every panel computes its part of c = V'x; the first one also |x|^2
//   RESID_VTF    panel 0 forms f = w - alpha v_i, |f|^2 and its part of V'f; the others V'f on the finished f
//   CORRECT_*    dst = src - V c needs every panel before V'dst: the panels subtract one after the other, the last
//                one (where dst is final) also computes |dst|^2 and its part of V'dst, then the other panels' parts
//                follow on the finished dst.  V is read 1.5 times instead of once — the price of ncv > 64.
int launch_orth(const mispec_ctx& ctx, OrthMode mode, const OrthArgs& a)
{
    MISPEC_REQUIRE(a.ncol >= 0 && a.ncol <= kMaxCols, "orth kernel: more than 1024 basis columns");
    MISPEC_REQUIRE(mode != ORTH_LAGGED || (a.ncol >= 1 && a.ncol < 2 * kPanelCols), "one-sweep orth kernel: needs 1 <= columns <= 127");
    if (a.ncol <= kPanelCols || mode == ORTH_LAGGED)
    {
        OrthArgs one = a;
        one.col0 = 0;
        one.norms = 1;
        return launch_orth_panel(ctx, mode, one, 0);
    }
    const int npan = (a.ncol + kPanelCols - 1) / kPanelCols;
    // all panels must leave the same number of records: one grid for all, valid for the coarsest tiling
    const int64_t ntiles = (a.n + 2 * kTileRows - 1) / (2 * kTileRows);
    const int grid = persistent_grid(ctx, ntiles, 4);
    auto panel = [&](int q) {
        OrthArgs b = a;
        b.col0 = q * kPanelCols;
        b.ncol = std::min(kPanelCols, a.ncol - b.col0);
        b.norms = 0;
        return b;
    };
    switch (mode)
    {
        case ORTH_VTF:
            for (int q = 0; q < npan; q++)
            {
                OrthArgs b = panel(q);
                b.norms = (q == 0);
                launch_orth_panel(ctx, ORTH_VTF, b, grid);
            }
            break;
        case ORTH_RESID_VTF:
            for (int q = 0; q < npan; q++)
            {
                OrthArgs b = panel(q);
                if (q == 0)
                    b.norms = 1;
                else
                    b.src = a.dst;  // the finished f
                launch_orth_panel(ctx, q == 0 ? ORTH_RESID_VTF : ORTH_VTF, b, grid);
            }
            break;
        case ORTH_CORRECT_VTF:
        case ORTH_CORRECT_ONLY:
            for (int q = 0; q < npan; q++)
            {
                OrthArgs b = panel(q);
                if (q > 0)
                    b.src = a.dst;
                const bool last = (q == npan - 1);
                b.norms = last;
                launch_orth_panel(ctx, (last && mode == ORTH_CORRECT_VTF) ? ORTH_CORRECT_VTF : ORTH_CORRECT_ONLY, b, grid);
            }
            if (mode == ORTH_CORRECT_VTF)
                for (int q = 0; q + 1 < npan; q++)
                {
                    OrthArgs b = panel(q);
                    b.src = a.dst;
                    launch_orth_panel(ctx, ORTH_VTF, b, grid);
                }
            break;
        case ORTH_LAGGED:  // single panel only (checked above)
            break;
    }
    return grid;
}

void launch_reduce_partials(const mispec_ctx& ctx, const double* partials, int64_t pstride, int nrec, int ncol, double* red,
                            const FinishArgs& fin)
{
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, ctx.stream, partials, pstride, nrec, ncol, red, fin);
    MISPEC_HIP(hipGetLastError());
}

void launch_publish_state(const mispec_ctx& ctx, const StepState* st, int m, StepState* host_dst, unsigned long long* host_flag,
                          unsigned long long seq)
{
    hipLaunchKernelGGL(k_publish_state, dim3(1), dim3(256), 0, ctx.stream, st, m, host_dst, host_flag, seq);
    MISPEC_HIP(hipGetLastError());
}

void launch_fetch_restart(const mispec_ctx& ctx, const double* host_src, int m, double* Qdev, StepState* st, int with_state)
{
    hipLaunchKernelGGL(k_fetch_restart, dim3(unsigned((m * m + 255) / 256)), dim3(256), 0, ctx.stream, host_src, m, Qdev, st,
                       with_state);
    MISPEC_HIP(hipGetLastError());
}

void launch_finish(const mispec_ctx& ctx, const double* stage, double* red, int ncol, const FinishArgs& fin)
{
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), 0, ctx.stream, stage, red, ncol, fin);
    MISPEC_HIP(hipGetLastError());
}

void launch_scale_step(const mispec_ctx& ctx, const double* src, double* dst, int64_t npad, StepState* st, int step,
                       double eps_sqrt)
{
    const int64_t npairs = npad / 2;
    const int grid = persistent_grid(ctx, (npairs + kThreads - 1) / kThreads, 8);
    hipLaunchKernelGGL(k_scale_step, dim3(unsigned(grid)), dim3(kThreads), 0, ctx.stream, src, dst, npairs, st, step, eps_sqrt);
    MISPEC_HIP(hipGetLastError());
}

void launch_reduce_sum(const mispec_ctx& ctx, const double* in, int64_t count, double* out)
{
    hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(1024), 0, ctx.stream, in, count, out);
    MISPEC_HIP(hipGetLastError());
}

void launch_scale(const mispec_ctx& ctx, const double* src, double* dst, int64_t npad, double divisor)
{
    const int64_t npairs = npad / 2;
    if (npairs == 0)
        return;
    const int grid = persistent_grid(ctx, (npairs + kThreads - 1) / kThreads, 8);
    hipLaunchKernelGGL(k_scale, dim3(unsigned(grid)), dim3(kThreads), 0, ctx.stream, src, dst, npairs, divisor);
    MISPEC_HIP(hipGetLastError());
}

int launch_axpby(const mispec_ctx& ctx, double* f, double a, const double* v, double b, int64_t n, double* partials,
                 int64_t pstride)
{
    const int64_t npairs = (n + 1) / 2;
    const int grid = persistent_grid(ctx, (npairs + kThreads - 1) / kThreads, 4);
    hipLaunchKernelGGL(k_axpby, dim3(unsigned(grid)), dim3(kThreads), 0, ctx.stream, f, a, v, b, npairs, partials, pstride);
    MISPEC_HIP(hipGetLastError());
    return grid;
}

int launch_resid_norms(const mispec_ctx& ctx, const double* y, const double* x, double lambda, int64_t n, double* partials,
                       int64_t pstride)
{
    const int64_t npairs = (n + 1) / 2;
    const int grid = persistent_grid(ctx, (npairs + kThreads - 1) / kThreads, 4);
    hipLaunchKernelGGL(k_resid_norms, dim3(unsigned(grid)), dim3(kThreads), 0, ctx.stream, y, x, lambda, npairs, partials, pstride);
    MISPEC_HIP(hipGetLastError());
    return grid;
}

int launch_resid_norms_complex(const mispec_ctx& ctx, const double* yr, const double* yi, const double* xr, const double* xi,
                               double a, double b, int64_t n, double* partials, int64_t pstride)
{
    const int64_t npairs = (n + 1) / 2;
    const int grid = persistent_grid(ctx, (npairs + kThreads - 1) / kThreads, 4);
    hipLaunchKernelGGL(k_resid_norms_complex, dim3(unsigned(grid)), dim3(kThreads), 0, ctx.stream, yr, yi, xr, xi, a, b, npairs,
                       partials, pstride);
    MISPEC_HIP(hipGetLastError());
    return grid;
}

namespace {
// one launch: at most kPanelCols input and output columns
void launch_vq_panel(const mispec_ctx& ctx, const double* V, int64_t ldv, int m, const double* Q, int ldq, int p, double* X,
                     int64_t ldx, int64_t n, int accumulate)
{
    const bool use_mfma = option_is("vq", "mfma");
    if (use_mfma)
    {
        constexpr int NB = 2;
        const int64_t ntiles_m = (n + 4 * 32 * NB - 1) / (4 * 32 * NB);
        const int grid_m = persistent_grid(ctx, ntiles_m, 4);
        const size_t lds_m = size_t((m + 3) / 4) * size_t((p + 15) / 16) * 64 * sizeof(double);
        const dim3 gm(static_cast<unsigned>(grid_m)), bm(kThreads);
#define MISPEC_VQM(KB, MB)                                                                                                  \
    hipLaunchKernelGGL((k_vq_mfma<KB, MB, NB>), gm, bm, lds_m, ctx.stream, V, ldv, m, Q, ldq, p, X, ldx, n, accumulate)
        const int kbn = (m + 3) / 4, mbn = (p + 15) / 16;
        if (kbn <= 8 && mbn <= 2)
            MISPEC_VQM(8, 2);
        else if (kbn <= 12 && mbn <= 2)
            MISPEC_VQM(12, 2);
        else if (kbn <= 12)
            MISPEC_VQM(12, 4);
        else
            MISPEC_VQM(16, 4);
#undef MISPEC_VQM
        MISPEC_HIP(hipGetLastError());
        return;
    }
    const int64_t ntiles = (n + kTileRows - 1) / kTileRows;
    const int slots = (p + 3) / 4;
    const int maxs = slots <= 4 ? 4 : slots <= 8 ? 8 : slots <= 12 ? 12 : 16;
    const size_t lds = (size_t(m) * kTileRows + size_t(m) * 4 * maxs) * sizeof(double);
    const int grid = persistent_grid(ctx, ntiles, 3);
    const dim3 g(static_cast<unsigned>(grid)), b(kThreads);
#define MISPEC_VQ(S)                                                                                                   \
    do                                                                                                                 \
    {                                                                                                                  \
        MISPEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_vq<S>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       int(lds)));                                                                     \
        hipLaunchKernelGGL((k_vq<S>), g, b, lds, ctx.stream, V, ldv, m, Q, ldq, p, X, ldx, n, accumulate);             \
    } while (0)
    if (maxs == 4)
        MISPEC_VQ(4);
    else if (maxs == 8)
        MISPEC_VQ(8);
    else if (maxs == 12)
        MISPEC_VQ(12);
    else
        MISPEC_VQ(16);
#undef MISPEC_VQ
    MISPEC_HIP(hipGetLastError());
}
}  // namespace

// X = V Q.  Up to kPanelCols x kPanelCols: one launch, X may alias V.  Wider: panels of output columns, each
// accumulated over panels of input columns (X must not alias V then; V is read once per output panel).
void launch_vq(const mispec_ctx& ctx, const double* V, int64_t ldv, int m, const double* Q, int ldq, int p, double* X,
               int64_t ldx, int64_t n)
{
    MISPEC_REQUIRE(m >= 1 && m <= kMaxCols && p >= 1 && p <= kMaxCols, "V*Q kernel: needs 1 <= m, p <= 1024");
    if (m <= kPanelCols && p <= kPanelCols)
    {
        launch_vq_panel(ctx, V, ldv, m, Q, ldq, p, X, ldx, n, 0);
        return;
    }
    MISPEC_REQUIRE(X != V, "V*Q kernel: in-place products need m, p <= 64");
    for (int p0 = 0; p0 < p; p0 += kPanelCols)
        for (int m0 = 0; m0 < m; m0 += kPanelCols)
            launch_vq_panel(ctx, V + int64_t(m0) * ldv, ldv, std::min(kPanelCols, m - m0), Q + m0 + int64_t(p0) * ldq, ldq,
                            std::min(kPanelCols, p - p0), X + int64_t(p0) * ldx, ldx, n, m0 > 0);
}

int launch_vq_fused(const mispec_ctx& ctx, const double* V, int64_t ldv, int m, const double* Q, int ldq, int p, double* X, int64_t ldx,
                    int64_t n, const VqFusedArgs& fa)
{
    MISPEC_REQUIRE(m >= 1 && m <= kPanelCols && p >= 1 && p <= kPanelCols && fa.kcol >= 0 && fa.kcol < p && fa.fnew != fa.ftilde,
                   "fused V*Q kernel: needs 1 <= m, p <= 64, kcol < p and a residual buffer of its own");
    const int64_t ntiles = (n + kTileRows - 1) / kTileRows;
    const int slots = (p + 3) / 4;
    const int maxs = slots <= 4 ? 4 : slots <= 8 ? 8 : slots <= 12 ? 12 : 16;
    const size_t lds = (size_t(m) * kTileRows + size_t(m) * 4 * maxs + size_t(m)) * sizeof(double);
    const int grid = persistent_grid(ctx, ntiles, 3);
    MISPEC_REQUIRE(fa.pstride >= grid, "fused V*Q kernel: partial-record stride smaller than the grid");
    const dim3 g(static_cast<unsigned>(grid)), b(kThreads);
#define MISPEC_VQF(S, J)                                                                                                       \
    do                                                                                                                         \
    {                                                                                                                          \
        MISPEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_vq_fused<S, J>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       int(lds)));                                                                             \
        hipLaunchKernelGGL((k_vq_fused<S, J>), g, b, lds, ctx.stream, V, ldv, m, Q, ldq, p, X, ldx, n, fa);                    \
    } while (0)
    // the benchmark's shape (ncv = 40: ten input columns per wavefront, <= 32 output columns) has an instantiation of its own:
    // 16 column slots cost 238 registers = 2 workgroups per CU, ten fit 3
    if (maxs == 8 && m <= 40)
        MISPEC_VQF(8, 10);
    else if (maxs == 4)
        MISPEC_VQF(4, 16);
    else if (maxs == 8)
        MISPEC_VQF(8, 16);
    else if (maxs == 12)
        MISPEC_VQF(12, 16);
    else
        MISPEC_VQF(16, 16);
#undef MISPEC_VQF
    MISPEC_HIP(hipGetLastError());
    return grid;
}

int lanczos_epilogue_records(const mispec_ctx& ctx, int64_t n)
{
    const int64_t npairs = (n + 1) / 2;
    return persistent_grid(ctx, (npairs + kThreads - 1) / kThreads, 4);
}

void launch_lanczos_epilogue(const mispec_ctx& ctx, double* w, const double* v, const double* v_prev, double h_prev, int64_t n,
                             double* partials, const double* h_prev_dev, const int* status)
{
    const int64_t npairs = (n + 1) / 2;
    const int grid = lanczos_epilogue_records(ctx, n);
    hipLaunchKernelGGL(k_lanczos_epilogue, dim3(unsigned(grid)), dim3(kThreads), 0, ctx.stream, w, v, v_prev, h_prev, npairs,
                       partials, h_prev_dev, status);
    MISPEC_HIP(hipGetLastError());
}

void launch_simple_random(const mispec_ctx& ctx, double* v, int64_t row_begin, int64_t nloc, uint64_t seed)
{
    if (nloc == 0)
        return;
    const int64_t threads = (nloc + 15) / 16;
    hipLaunchKernelGGL(k_simple_random, dim3(unsigned((threads + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx.stream, v,
                       row_begin, nloc, seed);
    MISPEC_HIP(hipGetLastError());
}

}  // namespace mispec
