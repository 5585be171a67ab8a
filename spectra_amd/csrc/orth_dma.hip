// The one-sweep orthogonalisation pass (krylov.hip k_orth_lagged, ORTH_LAGGED) with the basis streamed through an LDS ring by
// LDS-DMA (`global_load_lds_dwordx4`) instead of through registers — round 6, VERDICT r05 item 2.
//
// Why: the register version issues a tile's 20 x 16-byte loads per lane, waits for all of them, computes, and only then issues
// the next tile's loads; its SQ / TCC counters (profiles/r11a_counters_c2_round5_tree.txt) show 90 128-byte requests in flight
// per CU on average at 1214 cycles per request — the memory system is asked for 5.8 TB/s, not saturated.  Holding a second
// tile in flight costs 80 more VGPRs per lane in registers and nothing in LDS: here every wavefront keeps DEPTH - 1 tiles of its
// own columns in flight (wave-private regions of a DEPTH-slot ring, 1 KiB per column and tile = one DMA instruction), waits
// with a COUNTED `s_waitcnt vmcnt(N)` — never 0 inside the loop —, and the per-tile barrier of the row-sum exchange is a bare
// `s_barrier` that does not drain VMEM.  One 256-thread workgroup per CU (the ring is 3 x 44 KiB at 40 columns), grid-stride
// over 128-row tiles; 256 partial records instead of 1024, so the record reduction behind the pass reads a quarter.
//
// Ordering rules used (MI355X_MICROARCH.md "Two waves per SIMD" item 7, cdna_hip_programming.md 5.7):
//   * LDS-DMA data is ordered for a ds_read only by the ISSUING wave's covering vmcnt; other waves read it after a barrier the
//     issuer reached behind that wait: a wave reads its own columns after its own wait, and f, u = src and column i-1
//     (regions of waves 0, 1 and (i-1) % 4) after the exchange barrier of the tile;
//   * the asm DMA is invisible to hipcc's waitcnt bookkeeping: every wait for it is an explicit statement; the loop holds no
//     compiler-visible VMEM load, and wave 0's two stores per tile only make the counted wait conservative (vmcnt counts them
//     too; loads complete in order among themselves, so "at most NL outstanding" still means tile t has landed);
//   * a ring slot is refilled only behind the barrier that follows the last cross-wave read of it (slot of tile t-1 behind the
//     barrier of tile t; every wave's LDS reads are retired by the lgkmcnt(0) in front of that barrier).
//
// Arithmetic, record layout and the device-side step decisions are those of k_orth_lagged (same expressions per element; the
// partial sums are grouped by 256 workgroups x 128-row tiles instead of 1024 x 256, so the reduced sums differ from the
// register kernel's by rounding — as they do between any two grid sizes of that kernel).
// Replaces Lanczos.h:171 of step i-1 + :106, :139, :145-152 of step i, as k_orth_lagged does.
#include "krylov.hpp"

using namespace mispec;

namespace {

constexpr int kThreads = 256;
constexpr int kNW = 4;
constexpr int kRows = 128;  // rows of a tile: one 16-byte DMA element per lane and column covers two rows
typedef __attribute__((address_space(3))) char lds_char;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v = fmax(v, __shfl_down(v, off, 64));
    return v;
}

// one wave-wide LDS-DMA: lane l's 16 bytes at gsrc land at LDS byte lds_dst + 16 l (lds_dst wave-uniform)
template <bool NT>
__device__ __forceinline__ void dma16(const double* gsrc, unsigned lds_dst)
{
    unsigned keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(lds_dst)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(lds_dst)
                     : "memory");
}

template <int MAXS, int DEPTH, bool ONERED, bool NTV = true>
__global__ __launch_bounds__(kThreads, 1) void k_orth_lagged_dma(OrthArgs a, int contiguous)
{
    constexpr int NL = MAXS + 1;                  // DMA instructions per wavefront and tile: its columns + one of {f, src}
    constexpr int kRegion = kRows;                // doubles per (wave, column slot) region = 1 KiB
    constexpr int kSlotDoubles = kNW * NL * kRegion;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* ring = reinterpret_cast<double*>(smem);            // [DEPTH][kNW][NL][128]
    double* psum = ring + DEPTH * kSlotDoubles;                 // [2][kNW][128]
    double* cs = psum + 2 * kNW * kRows;                        // [64]

    if (a.status && *a.status != kStepOk)
        return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool pending = *a.pending != 0;
    const double alpha = *a.alpha_dev;
    const double beta = *a.beta_dev;
    if (tid < 64)
        cs[tid] = (pending && tid < a.ncol) ? a.c_in[tid] : 0.0;
    __syncthreads();  // (also retires every compiler-visible load above: nothing of the compiler's is in flight below)

    const double* colp[MAXS];
    double cw[MAXS], acc[MAXS], chk[MAXS];
#pragma unroll
    for (int jj = 0; jj < MAXS; jj++)
    {
        const int j = w + kNW * jj;
        colp[jj] = a.V + int64_t(j < a.ncol ? j : 0) * a.ldv;  // surplus slots re-read column 0 with coefficient 0
        cw[jj] = cs[j];
        acc[jj] = 0.0;
        chk[jj] = 0.0;
    }
    const double* extra = (w & 1) ? a.src : a.vi;  // wave 0: f, wave 1: u (waves 2, 3 repeat them: equal DMA counts per wave)
    double b2 = 0.0, mx = 0.0, dvi = 0.0;

    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_char*) smem));  // LDS byte offset of the ring
    const int64_t ntiles = (a.n + kRows - 1) / kRows;
    const int64_t G = gridDim.x;

    auto issue = [&](int64_t t, int slot) {
        const int64_t r = t * kRows + 2 * lane;
        const int64_t rc = r < a.n ? r : 0;
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + unsigned((slot * kNW + w) * NL) * 1024u);
#pragma unroll
        for (int jj = 0; jj < MAXS; jj++)
            dma16<NTV>(colp[jj] + rc, base + unsigned(jj) * 1024u);
        dma16<false>(extra + rc, base + unsigned(MAXS) * 1024u);
    };

    // tiles of this workgroup: b, b + G, b + 2G, ... (neighbouring workgroups read neighbouring KiB of a column at about the same
    // time) or, `contiguous`, one run of ceil(ntiles / G) tiles.  Below `t` is the tile id, advancing by `step`, ending at `tend`.
    const int64_t per = (ntiles + G - 1) / G;
    const int64_t step = contiguous ? 1 : G;
    int64_t t = contiguous ? int64_t(blockIdx.x) * per : int64_t(blockIdx.x);
    const int64_t tend = contiguous ? (t + per < ntiles ? t + per : ntiles) : ntiles;
    if (t < tend)
        issue(t, 0);
    if (DEPTH == 3 && t + step < tend)
        issue(t + step, 1);
    int slot = 0, buf = 0;
    const int jp = a.ncol - 1;  // ONERED: column i-1 = slot jp / 4 of wavefront jp % 4 (a.ncol = i >= 1)
    for (; t < tend; t += step)
    {
        // ---- tile t of this wave's columns has landed ----
        if (DEPTH == 3 && t + step < tend)
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NL) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const double* mine = ring + (slot * kNW + w) * NL * kRegion + 2 * lane;
        double2 vv[MAXS];
#pragma unroll
        for (int jj = 0; jj < MAXS; jj++)
            vv[jj] = *reinterpret_cast<const double2*>(mine + jj * kRegion);
        double2 p;
        p.x = 0.0;
        p.y = 0.0;
#pragma unroll
        for (int jj = 0; jj < MAXS; jj++)
        {
            p.x += vv[jj].x * cw[jj];
            p.y += vv[jj].y * cw[jj];
        }
        *reinterpret_cast<double2*>(&psum[(buf * kNW + w) * kRows + 2 * lane]) = p;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // ---- every wave has left tile t-1's slot: refill it (DEPTH 3: tile t+2G; DEPTH 2: tile t+G) ----
        if (DEPTH == 3)
        {
            if (t + 2 * step < tend)
                issue(t + 2 * step, slot == 0 ? 2 : slot - 1);
        }
        else if (t + step < tend)
            issue(t + step, slot ^ 1);
        const int64_t r = t * kRows + 2 * lane;
        const bool valid = r < a.n;  // rows come in even pairs; vectors are zero-padded to an even length
        const double* tile = ring + slot * kSlotDoubles + 2 * lane;
        double2 fv = *reinterpret_cast<const double2*>(tile + (0 * NL + MAXS) * kRegion);
        double2 wv = *reinterpret_cast<const double2*>(tile + (1 * NL + MAXS) * kRegion);
        if (!valid)
        {
            fv.x = fv.y = 0.0;
            wv.x = wv.y = 0.0;
        }
        if (ONERED)
        {
            const double2 vp = *reinterpret_cast<const double2*>(tile + ((jp % kNW) * NL + jp / kNW) * kRegion);
            wv.x = wv.x / beta - beta * vp.x;  // Lanczos.h:106 and :139 applied after the product
            wv.y = wv.y / beta - beta * vp.y;
            if (!valid)
            {
                wv.x = 0.0;
                wv.y = 0.0;
            }
        }
        const double* ps = psum + buf * kNW * kRows + 2 * lane;
        const double2 p0 = *reinterpret_cast<const double2*>(ps);
        const double2 p1 = *reinterpret_cast<const double2*>(ps + kRows);
        const double2 p2 = *reinterpret_cast<const double2*>(ps + 2 * kRows);
        const double2 p3 = *reinterpret_cast<const double2*>(ps + 3 * kRows);
        const double px = (p0.x + p1.x) + (p2.x + p3.x), py = (p0.y + p1.y) + (p2.y + p3.y);
        double2 vi, fn;
        vi.x = (fv.x - px) / beta;  // Lanczos.h:171 then :106 (true division)
        vi.y = (fv.y - py) / beta;
        if (!valid)  // rows past the end were loaded from row 0 (clamped address): they must not reach the sums
        {
            vi.x = 0.0;
            vi.y = 0.0;
        }
        fn.x = wv.x - alpha * vi.x;  // Lanczos.h:145
        fn.y = wv.y - alpha * vi.y;
        if (w == 0)
        {
            if (valid)
            {
                *reinterpret_cast<double2*>(a.vout + r) = vi;  // cacheable: the next product reads column i and f
                *reinterpret_cast<double2*>(a.dst + r) = fn;
            }
            b2 += fn.x * fn.x + fn.y * fn.y;
            dvi += vi.x * fn.x + vi.y * fn.y;
            mx = fmax(mx, fmax(fabs(fn.x), fabs(fn.y)));
        }
#pragma unroll
        for (int jj = 0; jj < MAXS; jj++)
        {
            acc[jj] += vv[jj].x * fn.x + vv[jj].y * fn.y;
            chk[jj] += vv[jj].x * vi.x + vv[jj].y * vi.y;
        }
        buf ^= 1;
        slot = (DEPTH == 3) ? (slot == 2 ? 0 : slot + 1) : (slot ^ 1);
    }

    double* rec = a.partials + blockIdx.x;
#pragma unroll
    for (int jj = 0; jj < MAXS; jj++)
    {
        const double s = wave_sum(acc[jj]);
        const double c = wave_sum(chk[jj]);
        const int j = w + kNW * jj;
        if (lane == 0 && j < a.ncol)
        {
            rec[int64_t(j) * a.pstride] = s;
            rec[int64_t(a.ncol + 1 + j) * a.pstride] = c;
        }
    }
    if (w == 0)
    {
        b2 = wave_sum(b2);
        dvi = wave_sum(dvi);
        mx = wave_max(mx);
        if (lane == 0)
        {
            rec[int64_t(a.ncol) * a.pstride] = dvi;
            rec[kSlotBeta2 * a.pstride] = b2;
            rec[kSlotMaxAbs * a.pstride] = mx;
        }
    }
}

template <int MAXS, int DEPTH, bool ONERED, bool NTV>
void launch_inst(const mispec_ctx& ctx, const OrthArgs& a, int grid, int contiguous)
{
    const size_t lds = (size_t(DEPTH) * kNW * (MAXS + 1) * kRows + 2 * kNW * kRows + 64) * sizeof(double);
    // (the attribute is per device and function; one context per device and process in practice, and setting it again is harmless)
    static thread_local int attr_dev = -1;
    if (attr_dev != ctx.device)
    {
        MISPEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_orth_lagged_dma<MAXS, DEPTH, ONERED, NTV>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        attr_dev = ctx.device;
    }
    hipLaunchKernelGGL((k_orth_lagged_dma<MAXS, DEPTH, ONERED, NTV>), dim3(unsigned(grid)), dim3(kThreads), lds, ctx.stream, a,
                       contiguous);
}

// flags: bit 0 contiguous tile runs per workgroup, bit 1 plain (not non-temporal) DMA loads of the basis — A/B switches
template <int MAXS, int DEPTH>
void launch_one(const mispec_ctx& ctx, const OrthArgs& a, int grid, int flags)
{
    const int contiguous = flags & 1;
    if (DEPTH == 3 && (flags & 2))
    {
        if (a.onered)
            launch_inst<MAXS, DEPTH, true, DEPTH != 3>(ctx, a, grid, contiguous);
        else
            launch_inst<MAXS, DEPTH, false, DEPTH != 3>(ctx, a, grid, contiguous);
        return;
    }
    if (a.onered)
        launch_inst<MAXS, DEPTH, true, true>(ctx, a, grid, contiguous);
    else
        launch_inst<MAXS, DEPTH, false, true>(ctx, a, grid, contiguous);
}

}  // namespace

namespace mispec {

// ring slots by columns per wavefront: three while 3 x 4 x (S + 1) KiB + 8.5 KiB fit the CU's 160 KiB (S <= 11), else two
bool orth_lagged_dma_eligible(const OrthArgs& a)
{
    const int slots = (a.ncol + 3) / 4;
    return a.ncol >= 1 && a.ncol < kPanelCols && slots <= 16 && a.n >= 1;
}

int launch_orth_lagged_dma(const mispec_ctx& ctx, const OrthArgs& a, int depth_override, int flags)
{
    const int64_t ntiles = (a.n + kRows - 1) / kRows;
    const int grid = int(std::min<int64_t>(ntiles, ctx.num_cu));  // one workgroup per CU (the ring does not leave room for two)
    MISPEC_REQUIRE(a.pstride >= grid, "one-sweep orth kernel (LDS-DMA): partial-record stride smaller than the grid");
    const int slots = (a.ncol + 3) / 4;
    const bool three = slots <= 11 && depth_override != 2;
    switch (slots)
    {
#define MISPEC_DMA_CASE(S)                       \
    case S:                                      \
        if (three)                               \
            launch_one<S, 3>(ctx, a, grid, flags);      \
        else                                     \
            launch_one<S, 2>(ctx, a, grid, flags);      \
        break;
        MISPEC_DMA_CASE(1)
        MISPEC_DMA_CASE(2)
        MISPEC_DMA_CASE(3)
        MISPEC_DMA_CASE(4)
        MISPEC_DMA_CASE(5)
        MISPEC_DMA_CASE(6)
        MISPEC_DMA_CASE(7)
        MISPEC_DMA_CASE(8)
        MISPEC_DMA_CASE(9)
        MISPEC_DMA_CASE(10)
        MISPEC_DMA_CASE(11)
#undef MISPEC_DMA_CASE
#define MISPEC_DMA_CASE2(S)                  \
    case S:                                  \
        launch_one<S, 2>(ctx, a, grid, flags);      \
        break;
        MISPEC_DMA_CASE2(12)
        MISPEC_DMA_CASE2(13)
        MISPEC_DMA_CASE2(14)
        MISPEC_DMA_CASE2(15)
        MISPEC_DMA_CASE2(16)
#undef MISPEC_DMA_CASE2
        default:
            throw Error(MISPEC_EINVAL, "one-sweep orth kernel (LDS-DMA): needs 1 <= columns <= 63");
    }
    MISPEC_HIP(hipGetLastError());
    return grid;
}

}  // namespace mispec
