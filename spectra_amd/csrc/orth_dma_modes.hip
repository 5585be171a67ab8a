// The passes over the basis of the REFERENCE control flow (Lanczos.h:145-152, :171-179) and of the Arnoldi process (Arnoldi.h:
// 251-255) — k_orth's modes ORTH_VTF, ORTH_RESID_VTF, ORTH_CORRECT_VTF, ORTH_CORRECT_ONLY (krylov.hip) — with the basis streamed
// through an LDS ring by LDS-DMA, as orth_dma.hip does for the one-sweep pass (round 6; the design, the ordering rules and the
// counter evidence are in that file's header: one 256-thread workgroup per CU, every wavefront keeps DEPTH - 1 tiles of its own
// columns in flight, counted s_waitcnt vmcnt(N), a bare s_barrier per tile, a ring slot refilled only behind the barrier that
// follows its last reader).  One column panel (<= 64 columns), vectors of at least 1024 tiles of 128 rows; everything else
// keeps k_orth.  Same expressions per element and the same record layout as k_orth; the partial sums are grouped by 256
// workgroups x 128-row tiles instead of 1024 x 256 rows (a different grid of the same kernel would differ by the same rounding).
//   VTF          c = V'x (+ |x|^2, max |x|)                       extra DMA column: x
//   RESID_VTF    f = w - alpha v_i -> dst ; c = V'f ; norms       extra DMA columns: w (wave 0), v_i (wave 1)
//   CORRECT_*    dst = src - V c_in ; norms ; (_VTF) c = V'dst    extra DMA column: src
#include "krylov.hpp"

using namespace mispec;

namespace {

constexpr int kThreads = 256;
constexpr int kNW = 4;
constexpr int kRows = 128;
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

template <int MODE, int MAXS, int DEPTH>
__global__ __launch_bounds__(kThreads, 1) void k_orth_dma(OrthArgs a)
{
    constexpr bool kCorrect = (MODE == ORTH_CORRECT_VTF || MODE == ORTH_CORRECT_ONLY);
    constexpr bool kVtf = (MODE != ORTH_CORRECT_ONLY);
    constexpr int NL = MAXS + 1;  // DMA instructions per wavefront and tile: its columns + one vector
    constexpr int kRegion = kRows;
    constexpr int kSlotDoubles = kNW * NL * kRegion;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* ring = reinterpret_cast<double*>(smem);  // [DEPTH][kNW][NL][128]
    double* psum = ring + DEPTH * kSlotDoubles;      // [2][kNW][128]
    double* cs = psum + 2 * kNW * kRows;             // [64]

    if (a.status && *a.status != kStepOk)
        return;
    if (a.need_corr && *a.need_corr == 0)
        return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    double alpha = 0.0;
    if (MODE == ORTH_RESID_VTF)
        alpha = *a.alpha_dev;
    if (tid < 64)
        cs[tid] = (kCorrect && tid < a.ncol) ? a.c_in[a.col0 + tid] : 0.0;
    __syncthreads();  // (also retires every compiler-visible load above)

    const double* colp[MAXS];
    double cw[MAXS], acc[MAXS];
#pragma unroll
    for (int jj = 0; jj < MAXS; jj++)
    {
        const int j = w + kNW * jj;
        colp[jj] = a.V + int64_t(a.col0 + (j < a.ncol ? j : 0)) * a.ldv;  // surplus slots re-read the panel's first column, coefficient 0
        cw[jj] = cs[j];
        acc[jj] = 0.0;
    }
    // wave 0: the input vector; wave 1: v_i (RESID) — the others repeat the input vector (equal DMA counts per wavefront)
    const double* extra = (MODE == ORTH_RESID_VTF && (w & 1)) ? a.vi : a.src;
    double b2 = 0.0, mx = 0.0;

    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_char*) smem));
    const int64_t ntiles = (a.n + kRows - 1) / kRows;
    const int64_t G = gridDim.x;

    auto issue = [&](int64_t t, int slot) {
        const int64_t r = t * kRows + 2 * lane;
        const int64_t rc = r < a.n ? r : 0;
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + unsigned((slot * kNW + w) * NL) * 1024u);
#pragma unroll
        for (int jj = 0; jj < MAXS; jj++)
            dma16<true>(colp[jj] + rc, base + unsigned(jj) * 1024u);
        dma16<false>(extra + rc, base + unsigned(MAXS) * 1024u);
    };

    int64_t t = blockIdx.x;
    if (t < ntiles)
        issue(t, 0);
    if (DEPTH == 3 && t + G < ntiles)
        issue(t + G, 1);
    int slot = 0, buf = 0;
    for (; t < ntiles; t += G)
    {
        // ---- tile t of this wave's columns has landed ----
        if (DEPTH == 3 && t + G < ntiles)
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NL) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const double* mine = ring + (slot * kNW + w) * NL * kRegion + 2 * lane;
        double2 vv[MAXS];
#pragma unroll
        for (int jj = 0; jj < MAXS; jj++)
            vv[jj] = *reinterpret_cast<const double2*>(mine + jj * kRegion);
        if (kCorrect)
        {
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
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // ---- every wave has left tile t-1's slot: refill it ----
        if (DEPTH == 3)
        {
            if (t + 2 * G < ntiles)
                issue(t + 2 * G, slot == 0 ? 2 : slot - 1);
        }
        else if (t + G < ntiles)
            issue(t + G, slot ^ 1);
        const int64_t r = t * kRows + 2 * lane;
        const bool valid = r < a.n;
        const double* tile = ring + slot * kSlotDoubles + 2 * lane;
        double2 fv = *reinterpret_cast<const double2*>(tile + (0 * NL + MAXS) * kRegion);  // wave 0's vector: src
        if (MODE == ORTH_RESID_VTF)
        {
            const double2 vi = *reinterpret_cast<const double2*>(tile + (1 * NL + MAXS) * kRegion);
            fv.x = fv.x - alpha * vi.x;  // Lanczos.h:145
            fv.y = fv.y - alpha * vi.y;
        }
        if (!valid)
        {
            fv.x = 0.0;
            fv.y = 0.0;
        }
        if (kCorrect)
        {
            const double* ps = psum + buf * kNW * kRows + 2 * lane;
            const double2 p0 = *reinterpret_cast<const double2*>(ps);
            const double2 p1 = *reinterpret_cast<const double2*>(ps + kRows);
            const double2 p2 = *reinterpret_cast<const double2*>(ps + 2 * kRows);
            const double2 p3 = *reinterpret_cast<const double2*>(ps + 3 * kRows);
            if (valid)
            {
                fv.x -= (p0.x + p1.x) + (p2.x + p3.x);  // Lanczos.h:171 / Arnoldi.h:254
                fv.y -= (p0.y + p1.y) + (p2.y + p3.y);
            }
            buf ^= 1;
        }
        if (w == 0)
        {
            if (MODE != ORTH_VTF && valid)
                *reinterpret_cast<double2*>(a.dst + r) = fv;
            b2 += fv.x * fv.x + fv.y * fv.y;
            mx = fmax(mx, fmax(fabs(fv.x), fabs(fv.y)));
        }
        if (kVtf)
        {
#pragma unroll
            for (int jj = 0; jj < MAXS; jj++)
                acc[jj] += vv[jj].x * fv.x + vv[jj].y * fv.y;
        }
        slot = (DEPTH == 3) ? (slot == 2 ? 0 : slot + 1) : (slot ^ 1);
    }

    double* rec = a.partials + blockIdx.x;
    if (kVtf)
    {
#pragma unroll
        for (int jj = 0; jj < MAXS; jj++)
        {
            const double s = wave_sum(acc[jj]);
            const int j = w + kNW * jj;
            if (lane == 0 && j < a.ncol)
                rec[int64_t(a.col0 + j) * a.pstride] = s;
        }
    }
    if (w == 0 && a.norms)
    {
        b2 = wave_sum(b2);
        mx = wave_max(mx);
        if (lane == 0)
        {
            rec[kSlotBeta2 * a.pstride] = b2;
            rec[kSlotMaxAbs * a.pstride] = mx;
        }
    }
}

template <int MODE, int MAXS, int DEPTH>
void launch_inst(const mispec_ctx& ctx, const OrthArgs& a, int grid)
{
    const size_t lds = (size_t(DEPTH) * kNW * (MAXS + 1) * kRows + 2 * kNW * kRows + 64) * sizeof(double);
    static thread_local int attr_dev = -1;
    if (attr_dev != ctx.device)
    {
        MISPEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_orth_dma<MODE, MAXS, DEPTH>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        attr_dev = ctx.device;
    }
    hipLaunchKernelGGL((k_orth_dma<MODE, MAXS, DEPTH>), dim3(unsigned(grid)), dim3(kThreads), lds, ctx.stream, a);
}

// slots (columns per wavefront) in steps of two: half the instantiations; an odd count reads at most four surplus columns
template <int MODE>
void launch_mode(const mispec_ctx& ctx, const OrthArgs& a, int grid)
{
    const int slots = (a.ncol + 3) / 4;
    switch ((slots + 1) / 2)
    {
        case 0:
        case 1:
            launch_inst<MODE, 2, 3>(ctx, a, grid);
            break;
        case 2:
            launch_inst<MODE, 4, 3>(ctx, a, grid);
            break;
        case 3:
            launch_inst<MODE, 6, 3>(ctx, a, grid);
            break;
        case 4:
            launch_inst<MODE, 8, 3>(ctx, a, grid);
            break;
        case 5:
            launch_inst<MODE, 10, 3>(ctx, a, grid);
            break;
        case 6:
            launch_inst<MODE, 12, 2>(ctx, a, grid);
            break;
        case 7:
            launch_inst<MODE, 14, 2>(ctx, a, grid);
            break;
        default:
            launch_inst<MODE, 16, 2>(ctx, a, grid);
            break;
    }
}

}  // namespace

namespace mispec {

bool orth_dma_modes_eligible(OrthMode mode, const OrthArgs& a)
{
    return mode != ORTH_LAGGED && a.ncol >= 1 && a.ncol <= kPanelCols && a.n >= int64_t(1024) * kRows;
}

int launch_orth_dma_mode(const mispec_ctx& ctx, OrthMode mode, const OrthArgs& a)
{
    const int64_t ntiles = (a.n + kRows - 1) / kRows;
    const int grid = int(std::min<int64_t>(ntiles, ctx.num_cu));
    MISPEC_REQUIRE(a.pstride >= grid, "orth kernel (LDS-DMA): partial-record stride smaller than the grid");
    switch (mode)
    {
        case ORTH_VTF:
            launch_mode<ORTH_VTF>(ctx, a, grid);
            break;
        case ORTH_RESID_VTF:
            launch_mode<ORTH_RESID_VTF>(ctx, a, grid);
            break;
        case ORTH_CORRECT_VTF:
            launch_mode<ORTH_CORRECT_VTF>(ctx, a, grid);
            break;
        case ORTH_CORRECT_ONLY:
            launch_mode<ORTH_CORRECT_ONLY>(ctx, a, grid);
            break;
        default:
            throw Error(MISPEC_EINVAL, "orth kernel (LDS-DMA): mode not handled here");
    }
    MISPEC_HIP(hipGetLastError());
    return grid;
}

}  // namespace mispec
