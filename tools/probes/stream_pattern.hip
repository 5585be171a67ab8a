// Stand-alone probe (not part of the library): what does the MEMORY PATTERN of the restart's V <- V Q pass cost on this device,
// without its arithmetic?  k_vq / k_vq_fused / k_vq_mfma all run at about 4.9 TB/s; this kernel moves the same
// bytes the same way — 128-row tiles, wave w of a 256-thread workgroup loads columns w, w+4, ... of an ldv-strided column-major
// basis (1 KiB per column and tile), every wave stores a share of the p output columns — and does nothing else.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/stream_pattern.bin tools/probes/stream_pattern.hip
//   tools/probes/stream_pattern.bin [n]          one JSON line per pattern
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v2d __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                  \
    do                                                                            \
    {                                                                             \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess)                                                     \
        {                                                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

// layout 0: column-major basis, column stride ldv (the library's layout).  layout 1: row-blocked: tile t of 128 rows holds its m
// columns contiguously (m KiB), the layout DESIGN.md section 8 names as the untried lever.
template <int NJ, bool NT>
__global__ __launch_bounds__(256) void k_pattern(const double* __restrict__ V, double* __restrict__ X, int64_t ldv, int64_t ldx, int m, int p,
                                                 int64_t n, int layout)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t ntiles = n / 128;
    const int nj = (m - w + 3) / 4, np = (p - w + 3) / 4;
    v2d pre[NJ];
    auto src = [&](int64_t t, int j) { return layout == 0 ? V + int64_t(j) * ldv + t * 128 + 2 * lane : V + (t * m + j) * 128 + 2 * lane; };
    auto dst = [&](int64_t t, int j) { return layout == 0 ? X + int64_t(j) * ldx + t * 128 + 2 * lane : X + (t * m + j) * 128 + 2 * lane; };
    int64_t t = blockIdx.x;
    if (t < ntiles)
    {
#pragma unroll
        for (int jj = 0; jj < NJ; jj++)
            if (jj < nj)
                pre[jj] = NT ? __builtin_nontemporal_load(reinterpret_cast<const v2d*>(src(t, w + 4 * jj))) : *reinterpret_cast<const v2d*>(src(t, w + 4 * jj));
    }
    for (; t < ntiles; t += gridDim.x)
    {
        v2d cur[NJ];
#pragma unroll
        for (int jj = 0; jj < NJ; jj++)
            cur[jj] = pre[jj];
        if (t + gridDim.x < ntiles)
        {
#pragma unroll
            for (int jj = 0; jj < NJ; jj++)
                if (jj < nj)
                    pre[jj] = NT ? __builtin_nontemporal_load(reinterpret_cast<const v2d*>(src(t + gridDim.x, w + 4 * jj)))
                                 : *reinterpret_cast<const v2d*>(src(t + gridDim.x, w + 4 * jj));
        }
        v2d acc = {0.0, 0.0};
#pragma unroll
        for (int jj = 0; jj < NJ; jj++)
            if (jj < nj)
                acc += cur[jj];
#pragma unroll
        for (int jj = 0; jj < NJ; jj++)
            if (jj < np)
            {
                const v2d o = acc * double(jj + 1);
                if (NT)
                    __builtin_nontemporal_store(o, reinterpret_cast<v2d*>(dst(t, w + 4 * jj)));
                else
                    *reinterpret_cast<v2d*>(dst(t, w + 4 * jj)) = o;
            }
    }
}

__global__ __launch_bounds__(256) void k_copy(const double* __restrict__ a, double* __restrict__ b, int64_t npairs)
{
    // four 16-byte loads in flight per thread, consecutive threads on consecutive 16 bytes
    const int64_t stride = int64_t(gridDim.x) * 256;
    int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    for (; i + 3 * stride < npairs; i += 4 * stride)
    {
        v2d t[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
            t[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(a) + i + u * stride);
#pragma unroll
        for (int u = 0; u < 4; u++)
            __builtin_nontemporal_store(t[u], reinterpret_cast<v2d*>(b) + i + u * stride);
    }
    for (; i < npairs; i += stride)
        __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const v2d*>(a) + i), reinterpret_cast<v2d*>(b) + i);
}

int main(int argc, char** argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
    const int m = 40;
    const int64_t ldv = n;
    double *V, *X;
    CHECK(hipMalloc(&V, size_t(ldv) * m * 8));
    CHECK(hipMalloc(&X, size_t(ldv) * m * 8));
    CHECK(hipMemset(V, 0, size_t(ldv) * m * 8));
    CHECK(hipMemset(X, 0, size_t(ldv) * m * 8));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto time_it = [&](const char* name, double bytes, auto launch) {
        for (int r = 0; r < 3; r++)
            launch();
        CHECK(hipEventRecord(e0));
        const int reps = 10;
        for (int r = 0; r < reps; r++)
            launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        printf("{\"pattern\": \"%s\", \"n\": %lld, \"ms\": %.4f, \"GB\": %.3f, \"TBps\": %.3f, \"frac_of_8\": %.3f}\n", name, (long long) n, ms, bytes / 1e9,
               bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 8e12);
        fflush(stdout);
    };
    const int cus = prop.multiProcessorCount;
    struct Case
    {
        const char* name;
        int p;
        bool inplace;
        int layout;
        bool nt;
        int wgs_per_cu;
    };
    const Case cases[] = {
        {"vq: 40 read / 27 written in place, column-major, non-temporal, 3 wg/cu", 27, true, 0, true, 3},
        {"vq: 40 read / 27 written in place, column-major, plain loads+stores, 3 wg/cu", 27, true, 0, false, 3},
        {"vq: 40 read / 27 written OUT of place, column-major, non-temporal, 3 wg/cu", 27, false, 0, true, 3},
        {"vq: 40 read / 27 written in place, column-major, non-temporal, 6 wg/cu", 27, true, 0, true, 6},
        {"vq: 40 read / 27 written in place, column-major, non-temporal, 12 wg/cu", 27, true, 0, true, 12},
        {"reads only: 40 read / 1 written, column-major, non-temporal, 3 wg/cu", 1, false, 0, true, 3},
        {"vq: 40 read / 27 written in place, ROW-BLOCKED tiles, non-temporal, 3 wg/cu", 27, true, 1, true, 3},
        {"vq: 40 read / 27 written in place, ROW-BLOCKED tiles, non-temporal, 6 wg/cu", 27, true, 1, true, 6},
        {"eigenvectors: 40 read / 20 written out of place, column-major, non-temporal, 3 wg/cu", 20, false, 0, true, 3},
    };
    for (const Case& c : cases)
    {
        const double bytes = 8.0 * double(n) * (m + c.p);
        const dim3 grid(unsigned(cus * c.wgs_per_cu));
        double* out = c.inplace ? V : X;
        if (c.nt)
            time_it(c.name, bytes, [&] { hipLaunchKernelGGL((k_pattern<10, true>), grid, dim3(256), 0, 0, V, out, ldv, ldv, m, c.p, n, c.layout); });
        else
            time_it(c.name, bytes, [&] { hipLaunchKernelGGL((k_pattern<10, false>), grid, dim3(256), 0, 0, V, out, ldv, ldv, m, c.p, n, c.layout); });
    }
    {
        const int64_t npairs = n * 20 / 2;  // 20 columns' worth: 1.6 GB read + 1.6 GB written
        time_it("flat copy of 1.6 GB, non-temporal (the device's copy rate)", 32.0 * double(npairs), [&] {
            hipLaunchKernelGGL(k_copy, dim3(unsigned(cus * 16)), dim3(256), 0, 0, V, X, npairs);
        });
    }
    return 0;
}
