// Stand-alone probe (not part of the library): does a buffer that one kernel writes and the next one reads stay in the 256 MiB
// Infinity Cache when it is small enough — i.e. would the staged SpMV gain from running its two phases group by group over a
// re-used slice of the product array instead of once over the whole 1.2 GB?
//
// Per group g of G the probe runs the byte flows of the two phases of M-rand (150 M entries):
//   P1: reads 10 B per entry from a large stream (values + local columns: never re-read), writes 8 B per entry to the slice
//   P2: reads the slice (8 B per entry) and 2 B per entry from a second large stream (row | rank); for G > 1 it also reads and
//       writes the 80 MB of row accumulators that have to survive between the groups
// and reports the time per complete pass (all groups) for G = 1, 2, 4, 8, 16, 32, with plain and with non-temporal accesses on
// the re-used slice.  If the slice stays on the die, G = 8 (150 MB slice) moves 1.5 + 0.3 GB through HBM plus 1.3 GB of
// accumulators instead of 4.2 GB.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mall_probe.bin tools/probes/mall_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

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

// out[i] = f(a[i], b-stream): reads n v2d of `a` (non-temporal) and n/4 v2d of `b`, writes n v2d of `out` (NT_OUT: non-temporal)
template <bool NT_OUT>
__global__ __launch_bounds__(1024) void k_p1(const v2d* __restrict__ a, const v2d* __restrict__ b, v2d* __restrict__ out, int64_t n)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    {
        v2d x = __builtin_nontemporal_load(a + i);
        if ((i & 3) == 0)
        {
            const v2d y = __builtin_nontemporal_load(b + (i >> 2));
            x += y;
        }
        if (NT_OUT)
            __builtin_nontemporal_store(x, out + i);
        else
            out[i] = x;
    }
}

// acc-stream: reads n v2d of the slice (NT_IN: non-temporal), n/4 v2d of `r`; CARRY: reads and writes nacc v2d of y
template <bool NT_IN, bool CARRY>
__global__ __launch_bounds__(1024) void k_p2(const v2d* __restrict__ slice, const v2d* __restrict__ r, v2d* __restrict__ y, int64_t n, int64_t nacc,
                                             double* __restrict__ sink)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    v2d s = {0.0, 0.0};
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    {
        const v2d x = NT_IN ? __builtin_nontemporal_load(slice + i) : slice[i];
        s += x;
        if ((i & 3) == 0)
            s += __builtin_nontemporal_load(r + (i >> 2));
    }
    if (CARRY)
        for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nacc; i += stride)
            y[i] = y[i] + s;
    if (s.x + s.y == 12345.678)
        sink[0] = s.x;
}

int main(int argc, char** argv)
{
    const int64_t entries = argc > 1 ? atoll(argv[1]) : 150000000;
    const int64_t nv = entries / 2;            // v2d elements of an 8-byte-per-entry array
    const int64_t nacc = 10000000 / 2;         // 80 MB of accumulators
    v2d *val, *lcol, *prod, *rr, *y;
    double* sink;
    CHECK(hipMalloc(&val, nv * sizeof(v2d)));
    CHECK(hipMalloc(&lcol, (nv / 4 + 4) * sizeof(v2d)));
    CHECK(hipMalloc(&prod, nv * sizeof(v2d)));
    CHECK(hipMalloc(&rr, (nv / 4 + 4) * sizeof(v2d)));
    CHECK(hipMalloc(&y, nacc * sizeof(v2d)));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(val, 0, nv * sizeof(v2d)));
    CHECK(hipMemset(lcol, 0, (nv / 4 + 4) * sizeof(v2d)));
    CHECK(hipMemset(prod, 0, nv * sizeof(v2d)));
    CHECK(hipMemset(rr, 0, (nv / 4 + 4) * sizeof(v2d)));
    CHECK(hipMemset(y, 0, nacc * sizeof(v2d)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grid = 512;
    for (int nt = 0; nt < 2; nt++)
        for (int G : {1, 2, 4, 8, 16, 32})
        {
            const int64_t per = (nv / G) & ~int64_t(3);
            float best = 1e30f;
            for (int rep = 0; rep < 6; rep++)
            {
                CHECK(hipEventRecord(e0, 0));
                for (int g = 0; g < G; g++)
                {
                    // the slice is RE-USED by every group (G = 1: the whole array)
                    if (nt)
                        hipLaunchKernelGGL((k_p1<true>), dim3(grid), dim3(1024), 0, 0, val + g * per, lcol + g * (per / 4), prod, per);
                    else
                        hipLaunchKernelGGL((k_p1<false>), dim3(grid), dim3(1024), 0, 0, val + g * per, lcol + g * (per / 4), prod, per);
                    if (G == 1)
                    {
                        if (nt)
                            hipLaunchKernelGGL((k_p2<true, false>), dim3(grid), dim3(1024), 0, 0, prod, rr, y, per, nacc, sink);
                        else
                            hipLaunchKernelGGL((k_p2<false, false>), dim3(grid), dim3(1024), 0, 0, prod, rr, y, per, nacc, sink);
                    }
                    else if (nt)
                        hipLaunchKernelGGL((k_p2<true, true>), dim3(grid), dim3(1024), 0, 0, prod, rr + g * (per / 4), y, per, nacc, sink);
                    else
                        hipLaunchKernelGGL((k_p2<false, true>), dim3(grid), dim3(1024), 0, 0, prod, rr + g * (per / 4), y, per, nacc, sink);
                }
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best)
                    best = ms;
            }
            const double slice_mb = double(per) * 16.0 / 1e6;
            const double bytes = double(per) * G * (16.0 + 4.0 + 16.0 + 16.0 + 4.0) + (G > 1 ? double(G) * nacc * 32.0 : 0.0);
            printf("{\"slice_accesses\": \"%s\", \"groups\": %d, \"slice_MB\": %.0f, \"ms_per_pass\": %.4f, \"bytes_requested_GB\": %.3f, "
                   "\"requested_TBps\": %.2f}\n",
                   nt ? "non-temporal" : "plain", G, slice_mb, best, bytes / 1e9, bytes / (best * 1e-3) / 1e12);
            fflush(stdout);
        }
    return 0;
}
