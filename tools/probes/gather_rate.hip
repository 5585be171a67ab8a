// Random 8-byte gathers from an L2-resident window of x: plain loads vs loads that bypass the vector L1 (non-temporal, sc1).
// hipcc --offload-arch=gfx950 -O3 tools/probes/gather_rate.hip -o /tmp/gather_rate && /tmp/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t z)
{
    z ^= z >> 16; z *= 0x7feb352du; z ^= z >> 15; z *= 0x846ca68bu; z ^= z >> 16;
    return z;
}

template <int MODE>
__device__ __forceinline__ double ld(const double* p)
{
    if (MODE == 1)
        return __builtin_nontemporal_load(p);
    if (MODE == 2)
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// every workgroup gathers from the window [w0, w0 + wlen) (wlen a power of two); K gathers per thread, 8 in flight
template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const double* __restrict__ x, uint32_t wlen, uint32_t nwin, int K, double* __restrict__ out)
{
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t w0 = (blockIdx.x % nwin) * wlen;  // blocks resident together use nwin different windows
    double acc = 0.0;
    uint32_t s = gid * 2654435761u + 12345u;
    for (int k = 0; k < K; k += 8)
    {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
            s = mix(s + u);
            v[u] = ld<MODE>(x + w0 + (s & (wlen - 1)));
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            acc += v[u];
    }
    if (acc == 123.456)
        out[gid] = acc;
}

int main()
{
    const size_t n = size_t(1) << 24;  // 128 MB of x
    double *x, *out;
    hipMalloc(&x, n * 8);
    hipMalloc(&out, size_t(1) << 24);
    hipMemset(x, 0, n * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * 8 * 4, K = 512;
    const double total = double(blocks) * 256 * K;
    for (uint32_t wlen : {uint32_t(1) << 12, uint32_t(1) << 16, uint32_t(1) << 18, uint32_t(1) << 20, uint32_t(1) << 24})
        for (uint32_t nwin : {1u, 8u})
        {
            if (uint64_t(wlen) * nwin > n)
                continue;
            for (int mode = 0; mode < 3; mode++)
            {
                float best = 1e30f;
                for (int rep = 0; rep < 3; rep++)
                {
                    hipEventRecord(e0);
                    if (mode == 0)
                        hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(256), 0, 0, x, wlen, nwin, K, out);
                    else if (mode == 1)
                        hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(256), 0, 0, x, wlen, nwin, K, out);
                    else
                        hipLaunchKernelGGL(k_gather<2>, dim3(blocks), dim3(256), 0, 0, x, wlen, nwin, K, out);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    best = ms < best ? ms : best;
                }
                printf("{\"window_KiB\": %u, \"windows\": %u, \"load\": \"%s\", \"ms\": %.4f, \"G_gathers_per_s\": %.1f}\n", wlen / 128, nwin,
                       mode == 0 ? "plain" : (mode == 1 ? "nontemporal" : "sc1 (atomic relaxed agent)"), best, total / best * 1e-6);
            }
        }
    return 0;
}
