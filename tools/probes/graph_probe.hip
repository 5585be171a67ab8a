// Stand-alone probe (not part of the library): what would capturing a sweep of device-driven Lanczos steps in a hipGraph buy for
// SMALL problems, where the solve is bound by launches (38 us per operation at n = 1000, 5 launches per step)?  A chain of K tiny
// dependent kernels (each reads a flag, bumps a counter) is timed three ways: enqueued eagerly with hipLaunchKernelGGL, replayed
// from an instantiated graph of the same K nodes, and — the floor — one kernel that loops K times.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/graph_probe.bin tools/probes/graph_probe.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

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

struct Args  // about the size of the library's OrthArgs / FinishArgs: passed by value
{
    double* p[8];
    double s[8];
    int i[8];
};

__global__ __launch_bounds__(256) void k_step(Args a, const int* __restrict__ status, double* __restrict__ x, int n)
{
    if (*status != 0)
        return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        x[i] = x[i] * a.s[0] + a.s[1];
}

int main(int argc, char** argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 150;   // launches of one sweep (30 steps x 5)
    const int n = argc > 2 ? atoi(argv[2]) : 1000;  // problem size: one to four workgroups
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    double* x;
    int* status;
    CHECK(hipMalloc(&x, size_t(n) * sizeof(double)));
    CHECK(hipMalloc(&status, sizeof(int)));
    CHECK(hipMemset(x, 0, size_t(n) * sizeof(double)));
    CHECK(hipMemset(status, 0, sizeof(int)));
    Args a{};
    a.s[0] = 1.0;
    a.s[1] = 1e-9;
    const dim3 grid((n + 255) / 256), block(256);
    auto enqueue = [&]() {
        for (int k = 0; k < K; k++)
            hipLaunchKernelGGL(k_step, grid, block, 0, st, a, status, x, n);
    };
    auto wall = [&](auto&& f, int reps) {
        f();
        CHECK(hipStreamSynchronize(st));
        double best = 1e30;
        for (int r = 0; r < reps; r++)
        {
            const auto t0 = std::chrono::steady_clock::now();
            f();
            CHECK(hipStreamSynchronize(st));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            best = us < best ? us : best;
        }
        return best;
    };
    const double eager = wall(enqueue, 20);
    hipGraph_t graph;
    hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    enqueue();
    CHECK(hipStreamEndCapture(st, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    const double replay = wall([&]() { CHECK(hipGraphLaunch(exec, st)); }, 20);
    printf("{\"launches\": %d, \"n\": %d, \"eager_us_per_launch\": %.2f, \"graph_us_per_launch\": %.2f, \"eager_us_per_sweep\": %.1f, "
           "\"graph_us_per_sweep\": %.1f}\n",
           K, n, eager / K, replay / K, eager, replay);
    return 0;
}
