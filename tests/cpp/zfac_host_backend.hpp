// TEST INFRASTRUCTURE.  A plain host implementation of the vector primitives spectra_amd/csrc/zfac_flow.hpp is written over (the
// library's own implementation is the HIP backend of csrc/zfac.hip): loops over std::complex<double> arrays, left to right.  Shared by
// tests/cpp/zfac_flow_host.cpp (the flow's checks as a program) and tests/cpp/zfac_host_capi.cpp (the same flow behind the C entry
// points, for tests/test_host_zfac.py).
#pragma once
#include <zfac_flow.hpp>

#include <algorithm>
#include <cmath>
#include <complex>
#include <stdexcept>
#include <vector>

using cd = std::complex<double>;
typedef int (*zop_fn)(void* user, const double* x_host, double* y_host);

struct zdense
{
    int64_t rows = 0, cols = 0;
    std::vector<cd> a;
};

struct HostBackend
{
    int64_t n = 0;
    const zdense* dense = nullptr;
    zop_fn op = nullptr;
    void* user = nullptr;
    cd* alloc(size_t count) { return new cd[count](); }
    void release(cd* p) { delete[] p; }
    void upload(cd* dev, const cd* host, int64_t count) { std::copy(host, host + count, dev); }
    void download(cd* host, const cd* dev, int64_t count) { std::copy(dev, dev + count, host); }
    void apply(const cd* x, cd* y)
    {
        if (op)
        {
            if (op(user, reinterpret_cast<const double*>(x), reinterpret_cast<double*>(y)) != 0)
                throw std::runtime_error("user operator failed");
            return;
        }
        for (int64_t i = 0; i < n; i++)
        {
            cd acc(0.0);
            for (int64_t j = 0; j < n; j++)
                acc += dense->a[size_t(j * n + i)] * x[j];
            y[i] = acc;
        }
    }
    void dotc(const cd* X, int64_t ldx, int ncols, const cd* y, cd* out)
    {
        for (int j = 0; j < ncols; j++)
        {
            cd acc(0.0);
            for (int64_t i = 0; i < n; i++)
                acc += std::conj(X[j * ldx + i]) * y[i];
            out[j] = acc;
        }
    }
    void update(cd* f, const cd* w, const cd* V, int64_t ldv, int ncols, const cd* h)
    {
        for (int64_t i = 0; i < n; i++)
        {
            cd acc = w[i];
            for (int j = 0; j < ncols; j++)
                acc -= V[j * ldv + i] * h[j];
            f[i] = acc;
        }
    }
    void scale_copy(cd* dst, const cd* src, double alpha)
    {
        for (int64_t i = 0; i < n; i++)
            dst[i] = alpha * src[i];
    }
    void axpy(cd* y, cd a, const cd* x)
    {
        for (int64_t i = 0; i < n; i++)
            y[i] += a * x[i];
    }
    double norm(const cd* x)
    {
        double s = 0.0;
        for (int64_t i = 0; i < n; i++)
            s += std::norm(x[i]);
        return std::sqrt(s);
    }
    double absmax(const cd* x)
    {
        double m = 0.0;
        for (int64_t i = 0; i < n; i++)
            m = std::max(m, std::abs(x[i]));
        return m;
    }
    void zero(cd* x) { std::fill(x, x + n, cd(0.0)); }
};

