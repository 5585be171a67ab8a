// The control flow of the complex-scalar factorisation (spectra_amd/csrc/zfac_flow.hpp) run on a plain HOST backend — test
// infrastructure: the library instantiates the same template with the HIP backend of zfac.hip, this program checks the flow where
// there is no GPU.  Checks of the reference's test/Arnoldi.cpp:20-85 (A V - V H = f e', V^H V = I to 1e-12) for general and
// Hermitian complex matrices, plus the breakdown paths (invariant subspace -> expand_basis, Arnoldi.h:66-115).
//   g++ -std=c++17 -O2 -I spectra_amd/csrc tests/cpp/zfac_flow_host.cpp   (shares tests/cpp/zfac_host_backend.hpp with zfac_host_capi.cpp)
#include "zfac_host_backend.hpp"

#include <cstdio>
#include <random>

static int failures = 0;
#define REQUIRE(cond)                                                        \
    do                                                                       \
    {                                                                        \
        if (!(cond))                                                         \
        {                                                                    \
            std::printf("REQUIRE failed at line %d: %s\n", __LINE__, #cond); \
            failures++;                                                      \
        }                                                                    \
    } while (0)

static std::mt19937 gen(123);
static double rnd() { return std::uniform_real_distribution<double>(-1.0, 1.0)(gen); }

// || A V - V H - f e_k' ||_max over the first k columns, and || V^H V - I ||_max
static void check(HostBackend& be, mispec::ZFacFlow<HostBackend>& fac, int k, double tol)
{
    const int64_t n = be.n;
    const int m = fac.max_dim();
    REQUIRE(fac.subspace_dim() == k);
    std::vector<cd> V(size_t(n) * k), f(static_cast<size_t>(n));
    fac.get_V(V.data(), k);
    fac.get_f(f.data());
    const std::vector<cd>& H = fac.matrix_H();
    double res = 0.0, orth = 0.0, fn = 0.0;
    std::vector<cd> av(static_cast<size_t>(n));
    for (int j = 0; j < k; j++)
    {
        be.apply(V.data() + size_t(j) * n, av.data());
        for (int64_t i = 0; i < n; i++)
        {
            cd r = av[size_t(i)];
            for (int l = 0; l < k; l++)
                r -= V[size_t(l) * n + i] * H[size_t(j) * m + l];
            if (j == k - 1)
                r -= f[size_t(i)];
            res = std::max(res, std::abs(r));
        }
    }
    for (int a = 0; a < k; a++)
        for (int b = 0; b < k; b++)
        {
            cd acc(0.0);
            for (int64_t i = 0; i < n; i++)
                acc += std::conj(V[size_t(a) * n + i]) * V[size_t(b) * n + i];
            orth = std::max(orth, std::abs(acc - (a == b ? cd(1.0) : cd(0.0))));
        }
    for (int64_t i = 0; i < n; i++)
        fn += std::norm(f[size_t(i)]);
    REQUIRE(res <= tol);
    REQUIRE(orth <= tol);
    REQUIRE(std::fabs(std::sqrt(fn) - fac.f_norm()) <= tol);
}

static void run(int64_t n, int m, bool hermitian, int kind)
{
    zdense D;  // the operator: a dense column-major matrix
    D.rows = D.cols = n;
    std::vector<cd>& A = D.a;
    HostBackend be;
    be.n = n;
    be.dense = &D;
    A.assign(size_t(n) * n, cd(0.0));
    if (kind == 0)  // dense random
        for (auto& a : A)
            a = cd(rnd(), rnd());
    else if (kind == 1)  // block diagonal: the start vector lives in a 3-dimensional invariant subspace -> breakdown at step 3
        for (int64_t j = 0; j < n; j++)
            for (int64_t i = 0; i < n; i++)
                if ((i < 3) == (j < 3))
                    A[size_t(j * n + i)] = cd(rnd(), rnd());
    // kind == 2: the zero matrix (every step restarts; test/Example4.cpp's situation)
    if (hermitian)
        for (int64_t j = 0; j < n; j++)
        {
            for (int64_t i = 0; i < j; i++)
            {
                const cd s = A[size_t(j * n + i)] + std::conj(A[size_t(i * n + j)]);
                A[size_t(j * n + i)] = s;
                A[size_t(i * n + j)] = std::conj(s);
            }
            A[size_t(j * n + j)] = cd(2.0 * A[size_t(j * n + j)].real(), 0.0);
        }
    std::vector<cd> v0(static_cast<size_t>(n), cd(0.0));
    for (int64_t i = 0; i < (kind == 1 ? 3 : n); i++)
        v0[size_t(i)] = cd(rnd(), rnd());
    mispec::ZFacFlow<HostBackend> fac(be, n, m, hermitian);
    int64_t ops = 0;
    fac.init(v0.data(), ops);
    REQUIRE(ops == 2);
    check(be, fac, 1, 1e-12);
    fac.factorize_from(1, m / 2, ops);
    check(be, fac, m / 2, 1e-12);
    fac.factorize_from(m / 2, m, ops);
    check(be, fac, m, 1e-12);
    REQUIRE(ops >= 2 + (m - 1));
    if (kind == 0)
        REQUIRE(ops == 2 + (m - 1));
    if (hermitian)
    {
        // Lanczos keeps H tridiagonal; the off-diagonal pair is stored equal, H(i-1,i) = H(i,i-1), as the reference does
        // (Lanczos.h:124-125)
        const std::vector<cd>& H = fac.matrix_H();
        double off = 0.0, asym = 0.0;
        for (int j = 0; j < m; j++)
            for (int i = 0; i < m; i++)
            {
                if (std::abs(i - j) > 1)
                    off = std::max(off, std::abs(H[size_t(j) * m + i]));
                asym = std::max(asym, std::abs(H[size_t(j) * m + i] - H[size_t(i) * m + j]));
            }
        REQUIRE(off == 0.0);
        REQUIRE(asym == 0.0);
    }
    bool threw = false;
    try
    {
        fac.factorize_from(m + 1, m + 2, ops);
    }
    catch (const std::invalid_argument&)
    {
        threw = true;
    }
    REQUIRE(threw);
}

int main()
{
    for (int herm = 0; herm < 2; herm++)
    {
        run(10, 6, herm != 0, 0);  // test/Arnoldi.cpp:122-158
        run(200, 30, herm != 0, 0);
        run(40, 10, herm != 0, 1);
        run(12, 6, herm != 0, 2);
    }
    {
        zdense D;
        D.rows = D.cols = 5;
        D.a.assign(25, cd(1.0));
        HostBackend be;
        be.n = 5;
        be.dense = &D;
        mispec::ZFacFlow<HostBackend> fac(be, 5, 3, false);
        std::vector<cd> z(5, cd(0.0));
        int64_t ops = 0;
        bool threw = false;
        try
        {
            fac.init(z.data(), ops);
        }
        catch (const std::invalid_argument&)
        {
            threw = true;
        }
        REQUIRE(threw);  // Arnoldi.h:146-148
    }
    std::printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
    return failures ? 1 : 0;
}
