// Control flow of the Arnoldi / Lanczos factorisation  A V = V H + f e'  for COMPLEX scalars, written once over a small set of
// vector primitives (the `Backend`).  The library instantiates it with the HIP backend of zfac.hip (V, f, w in HBM, the primitives
// are kernels); tests/cpp/zfac_flow_host.cpp instantiates the same flow with a plain host backend to check the control flow
// where there is no GPU.  Outside the hot path of SURVEY.md section 8 (the configs are real fp64): it exists because the
// reference's LinAlg/Arnoldi.h and Lanczos.h are templates over the scalar and its test/Arnoldi.cpp instantiates them with
// std::complex<double> next to double.  Host-driven steps in the reference's order — no fusion, no device-driven sweeps.
//
// Reference followed (yixuan/spectra, include/Spectra/): LinAlg/Arnoldi.h:66-115 (expand_basis), :136-195 (init), :198-295
// (factorize_from); LinAlg/Lanczos.h:62-187 (factorize_from); MatOp/internal/ArnoldiOp.h:113-162 (x^H y, X^H y, norm);
// Util/SimpleRandom.h:67-77 (complex draws: real part first).
//
// Backend concept (cd = std::complex<double>; "dev" pointers are whatever the backend allocates):
//   cd* alloc(size_t count);  void release(cd*);
//   void upload(cd* dev, const cd* host, int64_t count);  void download(cd* host, const cd* dev, int64_t count);
//   void apply(const cd* x_dev, cd* y_dev);                                   y = A x
//   void dotc(const cd* X_dev, int64_t ldx, int ncols, const cd* y_dev, cd* out_host);   out[j] = X[:, j]^H y
//   void update(cd* f_dev, const cd* w_dev, const cd* V_dev, int64_t ldv, int ncols, const cd* h_host);   f = w - V h (w may be f)
//   void scale_copy(cd* dst_dev, const cd* src_dev, double alpha);            dst = alpha * src
//   void axpy(cd* y_dev, cd a, const cd* x_dev);                              y += a x
//   double norm(const cd* x_dev);   double absmax(const cd* x_dev);   void zero(cd* x_dev);
#pragma once

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <complex>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace mispec {

// The full column-major image of a dense complex operator: a general matrix as given, or a Hermitian one from the triangle
// `uplo` ('L' / 'U') of the input — mirrored conjugated, the diagonal's imaginary part dropped, the other triangle never read
// (what mat.selfadjointView<Uplo>() means, MatOp/DenseHermMatProd.h).  src is rows x cols with leading dimension ld, row- or
// column-major; out has rows * cols entries.
inline void zdense_expand(int64_t rows, int64_t cols, const std::complex<double>* src, int64_t ld, bool row_major, char uplo,
                          std::complex<double>* out)
{
    using cd = std::complex<double>;
    auto in = [&](int64_t i, int64_t j) -> cd { return row_major ? src[i * ld + j] : src[j * ld + i]; };
    for (int64_t j = 0; j < cols; j++)
        for (int64_t i = 0; i < rows; i++)
        {
            cd v;
            if (uplo == 0)
                v = in(i, j);
            else if (i == j)
                v = cd(in(i, i).real(), 0.0);
            else
            {
                const bool stored = (uplo == 'L') ? (i > j) : (i < j);
                v = stored ? in(i, j) : std::conj(in(j, i));
            }
            out[j * rows + i] = v;
        }
}

template <typename Backend>
class ZFacFlow
{
public:
    using cd = std::complex<double>;

private:
    Backend& m_be;
    const int64_t m_n;
    const int m_m;
    const bool m_hermitian;  // Lanczos.h's three-term flow instead of Arnoldi.h's full projection
    int m_k = 0;
    double m_beta = 0.0;
    cd* m_V = nullptr;  // n x m, column-major, leading dimension n
    cd* m_f = nullptr;
    cd* m_w = nullptr;
    cd* m_t = nullptr;  // scratch vector (start vector, random draws)
    std::vector<cd> m_H;  // m x m, column-major, host

    static constexpr double kEps = DBL_EPSILON;
    static constexpr double kNear0 = DBL_MIN * 10;

    cd& H(int i, int j) { return m_H[size_t(j) * m_m + i]; }
    cd* col(int j) const { return m_V + int64_t(j) * m_n; }

    static double inf_norm(const cd* v, int count)
    {
        double e = 0.0;
        for (int i = 0; i < count; i++)
            e = std::max(e, std::abs(v[i]));
        return e;
    }

    // SimpleRandom<std::complex<double>>(seed).random_vec (SimpleRandom.h:30-123): one LCG stream, real part first
    void random_vec(cd* dev, unsigned long seed)
    {
        std::vector<cd> host(static_cast<size_t>(m_n));
        const unsigned long mx = 2147483647UL;
        unsigned long state = seed ? (seed & mx) : 1;
        auto next = [&]() -> double {
            const unsigned long lo = 16807UL * (state & 0xFFFFUL), hi = 16807UL * (state >> 16);
            unsigned long s = lo + ((hi & 0x7FFFUL) << 16);
            if (s > mx)
            {
                s &= mx;
                ++s;
            }
            s += hi >> 15;
            if (s > mx)
            {
                s &= mx;
                ++s;
            }
            state = s;
            return double(long(s)) / double(mx) - 0.5;
        };
        for (int64_t i = 0; i < m_n; i++)
        {
            const double re = next();
            const double im = next();
            host[size_t(i)] = cd(re, im);
        }
        m_be.upload(dev, host.data(), m_n);
    }

    // Arnoldi.h:66-115: a non-zero f with V[:, :ncols]^H f = 0
    void expand_basis(int ncols, unsigned long seed, int64_t& op_counter)
    {
        std::vector<cd> Vf(static_cast<size_t>(std::max(ncols, 1)));
        for (int iter = 0; iter < 5; iter++)
        {
            if (iter == 0)
            {
                random_vec(m_t, seed + 123UL * iter);
                m_be.apply(m_t, m_f);
                op_counter++;
            }
            else
                random_vec(m_f, seed + 123UL * iter);
            m_be.dotc(m_V, m_n, ncols, m_f, Vf.data());
            m_be.update(m_f, m_f, m_V, m_n, ncols, Vf.data());
            m_beta = m_be.norm(m_f);
            m_be.dotc(m_V, m_n, ncols, m_f, Vf.data());
            double ortho_err = inf_norm(Vf.data(), ncols);
            int count = 0;
            while (count < 3 && ortho_err >= kEps * m_beta)
            {
                m_be.update(m_f, m_f, m_V, m_n, ncols, Vf.data());
                m_beta = m_be.norm(m_f);
                m_be.dotc(m_V, m_n, ncols, m_f, Vf.data());
                ortho_err = inf_norm(Vf.data(), ncols);
                count++;
            }
            if (ortho_err < kEps * m_beta)
                return;
        }
    }

    void clear_H_from(int from_k)
    {
        for (int j = 0; j < m_m; j++)
            for (int i = 0; i < m_m; i++)
                if (j >= from_k || i >= from_k)
                    H(i, j) = cd(0.0);
    }

    // Arnoldi.h:198-295
    void factorize_general(int from_k, int to_m, int64_t& op_counter)
    {
        const double beta_thresh = kEps * std::sqrt(double(m_n));
        std::vector<cd> Vf(static_cast<size_t>(to_m)), h(static_cast<size_t>(to_m));
        clear_H_from(from_k);
        for (int i = from_k; i <= to_m - 1; i++)
        {
            bool restart = false;
            if (m_beta < kNear0)
            {
                expand_basis(i, 2UL * i, op_counter);
                restart = true;
            }
            m_be.scale_copy(col(i), m_f, 1.0 / m_beta);
            H(i, i - 1) = restart ? cd(0.0) : cd(m_beta);
            m_be.apply(col(i), m_w);
            op_counter++;
            const int i1 = i + 1;
            m_be.dotc(m_V, m_n, i1, m_w, h.data());
            m_be.update(m_f, m_w, m_V, m_n, i1, h.data());
            m_beta = m_be.norm(m_f);
            double hnorm2 = 0.0;
            for (int j = 0; j < i1; j++)
                hnorm2 += std::norm(h[size_t(j)]);
            if (!(m_beta > 0.717 * std::sqrt(hnorm2)))
            {
                m_be.dotc(m_V, m_n, i1, m_f, Vf.data());
                double ortho_err = inf_norm(Vf.data(), i1);
                int count = 0;
                while (count < 5 && ortho_err > kEps * m_beta)
                {
                    if (m_beta < beta_thresh)
                    {
                        m_be.zero(m_f);
                        m_beta = 0.0;
                        break;
                    }
                    m_be.update(m_f, m_f, m_V, m_n, i1, Vf.data());
                    for (int j = 0; j < i1; j++)
                        h[size_t(j)] += Vf[size_t(j)];
                    m_beta = m_be.norm(m_f);
                    m_be.dotc(m_V, m_n, i1, m_f, Vf.data());
                    ortho_err = inf_norm(Vf.data(), i1);
                    count++;
                }
            }
            for (int j = 0; j < i1; j++)
                H(j, i) = h[size_t(j)];
        }
    }

    // Lanczos.h:62-187
    void factorize_hermitian(int from_k, int to_m, int64_t& op_counter)
    {
        const double beta_thresh = kEps * std::sqrt(double(m_n));
        const double eps_sqrt = std::sqrt(kEps);
        std::vector<cd> Vf(static_cast<size_t>(to_m));
        clear_H_from(from_k);
        for (int i = from_k; i <= to_m - 1; i++)
        {
            bool restart = (m_beta < kNear0);
            if (!restart)
            {
                m_be.scale_copy(col(i), m_f, 1.0 / m_beta);
                if (m_beta < eps_sqrt)
                {
                    cd Viv;
                    m_be.dotc(col(i - 1), m_n, 1, col(i), &Viv);
                    restart = (std::abs(Viv) > eps_sqrt);
                }
            }
            if (restart)
            {
                expand_basis(i, 2UL * i, op_counter);
                m_be.scale_copy(col(i), m_f, 1.0 / m_beta);
            }
            H(i, i - 1) = restart ? cd(0.0) : cd(m_beta);
            H(i - 1, i) = H(i, i - 1);
            m_be.apply(col(i), m_w);
            op_counter++;
            if (!restart)
                m_be.axpy(m_w, -H(i, i - 1), col(i - 1));
            cd alpha;
            m_be.dotc(col(i), m_n, 1, m_w, &alpha);
            H(i, i) = alpha;
            m_be.update(m_f, m_w, col(i), m_n, 1, &alpha);
            m_beta = m_be.norm(m_f);
            const int i1 = i + 1;
            m_be.dotc(m_V, m_n, i1, m_f, Vf.data());
            double ortho_err = inf_norm(Vf.data(), i1);
            int count = 0;
            while (count < 5 && ortho_err > kEps * m_beta)
            {
                if (m_beta < beta_thresh)
                {
                    m_be.zero(m_f);
                    m_beta = 0.0;
                    break;
                }
                m_be.update(m_f, m_f, m_V, m_n, i1, Vf.data());
                H(i - 1, i) += Vf[size_t(i - 1)];
                H(i, i - 1) = H(i - 1, i);
                H(i, i) += Vf[size_t(i)];
                m_beta = m_be.norm(m_f);
                m_be.dotc(m_V, m_n, i1, m_f, Vf.data());
                ortho_err = inf_norm(Vf.data(), i1);
                count++;
            }
        }
    }

public:
    ZFacFlow(Backend& be, int64_t n, int m, bool hermitian) : m_be(be), m_n(n), m_m(m), m_hermitian(hermitian)
    {
        if (n < 1 || m < 1 || m > n)
            throw std::invalid_argument("complex factorisation: need 1 <= m <= n");
        m_V = m_be.alloc(size_t(n) * size_t(m));
        m_f = m_be.alloc(size_t(n));
        m_w = m_be.alloc(size_t(n));
        m_t = m_be.alloc(size_t(n));
        m_H.assign(size_t(m) * size_t(m), cd(0.0));
    }
    ZFacFlow(const ZFacFlow&) = delete;
    ZFacFlow& operator=(const ZFacFlow&) = delete;
    ~ZFacFlow()
    {
        m_be.release(m_V);
        m_be.release(m_f);
        m_be.release(m_w);
        m_be.release(m_t);
    }

    // Arnoldi.h:136-195
    void init(const cd* v0_host, int64_t& op_counter)
    {
        std::fill(m_H.begin(), m_H.end(), cd(0.0));
        m_be.upload(m_t, v0_host, m_n);
        const double v0norm = m_be.norm(m_t);
        if (v0norm < kNear0)
            throw std::invalid_argument("initial residual vector cannot be zero");
        cd* v = col(0);
        m_be.apply(m_t, v);
        op_counter++;
        const double vnorm = m_be.norm(v);
        if (vnorm < kNear0)
            m_be.scale_copy(v, m_t, 1.0 / v0norm);
        else
            m_be.scale_copy(v, v, 1.0 / vnorm);
        m_be.apply(v, m_w);
        op_counter++;
        cd h00;
        m_be.dotc(v, m_n, 1, m_w, &h00);
        H(0, 0) = h00;
        m_be.update(m_f, m_w, v, m_n, 1, &h00);
        if (m_be.absmax(m_f) < kEps * std::abs(h00))
        {
            m_be.zero(m_f);
            m_beta = 0.0;
        }
        else
            m_beta = m_be.norm(m_f);
        m_k = 1;
    }

    void factorize_from(int from_k, int to_m, int64_t& op_counter)
    {
        if (to_m <= from_k)
            return;
        if (from_k > m_k)
            throw std::invalid_argument(std::string(m_hermitian ? "Lanczos" : "Arnoldi") + ": from_k (= " + std::to_string(from_k) +
                                        ") is larger than the current subspace dimension (= " + std::to_string(m_k) + ")");
        if (from_k < 1 || to_m > m_m)
            throw std::invalid_argument("complex factorisation: need 1 <= from_k and to_m <= m");
        if (m_hermitian)
            factorize_hermitian(from_k, to_m, op_counter);
        else
            factorize_general(from_k, to_m, op_counter);
        m_k = to_m;
    }

    int subspace_dim() const { return m_k; }
    double f_norm() const { return m_beta; }
    int64_t rows() const { return m_n; }
    int max_dim() const { return m_m; }
    const std::vector<cd>& matrix_H() const { return m_H; }
    void get_V(cd* host, int ncols) const { m_be.download(host, m_V, m_n * int64_t(ncols)); }
    void get_f(cd* host) const { m_be.download(host, m_f, m_n); }
};

}  // namespace mispec
