// Shifted QR factorisations of the projected matrix, host side (reference: LinAlg/UpperHessenbergQR.h).
//   UpperHessenbergQR<double>: H - sI = QR for an upper Hessenberg H   (:45-460)
//   TridiagQR<double>:         the same for a symmetric tridiagonal T  (:470-711)
// Same member names as the reference: compute(), matrix_R(), matrix_QtHQ(), apply_QY(), apply_QtY(), apply_YQ(),
// apply_YQt() (vector and matrix forms, :204-460).  The symmetric solver does
// NOT use these classes on its fast path — all sweeps of a restart run as one skewed pipeline
// (internal/SmallDensePipelined.h: host core by default, csrc/small.hip kernels with small=device; same
// arithmetic as internal/SmallDense.h); they serve the general solver's restart, user code and the
// CPU-side unit tests.
#ifndef MISPEC_SPECTRA_UPPER_HESSENBERG_QR_H
#define MISPEC_SPECTRA_UPPER_HESSENBERG_QR_H

#include <complex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../internal/Dense.h"
#include "../internal/SmallDenseComplex.h"
#include "../internal/SmallDenseGen.h"

namespace Spectra {

template <typename Scalar = double>
class UpperHessenbergQR
{
protected:
    using Matrix = DenseMatrix<Scalar>;
    Index m_n = 0;
    Scalar m_shift = 0;
    std::vector<double> m_rot;   // cos[0..n-1), sin at m_rot[n + i]
    std::vector<double> m_QtHQ;  // n x n column-major
    std::vector<double> m_Hs;    // H - sI as given (the part compute() reads), for matrix_R()
    bool m_computed = false;
    using Vector = DenseVector<Scalar>;

    // Q = G_0 G_1 ... G_{n-2}; on the pair (i, i+1) G_i is [c s; -s c]  (reference :84-92)
    // rows of an (n x ncol) block: Y <- G_i Y  (transpose == false)  or  Y <- G_i' Y
    void rotate_rows(Scalar* Y, Index ldy, Index ncol, Index i, bool transpose) const
    {
        const double c = m_rot[std::size_t(i)], s = transpose ? -m_rot[std::size_t(m_n + i)] : m_rot[std::size_t(m_n + i)];
        for (Index j = 0; j < ncol; j++)
        {
            Scalar* col = Y + j * ldy;
            const Scalar a = col[i], b = col[i + 1];
            col[i] = c * a + s * b;
            col[i + 1] = -s * a + c * b;
        }
    }

    void require_computed(const char* who) const
    {
        if (!m_computed)
            throw std::logic_error(std::string(who) + ": need to call compute() first");
    }

public:
    explicit UpperHessenbergQR(Index size = 0) : m_n(size) {}
    UpperHessenbergQR(const Matrix& mat, const Scalar& shift = Scalar(0)) { compute(mat, shift); }
    virtual ~UpperHessenbergQR() {}

    // Only the upper triangle and the sub-diagonal of mat are read.
    virtual void compute(const Matrix& mat, const Scalar& shift = Scalar(0))
    {
        m_n = mat.rows();
        if (m_n != mat.cols())
            throw std::invalid_argument("UpperHessenbergQR: matrix must be square");
        m_shift = shift;
        const int n = static_cast<int>(m_n);
        m_QtHQ.assign(mat.data(), mat.data() + std::size_t(n) * n);
        m_Hs.assign(std::size_t(n) * n, 0.0);
        for (int j = 0; j < n; j++)
            for (int i = 0; i <= (j + 1 < n ? j + 1 : n - 1); i++)
                m_Hs[std::size_t(j) * n + i] = mat(i, j) - (i == j ? shift : Scalar(0));
        m_rot.assign(std::size_t(2) * n, 0.0);
        double dummy = 0.0;
        mispec::small::hess_shifted_qr(n, m_QtHQ.data(), n, shift, &dummy, 1, 0, m_rot.data());
        m_computed = true;
    }

    // The R factor of H - sI = QR, an upper triangular matrix (reference :204-210)
    virtual Matrix matrix_R() const
    {
        require_computed("UpperHessenbergQR");
        Matrix R(m_n, m_n);
        std::copy(m_Hs.begin(), m_Hs.end(), R.data());
        for (Index i = 0; i < m_n - 1; i++)  // R = G_{n-2}' ... G_0' (H - sI)
            rotate_rows(R.data(), m_n, m_n, i, true);
        for (Index j = 0; j < m_n; j++)
            for (Index i = j + 1; i < m_n; i++)
                R(i, j) = Scalar(0);
        return R;
    }

    // Y <- Q Y = G_0 G_1 ... Y   (reference :266-287, :322-346)
    void apply_QY(Vector& Y) const
    {
        require_computed("UpperHessenbergQR");
        for (Index i = m_n - 2; i >= 0; i--)
            rotate_rows(Y.data(), m_n, 1, i, false);
    }
    void apply_QY(Matrix& Y) const
    {
        require_computed("UpperHessenbergQR");
        for (Index i = m_n - 2; i >= 0; i--)
            rotate_rows(Y.data(), Y.rows(), Y.cols(), i, false);
    }
    // Y <- Q' Y   (reference :293-316, :352-377)
    void apply_QtY(Vector& Y) const
    {
        require_computed("UpperHessenbergQR");
        for (Index i = 0; i < m_n - 1; i++)
            rotate_rows(Y.data(), m_n, 1, i, true);
    }
    void apply_QtY(Matrix& Y) const
    {
        require_computed("UpperHessenbergQR");
        for (Index i = 0; i < m_n - 1; i++)
            rotate_rows(Y.data(), Y.rows(), Y.cols(), i, true);
    }
    // Y <- Y Q'   (reference :429-459)
    void apply_YQt(Matrix& Y) const
    {
        require_computed("UpperHessenbergQR");
        const Index nrow = Y.rows();
        for (Index i = m_n - 2; i >= 0; i--)
        {
            const double c = m_rot[std::size_t(i)], s = m_rot[std::size_t(m_n + i)];
            Scalar* a = Y.data() + i * nrow;
            Scalar* b = a + nrow;
            for (Index j = 0; j < nrow; j++)
            {
                const Scalar t = a[j];
                a[j] = c * t + s * b[j];
                b[j] = -s * t + c * b[j];
            }
        }
    }

    // dest <- Q'HQ = RQ + sI
    virtual void matrix_QtHQ(Matrix& dest) const
    {
        require_computed("UpperHessenbergQR");
        dest.resize(m_n, m_n);
        std::copy(m_QtHQ.begin(), m_QtHQ.end(), dest.data());
    }

    // Y <- Y * Q = Y * G1 * G2 * ...
    void apply_YQ(Matrix& Y) const
    {
        require_computed("UpperHessenbergQR");
        const Index nrow = Y.rows();
        for (Index i = 0; i < m_n - 1; i++)
        {
            const double c = m_rot[std::size_t(i)], s = m_rot[std::size_t(m_n + i)];
            Scalar* a = Y.data() + i * nrow;
            Scalar* b = a + nrow;
            for (Index j = 0; j < nrow; j++)
            {
                const Scalar t = a[j];
                a[j] = c * t - s * b[j];
                b[j] = s * t + c * b[j];
            }
        }
    }
};

// Complex upper Hessenberg matrices (reference: the same class template instantiated with std::complex, :45-460):
// G_i = [c s; -conj(s) c] with real c, Q = G_0 G_1 ... G_{n-2}, Q^H in place of Q'.  Host arithmetic only
// (internal/SmallDenseComplex.h) — the device solvers are real.
template <typename RealScalar>
class UpperHessenbergQR<std::complex<RealScalar>>
{
public:
    using Scalar = std::complex<RealScalar>;

protected:
    using Matrix = DenseMatrix<Scalar>;
    using Vector = DenseVector<Scalar>;
    using cd = std::complex<double>;
    Index m_n = 0;
    Scalar m_shift = Scalar(0);
    std::vector<double> m_cos;
    std::vector<cd> m_sin;
    std::vector<cd> m_R;  // n x n column-major
    bool m_computed = false;

    void require_computed() const
    {
        if (!m_computed)
            throw std::logic_error("UpperHessenbergQR: need to call compute() first");
    }
    // rows i, i+1 of an (. x ncol) block with leading dimension ldy:  Y <- G_i Y  or  Y <- G_i^H Y
    void rotate_rows(Scalar* Y, Index ldy, Index ncol, Index i, bool adjoint) const
    {
        const double c = m_cos[std::size_t(i)];
        const cd s = adjoint ? -m_sin[std::size_t(i)] : m_sin[std::size_t(i)];
        for (Index j = 0; j < ncol; j++)
        {
            Scalar* col = Y + j * ldy;
            const cd a = col[i], b = col[i + 1];
            col[i] = Scalar(c * a + s * b);
            col[i + 1] = Scalar(-std::conj(s) * a + c * b);
        }
    }
    // columns i, i+1 of an (nrow x .) block:  Y <- Y G_i  or  Y <- Y G_i^H
    void rotate_cols(Scalar* Y, Index nrow, Index i, bool adjoint) const
    {
        const double c = m_cos[std::size_t(i)];
        const cd s = adjoint ? -m_sin[std::size_t(i)] : m_sin[std::size_t(i)];
        Scalar* a = Y + i * nrow;
        Scalar* b = a + nrow;
        for (Index j = 0; j < nrow; j++)
        {
            const cd t = a[j], u = b[j];
            a[j] = Scalar(c * t - std::conj(s) * u);
            b[j] = Scalar(s * t + c * u);
        }
    }

public:
    explicit UpperHessenbergQR(Index size = 0) : m_n(size) {}
    UpperHessenbergQR(const Matrix& mat, const Scalar& shift = Scalar(0)) { compute(mat, shift); }
    virtual ~UpperHessenbergQR() {}

    virtual void compute(const Matrix& mat, const Scalar& shift = Scalar(0))
    {
        m_n = mat.rows();
        if (m_n != mat.cols())
            throw std::invalid_argument("UpperHessenbergQR: matrix must be square");
        m_shift = shift;
        const int n = static_cast<int>(m_n);
        std::vector<cd> H(std::size_t(n) * n);
        for (int j = 0; j < n; j++)
            for (int i = 0; i < n; i++)
                H[std::size_t(j) * n + i] = cd(mat(i, j));
        m_cos.assign(std::size_t(n > 0 ? n : 1), 0.0);
        m_sin.assign(std::size_t(n > 0 ? n : 1), cd(0.0));
        m_R.assign(std::size_t(n) * n, cd(0.0));
        mispec::small::hess_shifted_qr_complex(n, H.data(), n, cd(shift), m_R.data(), m_cos.data(), m_sin.data());
        m_computed = true;
    }

    virtual Matrix matrix_R() const
    {
        require_computed();
        Matrix R(m_n, m_n);
        for (Index j = 0; j < m_n; j++)
            for (Index i = 0; i < m_n; i++)
                R(i, j) = Scalar(m_R[std::size_t(j) * m_n + i]);
        return R;
    }

    // dest <- Q^H H Q = RQ + sI
    virtual void matrix_QtHQ(Matrix& dest) const
    {
        require_computed();
        const int n = static_cast<int>(m_n);
        std::vector<cd> out(std::size_t(n) * n);
        mispec::small::hess_rq_complex(n, m_R.data(), cd(m_shift), m_cos.data(), m_sin.data(), out.data());
        dest.resize(m_n, m_n);
        for (Index j = 0; j < m_n; j++)
            for (Index i = 0; i < m_n; i++)
                dest(i, j) = Scalar(out[std::size_t(j) * m_n + i]);
    }

    void apply_QY(Vector& Y) const
    {
        require_computed();
        for (Index i = m_n - 2; i >= 0; i--)
            rotate_rows(Y.data(), m_n, 1, i, false);
    }
    void apply_QY(Matrix& Y) const
    {
        require_computed();
        for (Index i = m_n - 2; i >= 0; i--)
            rotate_rows(Y.data(), Y.rows(), Y.cols(), i, false);
    }
    void apply_QtY(Vector& Y) const
    {
        require_computed();
        for (Index i = 0; i < m_n - 1; i++)
            rotate_rows(Y.data(), m_n, 1, i, true);
    }
    void apply_QtY(Matrix& Y) const
    {
        require_computed();
        for (Index i = 0; i < m_n - 1; i++)
            rotate_rows(Y.data(), Y.rows(), Y.cols(), i, true);
    }
    void apply_YQ(Matrix& Y) const
    {
        require_computed();
        for (Index i = 0; i < m_n - 1; i++)
            rotate_cols(Y.data(), Y.rows(), i, false);
    }
    void apply_YQt(Matrix& Y) const
    {
        require_computed();
        for (Index i = m_n - 2; i >= 0; i--)
            rotate_cols(Y.data(), Y.rows(), i, true);
    }
};

template <typename Scalar = double>
class TridiagQR : public UpperHessenbergQR<Scalar>
{
    using Base = UpperHessenbergQR<Scalar>;
    using typename Base::Matrix;
    using Base::m_computed;
    using Base::m_n;
    using Base::m_QtHQ;
    using Base::m_rot;
    using Base::m_shift;

public:
    explicit TridiagQR(Index size = 0) : Base(size) {}
    TridiagQR(const Matrix& mat, const Scalar& shift = Scalar(0)) { compute(mat, shift); }

    // Only the diagonal and the sub-diagonal of mat are read.
    void compute(const Matrix& mat, const Scalar& shift = Scalar(0)) override
    {
        m_n = mat.rows();
        if (m_n != mat.cols())
            throw std::invalid_argument("TridiagQR: matrix must be square");
        m_shift = shift;
        const int n = static_cast<int>(m_n);
        std::vector<double> diag(static_cast<std::size_t>(n)), subd(static_cast<std::size_t>(n), 0.0), work(std::size_t(4) * n);
        for (int i = 0; i < n; i++)
            diag[std::size_t(i)] = mat(i, i);
        for (int i = 0; i < n - 1; i++)
            subd[std::size_t(i)] = mat(i + 1, i);
        double dummy = 0.0;
        mispec::small::tridiag_shifted_qr(n, diag.data(), subd.data(), shift, &dummy, 1, 0, work.data(),
                                          mispec::small::Lanes{0, 1});
        m_rot.assign(work.begin(), work.begin() + 2 * n);  // [cos | sin]
        this->m_Hs.assign(std::size_t(n) * n, 0.0);
        for (int i = 0; i < n; i++)
            this->m_Hs[std::size_t(i) * n + i] = mat(i, i) - shift;
        for (int i = 0; i < n - 1; i++)
            this->m_Hs[std::size_t(i) * n + i + 1] = this->m_Hs[std::size_t(i + 1) * n + i] = mat(i + 1, i);
        m_QtHQ.assign(std::size_t(n) * n, 0.0);
        for (int i = 0; i < n; i++)
            m_QtHQ[std::size_t(i) * n + i] = diag[std::size_t(i)];
        for (int i = 0; i < n - 1; i++)
            m_QtHQ[std::size_t(i) * n + i + 1] = m_QtHQ[std::size_t(i + 1) * n + i] = subd[std::size_t(i)];
        m_computed = true;
    }
};

}  // namespace Spectra

#endif
