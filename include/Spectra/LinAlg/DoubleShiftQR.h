// Francis double-shift QR step on a real upper Hessenberg matrix, host side
// (reference: LinAlg/DoubleShiftQR.h:20-470): H^2 - sH + tI = QR handled implicitly with 3-element
// Householder reflectors.  Same members as the reference: compute(mat, s, t), matrix_QtHQ(), apply_QtY(), apply_YQ().
#ifndef MISPEC_SPECTRA_DOUBLE_SHIFT_QR_H
#define MISPEC_SPECTRA_DOUBLE_SHIFT_QR_H

#include <stdexcept>
#include <vector>

#include "../internal/Dense.h"
#include "../internal/SmallDenseGen.h"

namespace Spectra {

template <typename Scalar = double>
class DoubleShiftQR
{
    using Matrix = DenseMatrix<Scalar>;
    Index m_n = 0;
    std::vector<double> m_H;  // Q'HQ
    std::vector<double> m_Q;  // explicit Q (n x n): the matrices here are a few dozen rows
    bool m_computed = false;

public:
    explicit DoubleShiftQR(Index size = 0) : m_n(size) {}
    DoubleShiftQR(const Matrix& mat, const Scalar& s, const Scalar& t) { compute(mat, s, t); }

    void compute(const Matrix& mat, const Scalar& s, const Scalar& t)
    {
        m_n = mat.rows();
        if (m_n != mat.cols())
            throw std::invalid_argument("DoubleShiftQR: matrix must be square");
        const int n = static_cast<int>(m_n);
        m_H.assign(mat.data(), mat.data() + std::size_t(n) * n);
        m_Q.assign(std::size_t(n) * n, 0.0);
        for (int i = 0; i < n; i++)
            m_Q[std::size_t(i) * n + i] = 1.0;
        mispec::small::DoubleShiftStep(n, m_H.data(), n, s, t, m_Q.data(), n, n);
        m_computed = true;
    }

    void matrix_QtHQ(Matrix& dest) const
    {
        if (!m_computed)
            throw std::logic_error("DoubleShiftQR: need to call compute() first");
        dest.resize(m_n, m_n);
        std::copy(m_H.begin(), m_H.end(), dest.data());
    }

    // y <- Q' y   (reference :410-422)
    void apply_QtY(DenseVector<Scalar>& y) const
    {
        if (!m_computed)
            throw std::logic_error("DoubleShiftQR: need to call compute() first");
        std::vector<double> out(static_cast<std::size_t>(m_n));
        for (Index j = 0; j < m_n; j++)
        {
            double acc = 0.0;
            for (Index k = 0; k < m_n; k++)
                acc += m_Q[std::size_t(j) * m_n + k] * y[k];
            out[std::size_t(j)] = acc;
        }
        for (Index j = 0; j < m_n; j++)
            y[j] = out[std::size_t(j)];
    }

    // Y <- Y * Q
    void apply_YQ(Matrix& Y) const
    {
        if (!m_computed)
            throw std::logic_error("DoubleShiftQR: need to call compute() first");
        const Index nrow = Y.rows();
        std::vector<double> row(static_cast<std::size_t>(m_n));
        for (Index r = 0; r < nrow; r++)
        {
            for (Index j = 0; j < m_n; j++)
            {
                double acc = 0.0;
                for (Index k = 0; k < m_n; k++)
                    acc += Y(r, k) * m_Q[std::size_t(j) * m_n + k];
                row[std::size_t(j)] = acc;
            }
            for (Index j = 0; j < m_n; j++)
                Y(r, j) = row[std::size_t(j)];
        }
    }
};

}  // namespace Spectra

#endif
