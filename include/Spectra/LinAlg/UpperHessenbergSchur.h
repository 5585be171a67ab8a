// Real Schur decomposition H = U T U' of an upper Hessenberg matrix, host side
// (reference: LinAlg/UpperHessenbergSchur.h:24-456, itself adapted from Eigen's RealSchur).
#ifndef MISPEC_SPECTRA_UPPER_HESSENBERG_SCHUR_H
#define MISPEC_SPECTRA_UPPER_HESSENBERG_SCHUR_H

#include <stdexcept>
#include <utility>

#include "../internal/Dense.h"
#include "../internal/SmallDenseGen.h"

namespace Spectra {

template <typename Scalar = double>
class UpperHessenbergSchur
{
    using Matrix = DenseMatrix<Scalar>;
    Index m_n = 0;
    Matrix m_T, m_U;
    bool m_computed = false;

public:
    UpperHessenbergSchur() {}
    explicit UpperHessenbergSchur(const Matrix& mat) { compute(mat); }

    void compute(const Matrix& mat)
    {
        if (mat.rows() != mat.cols())
            throw std::invalid_argument("UpperHessenbergSchur: matrix must be square");
        m_n = mat.rows();
        m_T = mat;
        m_U.resize(m_n, m_n);
        const int n = static_cast<int>(m_n);
        if (!mispec::small::hess_real_schur(n, m_T.data(), n, m_U.data(), n))
            throw std::runtime_error("UpperHessenbergSchur: Schur decomposition failed");
        m_computed = true;
    }

    const Matrix& matrix_T() const
    {
        if (!m_computed)
            throw std::logic_error("UpperHessenbergSchur: need to call compute() first");
        return m_T;
    }
    const Matrix& matrix_U() const
    {
        if (!m_computed)
            throw std::logic_error("UpperHessenbergSchur: need to call compute() first");
        return m_U;
    }

    // Hand the results over without a copy (reference :443-451)
    void swap_T(Matrix& other) { std::swap(m_T, other); }
    void swap_U(Matrix& other) { std::swap(m_U, other); }
};

}  // namespace Spectra

#endif
