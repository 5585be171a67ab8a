// Eigen-decomposition of a symmetric tridiagonal matrix by implicit Wilkinson-shift QR, host side
// (reference: LinAlg/TridiagEigen.h:24-230).  Same members: compute(), eigenvalues(), eigenvectors().
// The arithmetic is internal/SmallDense.h::tridiag_eigen, shared with the device kernel the symmetric
// solver uses (csrc/small.hip k_tridiag_eigen).
#ifndef MISPEC_SPECTRA_TRIDIAG_EIGEN_H
#define MISPEC_SPECTRA_TRIDIAG_EIGEN_H

#include <stdexcept>
#include <vector>

#include "../internal/Dense.h"
#include "../internal/SmallDense.h"

namespace Spectra {

template <typename Scalar = double>
class TridiagEigen
{
    using Matrix = DenseMatrix<Scalar>;
    using Vector = DenseVector<Scalar>;
    Index m_n = 0;
    Vector m_evals;
    Matrix m_evecs;
    bool m_computed = false;

public:
    TridiagEigen() {}
    explicit TridiagEigen(const Matrix& mat) { compute(mat); }

    // Only the diagonal and the sub-diagonal of mat are read.
    void compute(const Matrix& mat)
    {
        m_n = mat.rows();
        if (m_n != mat.cols())
            throw std::invalid_argument("TridiagEigen: matrix must be square");
        const int n = static_cast<int>(m_n);
        std::vector<double> subd(static_cast<std::size_t>(n > 0 ? n : 1), 0.0);
        m_evals.resize(m_n);
        m_evecs.resize(m_n, m_n);
        for (int i = 0; i < n; i++)
        {
            m_evals[i] = mat(i, i);
            for (int j = 0; j < n; j++)
                m_evecs(i, j) = (i == j) ? Scalar(1) : Scalar(0);
        }
        for (int i = 0; i < n - 1; i++)
            subd[std::size_t(i)] = mat(i + 1, i);
        if (mispec::small::tridiag_eigen(n, m_evals.data(), subd.data(), m_evecs.data(), n, mispec::small::Lanes{0, 1}) != 0)
            throw std::runtime_error("TridiagEigen: eigen decomposition failed");
        m_computed = true;
    }

    const Vector& eigenvalues() const
    {
        if (!m_computed)
            throw std::logic_error("TridiagEigen: need to call compute() first");
        return m_evals;
    }
    const Matrix& eigenvectors() const
    {
        if (!m_computed)
            throw std::logic_error("TridiagEigen: need to call compute() first");
        return m_evecs;
    }
};

}  // namespace Spectra

#endif
