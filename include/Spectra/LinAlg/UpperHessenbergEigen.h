// Eigen-decomposition of a real upper Hessenberg matrix (complex eigenvalues / eigenvectors), host side
// (reference: LinAlg/UpperHessenbergEigen.h:28-320): real Schur form, back-substitution, unit columns.
#ifndef MISPEC_SPECTRA_UPPER_HESSENBERG_EIGEN_H
#define MISPEC_SPECTRA_UPPER_HESSENBERG_EIGEN_H

#include <complex>
#include <stdexcept>
#include <vector>

#include "../internal/Dense.h"
#include "../internal/SmallDenseComplex.h"
#include "../internal/SmallDenseGen.h"

namespace Spectra {

template <typename Scalar = double>
class UpperHessenbergEigen
{
    using Matrix = DenseMatrix<Scalar>;
    using Complex = std::complex<Scalar>;
    using ComplexMatrix = DenseMatrix<Complex>;
    using ComplexVector = DenseVector<Complex>;
    Index m_n = 0;
    ComplexVector m_evals;
    ComplexMatrix m_evecs;
    bool m_computed = false;

public:
    UpperHessenbergEigen() {}
    explicit UpperHessenbergEigen(const Matrix& mat) { compute(mat); }

    void compute(const Matrix& mat)
    {
        if (mat.rows() != mat.cols())
            throw std::invalid_argument("UpperHessenbergEigen: matrix must be square");
        m_n = mat.rows();
        m_evals.resize(m_n);
        m_evecs.resize(m_n, m_n);
        const int n = static_cast<int>(m_n);
        mispec::small::hess_eigen(n, mat.data(), n, m_evals.data(), m_evecs.data());
        m_computed = true;
    }

    const ComplexVector& eigenvalues() const
    {
        if (!m_computed)
            throw std::logic_error("UpperHessenbergEigen: need to call compute() first");
        return m_evals;
    }
    ComplexMatrix eigenvectors() const
    {
        if (!m_computed)
            throw std::logic_error("UpperHessenbergEigen: need to call compute() first");
        return m_evecs;
    }
};

// Complex upper Hessenberg matrices (reference: LinAlg/UpperHessenbergEigen.h:325-455): complex Schur form, back
// substitution, unit columns, eigenvalues in increasing modulus.  Host arithmetic only.
template <typename RealScalar>
class UpperHessenbergEigen<std::complex<RealScalar>>
{
    using Scalar = std::complex<RealScalar>;
    using Matrix = DenseMatrix<Scalar>;
    using ComplexMatrix = Matrix;
    using ComplexVector = DenseVector<Scalar>;
    Index m_n = 0;
    ComplexVector m_evals;
    ComplexMatrix m_evecs;
    bool m_computed = false;

public:
    UpperHessenbergEigen() {}
    explicit UpperHessenbergEigen(const Matrix& mat) { compute(mat); }

    void compute(const Matrix& mat)
    {
        if (mat.rows() != mat.cols())
            throw std::invalid_argument("UpperHessenbergEigen: matrix must be square");
        m_n = mat.rows();
        const int n = static_cast<int>(m_n);
        using cd = std::complex<double>;
        std::vector<cd> H(std::size_t(n) * n), vals(static_cast<std::size_t>(n)), vecs(std::size_t(n) * n);
        for (int j = 0; j < n; j++)
            for (int i = 0; i < n; i++)
                H[std::size_t(j) * n + i] = (i <= j + 1) ? cd(mat(i, j)) : cd(0.0);
        mispec::small::hess_eigen_complex(n, H.data(), n, vals.data(), vecs.data());
        m_evals.resize(m_n);
        m_evecs.resize(m_n, m_n);
        for (int j = 0; j < n; j++)
        {
            m_evals[j] = Scalar(vals[std::size_t(j)]);
            for (int i = 0; i < n; i++)
                m_evecs(i, j) = Scalar(vecs[std::size_t(j) * n + i]);
        }
        m_computed = true;
    }

    const ComplexVector& eigenvalues() const
    {
        if (!m_computed)
            throw std::logic_error("UpperHessenbergEigen: need to call compute() first");
        return m_evals;
    }
    const ComplexMatrix& eigenvectors() const
    {
        if (!m_computed)
            throw std::logic_error("UpperHessenbergEigen: need to call compute() first");
        return m_evecs;
    }
};

}  // namespace Spectra

#endif
