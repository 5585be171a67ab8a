// The Cholesky operator B = L L' of the generalized solver for a dense positive-definite B.  Same template signature
// and members as the reference class (MatOp/DenseCholesky.h:29-105: rows(), cols(), info(), lower_triangular_solve(),
// upper_triangular_solve()); the reference wraps Eigen::LLT, here the matrix goes through SparseCholesky's device factor
// (dense Cholesky on the host, L^{-1} / L^{-T} applied by GEMV kernels; n <= 4096).
#ifndef MISPEC_SPECTRA_DENSE_CHOLESKY_H
#define MISPEC_SPECTRA_DENSE_CHOLESKY_H

#include <stdexcept>

#include "../internal/DenseToSparse.h"
#include "SparseCholesky.h"

namespace Spectra {

template <typename Scalar_, int Uplo = Lower, int Flags = ColMajor>
class DenseCholesky : public SparseCholesky<Scalar_, Uplo, ColMajor, int>
{
    using Base = SparseCholesky<Scalar_, Uplo, ColMajor, int>;

    static const DenseView<Scalar_>& checked(const DenseView<Scalar_>& A)
    {
        if (A.rows != A.cols)
            throw std::invalid_argument("DenseCholesky: matrix must be square");
        if (A.row_major != (Flags == RowMajor))
            throw std::invalid_argument(
                "DenseCholesky: the \"Flags\" template parameter does not match the input matrix (ColMajor/RowMajor)");
        return A;
    }

public:
    using Scalar = Scalar_;

    explicit DenseCholesky(const DenseView<Scalar>& mat, internal::CtxPtr ctx = internal::CtxPtr()) :
        Base(internal::CompressedCopy(checked(mat)).view(), ctx)
    {}
#ifdef MISPEC_HAVE_EIGEN
    // The reference's constructor: an Eigen matrix or Map of matching storage order (copied once on the host)
    template <typename Derived>
    DenseCholesky(const Eigen::MatrixBase<Derived>& mat) :
        DenseCholesky(eigen_view(Eigen::Matrix<Scalar, Eigen::Dynamic, Eigen::Dynamic, Flags>(mat)))
    {
        static_assert(static_cast<int>(Derived::PlainObject::IsRowMajor) == (Flags == RowMajor ? 1 : 0),
                      "DenseCholesky: the \"Flags\" template parameter does not match the input matrix");
    }

private:
    static DenseView<Scalar> eigen_view(const Eigen::Matrix<Scalar, Eigen::Dynamic, Eigen::Dynamic, Flags>& m)
    {
        return DenseView<Scalar>(m.rows(), m.cols(), m.data(), m.outerStride(), Flags == RowMajor);
    }
#endif
};

}  // namespace Spectra

#endif
