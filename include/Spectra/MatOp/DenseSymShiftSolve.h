// y = (A - sigma I)^{-1} x for a real symmetric dense A of which the `Uplo` triangle is read — the operator of
// SymEigsShiftSolver for dense matrices.  Same template signature and members as the reference class
// (MatOp/DenseSymShiftSolve.h:28-107: rows(), cols(), set_shift(), perform_op()).  The reference factors with its own
// Bunch-Kaufman LDL' (LinAlg/BKLDLT.h); here the matrix goes through the device factorisation of SparseSymShiftSolve
// (dense n <= 4096: LU with partial pivoting on the host, explicit inverse applied by a GEMV kernel; banded: LDL' on
// the device), with the same error behaviour: std::invalid_argument "factorization failed with the given shift".
#ifndef MISPEC_SPECTRA_DENSE_SYM_SHIFT_SOLVE_H
#define MISPEC_SPECTRA_DENSE_SYM_SHIFT_SOLVE_H

#include <stdexcept>

#include "../internal/DenseToSparse.h"
#include "SparseSymShiftSolve.h"

namespace Spectra {

template <typename Scalar_, int Uplo = Lower, int Flags = ColMajor>
class DenseSymShiftSolve : public SparseSymShiftSolve<Scalar_, Uplo, ColMajor, int>
{
    using Base = SparseSymShiftSolve<Scalar_, Uplo, ColMajor, int>;

    static const DenseView<Scalar_>& checked(const DenseView<Scalar_>& A)
    {
        if (A.rows != A.cols)
            throw std::invalid_argument("DenseSymShiftSolve: matrix must be square");
        if (A.row_major != (Flags == RowMajor))
            throw std::invalid_argument(
                "DenseSymShiftSolve: the \"Flags\" template parameter does not match the input matrix (ColMajor/RowMajor)");
        return A;
    }

public:
    using Scalar = Scalar_;

    explicit DenseSymShiftSolve(const DenseView<Scalar>& mat, internal::CtxPtr ctx = internal::CtxPtr()) :
        Base(internal::CompressedCopy(checked(mat)).view(), ctx)
    {}

#ifdef MISPEC_HAVE_EIGEN
    // The reference's constructor: an Eigen matrix or Map of matching storage order (copied once on the host)
    template <typename Derived>
    DenseSymShiftSolve(const Eigen::MatrixBase<Derived>& mat) :
        DenseSymShiftSolve(eigen_view(Eigen::Matrix<Scalar, Eigen::Dynamic, Eigen::Dynamic, Flags>(mat)))
    {
        static_assert(static_cast<int>(Derived::PlainObject::IsRowMajor) == (Flags == RowMajor ? 1 : 0),
                      "DenseSymShiftSolve: the \"Flags\" template parameter does not match the input matrix");
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
