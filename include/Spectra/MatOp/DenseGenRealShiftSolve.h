// y = (A - sigma I)^{-1} x for a general real dense A and a real shift — the operator of GenEigsRealShiftSolver for
// dense matrices.  Same template signature and members as the reference class (MatOp/DenseGenRealShiftSolve.h:27-100),
// which factors with Eigen::PartialPivLU; here the matrix goes through SparseGenRealShiftSolve's device path (LU with
// partial pivoting, explicit inverse in HBM, GEMV kernel; n <= 4096).
#ifndef MISPEC_SPECTRA_DENSE_GEN_REAL_SHIFT_SOLVE_H
#define MISPEC_SPECTRA_DENSE_GEN_REAL_SHIFT_SOLVE_H

#include <stdexcept>

#include "../internal/DenseToSparse.h"
#include "SparseGenRealShiftSolve.h"

namespace Spectra {

template <typename Scalar_, int Flags = ColMajor>
class DenseGenRealShiftSolve : public SparseGenRealShiftSolve<Scalar_, ColMajor, int>
{
    using Base = SparseGenRealShiftSolve<Scalar_, ColMajor, int>;

    static const DenseView<Scalar_>& checked(const DenseView<Scalar_>& A)
    {
        if (A.rows != A.cols)
            throw std::invalid_argument("DenseGenRealShiftSolve: matrix must be square");
        if (A.row_major != (Flags == RowMajor))
            throw std::invalid_argument(
                "DenseGenRealShiftSolve: the \"Flags\" template parameter does not match the input matrix (ColMajor/RowMajor)");
        return A;
    }

public:
    using Scalar = Scalar_;

    explicit DenseGenRealShiftSolve(const DenseView<Scalar>& mat, internal::CtxPtr ctx = internal::CtxPtr()) :
        Base(internal::CompressedCopy(checked(mat)).view(), ctx)
    {}
#ifdef MISPEC_HAVE_EIGEN
    // The reference's constructor: an Eigen matrix or Map of matching storage order (copied once on the host)
    template <typename Derived>
    DenseGenRealShiftSolve(const Eigen::MatrixBase<Derived>& mat) :
        DenseGenRealShiftSolve(eigen_view(Eigen::Matrix<Scalar, Eigen::Dynamic, Eigen::Dynamic, Flags>(mat)))
    {
        static_assert(static_cast<int>(Derived::PlainObject::IsRowMajor) == (Flags == RowMajor ? 1 : 0),
                      "DenseGenRealShiftSolve: the \"Flags\" template parameter does not match the input matrix");
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
