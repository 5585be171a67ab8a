// y = A x for a complex Hermitian dense A of which ONE triangle is read (reference: MatOp/DenseHermMatProd.h — the same
// members as DenseSymMatProd, `mat.selfadjointView<Uplo>() * x`).  The `Uplo` triangle is mirrored (conjugated, the diagonal taken
// real) into a full matrix in HBM at construction.  Complex scalars are outside the hot path; for a real
// scalar this is DenseSymMatProd.
#ifndef MISPEC_SPECTRA_DENSE_HERM_MAT_PROD_H
#define MISPEC_SPECTRA_DENSE_HERM_MAT_PROD_H

#include "../internal/ComplexDense.h"
#include "DenseSymMatProd.h"

namespace Spectra {

template <typename Scalar_, int Uplo = Lower, int Flags = ColMajor>
class DenseHermMatProd : public DenseSymMatProd<Scalar_, Uplo, Flags>
{
public:
    using DenseSymMatProd<Scalar_, Uplo, Flags>::DenseSymMatProd;
};

template <int Uplo, int Flags>
class DenseHermMatProd<std::complex<double>, Uplo, Flags> : public internal::ComplexDenseOp
{
    static_assert(Uplo == Lower || Uplo == Upper, "Uplo must be Lower or Upper");

public:
    using Scalar = std::complex<double>;

    explicit DenseHermMatProd(const DenseView<Scalar>& mat, internal::CtxPtr ctx = internal::CtxPtr()) : internal::ComplexDenseOp(ctx)
    {
        ingest(mat, Flags == RowMajor, Uplo == Lower ? 'L' : 'U', "DenseHermMatProd");
    }
#ifdef MISPEC_HAVE_EIGEN
    template <typename Derived>
    DenseHermMatProd(const Eigen::MatrixBase<Derived>& mat) : internal::ComplexDenseOp(internal::CtxPtr())
    {
        using Plain = Eigen::Matrix<Scalar, Eigen::Dynamic, Eigen::Dynamic, Flags>;
        static_assert(static_cast<int>(Derived::PlainObject::IsRowMajor) == static_cast<int>(Plain::IsRowMajor),
                      "DenseHermMatProd: the \"Flags\" template parameter does not match the input matrix");
        const Plain tmp(mat);
        ingest(DenseView<Scalar>(tmp.rows(), tmp.cols(), tmp.data(), tmp.outerStride(), Plain::IsRowMajor), Flags == RowMajor,
               Uplo == Lower ? 'L' : 'U', "DenseHermMatProd");
    }
#endif
};

}  // namespace Spectra

#endif
