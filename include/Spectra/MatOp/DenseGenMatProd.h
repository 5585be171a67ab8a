// y = A x for a general real dense A.  Same template signature and members as the reference class
// (MatOp/DenseGenMatProd.h:27-102): Scalar, rows(), cols(), perform_op(), operator*, operator().
// The matrix is copied to HBM once (row-major there, whatever the input order); the solvers bind the device
// matrix and never call perform_op, which keeps the reference's host-pointer contract for other callers.
#ifndef MISPEC_SPECTRA_DENSE_GEN_MAT_PROD_H
#define MISPEC_SPECTRA_DENSE_GEN_MAT_PROD_H

#include "../../mispec_extras.h"  // outside the hot path of SURVEY.md section 8: declared apart from the thin shim
#include <stdexcept>
#include <type_traits>

#include "../internal/ComplexDense.h"
#include "../internal/Dense.h"
#include "../internal/Device.h"

namespace Spectra {

template <typename Scalar_, int Flags = ColMajor>
class DenseGenMatProd
{
public:
    using Scalar = Scalar_;

private:
    static_assert(std::is_same<Scalar_, double>::value, "the MI355X path computes in fp64: Scalar must be double");
    using Matrix = DenseMatrix<Scalar>;

    internal::CtxPtr m_ctx;
    std::shared_ptr<mispec_dense> m_mat;

    void ingest(const DenseView<Scalar>& A)
    {
        if (A.row_major != (Flags == RowMajor))
            throw std::invalid_argument(
                "DenseGenMatProd: the \"Flags\" template parameter does not match the input matrix (ColMajor/RowMajor)");
        mispec_dense* raw = nullptr;
        internal::check(mispec_dense_upload(m_ctx.get(), A.rows, A.cols, A.data, A.ld, A.row_major ? 1 : 0, 0, &raw));
        m_mat = std::shared_ptr<mispec_dense>(raw, [](mispec_dense* p) { (void) mispec_dense_destroy(p); });
    }

public:
    explicit DenseGenMatProd(const DenseView<Scalar>& mat, internal::CtxPtr ctx = internal::CtxPtr()) :
        m_ctx(ctx ? ctx : internal::default_context())
    {
        ingest(mat);
    }

#ifdef MISPEC_HAVE_EIGEN
    // The reference's constructor (DenseGenMatProd.h:55-62)
    template <typename Derived>
    DenseGenMatProd(const Eigen::MatrixBase<Derived>& mat) : m_ctx(internal::default_context())
    {
        using Plain = Eigen::Matrix<Scalar, Eigen::Dynamic, Eigen::Dynamic, Flags>;
        static_assert(static_cast<int>(Derived::PlainObject::IsRowMajor) == static_cast<int>(Plain::IsRowMajor),
                      "DenseGenMatProd: the \"Flags\" template parameter does not match the input matrix");
        const Plain tmp(mat);
        ingest(DenseView<Scalar>(tmp.rows(), tmp.cols(), tmp.data(), tmp.outerStride(), Plain::IsRowMajor));
    }
#endif

    Index rows() const { return static_cast<Index>(mispec_dense_rows(m_mat.get())); }
    Index cols() const { return static_cast<Index>(mispec_dense_cols(m_mat.get())); }

    // y_out = A * x_in, host pointers (DenseGenMatProd.h:78-83)
    void perform_op(const Scalar* x_in, Scalar* y_out) const { internal::check(mispec_dense_gemv_host(m_mat.get(), x_in, y_out)); }

    // Y = A * X (DenseGenMatProd.h:88-91)
    Matrix operator*(const Matrix& mat_in) const
    {
        Matrix res(rows(), mat_in.cols());
        internal::check(mispec_dense_gemm_host(m_mat.get(), mat_in.data(), mat_in.rows(), static_cast<int>(mat_in.cols()), res.data(),
                                               res.rows()));
        return res;
    }

    // A(i, j) (DenseGenMatProd.h:96-99)
    Scalar operator()(Index i, Index j) const
    {
        Scalar v = 0;
        internal::check(mispec_dense_coeff(m_mat.get(), i, j, &v));
        return v;
    }

    mispec_ctx* mispec_context() const { return m_ctx.get(); }
    const mispec_dense* mispec_dense_matrix() const { return m_mat.get(); }
};

// Complex general dense A (the reference instantiates the same template with std::complex, test/Arnoldi.cpp:122-138): the
// matrix in HBM as interleaved (re, im) pairs; outside the hot path of SURVEY.md section 8.
template <int Flags>
class DenseGenMatProd<std::complex<double>, Flags> : public internal::ComplexDenseOp
{
public:
    using Scalar = std::complex<double>;

    explicit DenseGenMatProd(const DenseView<Scalar>& mat, internal::CtxPtr ctx = internal::CtxPtr()) : internal::ComplexDenseOp(ctx)
    {
        ingest(mat, Flags == RowMajor, 0, "DenseGenMatProd");
    }
#ifdef MISPEC_HAVE_EIGEN
    template <typename Derived>
    DenseGenMatProd(const Eigen::MatrixBase<Derived>& mat) : internal::ComplexDenseOp(internal::CtxPtr())
    {
        using Plain = Eigen::Matrix<Scalar, Eigen::Dynamic, Eigen::Dynamic, Flags>;
        static_assert(static_cast<int>(Derived::PlainObject::IsRowMajor) == static_cast<int>(Plain::IsRowMajor),
                      "DenseGenMatProd: the \"Flags\" template parameter does not match the input matrix");
        const Plain tmp(mat);
        ingest(DenseView<Scalar>(tmp.rows(), tmp.cols(), tmp.data(), tmp.outerStride(), Plain::IsRowMajor), Flags == RowMajor, 0,
               "DenseGenMatProd");
    }
#endif
};

}  // namespace Spectra

#endif
