// y = A x for a general real dense A.  Same template signature and members as the reference class
// (MatOp/DenseGenMatProd.h:27-102): Scalar, rows(), cols(), perform_op(), operator*, operator().
// The matrix is copied to HBM once (row-major there, whatever the input order); the solvers bind the device
// matrix and never call perform_op, which keeps the reference's host-pointer contract for other callers.
// Scalar = float is accepted at this boundary (test/DenseGenMatProd.cpp:12): widened to the device's fp64, rounded once on the way out.
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
    static_assert(internal::is_device_scalar<Scalar_>::value, "Scalar must be double (or float, widened: the MI355X path computes in fp64)");
    using Matrix = DenseMatrix<Scalar>;

    internal::CtxPtr m_ctx;
    std::shared_ptr<mispec_dense> m_mat;

    void ingest(const DenseView<Scalar>& A)
    {
        if (A.row_major != (Flags == RowMajor))
            throw std::invalid_argument(
                "DenseGenMatProd: the \"Flags\" template parameter does not match the input matrix (ColMajor/RowMajor)");
        mispec_dense* raw = nullptr;
        const Index outer = A.row_major ? A.rows : A.cols, inner = A.row_major ? A.cols : A.rows;
        const internal::WidenedIn<Scalar> values(A.data, static_cast<std::size_t>(outer > 0 ? (outer - 1) * A.ld + inner : 0));
        internal::check(mispec_dense_upload(m_ctx.get(), A.rows, A.cols, values.data(), A.ld, A.row_major ? 1 : 0, 0, &raw));
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
    void perform_op(const Scalar* x_in, Scalar* y_out) const
    {
        const internal::WidenedIn<Scalar> x(x_in, static_cast<std::size_t>(cols()));
        internal::NarrowedOut<Scalar> y(y_out, static_cast<std::size_t>(rows()));
        internal::check(mispec_dense_gemv_host(m_mat.get(), x.data(), y.data()));
        y.store();
    }

    // Y = A * X (DenseGenMatProd.h:88-91)
    Matrix operator*(const Matrix& mat_in) const
    {
        Matrix res(rows(), mat_in.cols());
        const internal::WidenedIn<Scalar> in(mat_in.data(), static_cast<std::size_t>(mat_in.rows() * mat_in.cols()));
        internal::NarrowedOut<Scalar> out(res.data(), static_cast<std::size_t>(res.rows() * res.cols()));
        internal::check(mispec_dense_gemm_host(m_mat.get(), in.data(), mat_in.rows(), static_cast<int>(mat_in.cols()), out.data(), res.rows()));
        out.store();
        return res;
    }

    // A(i, j) (DenseGenMatProd.h:96-99)
    Scalar operator()(Index i, Index j) const
    {
        double v = 0;
        internal::check(mispec_dense_coeff(m_mat.get(), i, j, &v));
        return static_cast<Scalar>(v);
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
