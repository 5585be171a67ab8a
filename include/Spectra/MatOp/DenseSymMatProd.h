// y = A x for a real symmetric dense A of which ONE triangle is read.  Same template signature and members as the
// reference class (MatOp/DenseSymMatProd.h:28-105): Scalar, rows(), cols(), perform_op(), operator*, operator().
//
// Differences that a user can observe:
//   * the matrix is copied to the GPU at construction (the reference keeps an Eigen::Ref): the `Uplo` triangle is
//     mirrored into a full row-major matrix in HBM; the other triangle of the input is never read, exactly as
//     selfadjointView<Uplo> ignores it;
//   * operator()(i, j) answers from the mirrored matrix, so both triangles return the coefficient of the
//     symmetric operator (the reference returns whatever the input stores at (i, j));
//   * perform_op(x_in, y_out) keeps the host-pointer contract (staged H2D / GEMV kernel / D2H); the solvers do not
//     go through it — they bind the device matrix and keep the Krylov basis in HBM.
// Scalar = float is accepted at this boundary (test/DenseSymMatProd.cpp:12): widened to the device's fp64, rounded once on the way out.
#ifndef MISPEC_SPECTRA_DENSE_SYM_MAT_PROD_H
#define MISPEC_SPECTRA_DENSE_SYM_MAT_PROD_H

#include "../../mispec_extras.h"  // outside the hot path of SURVEY.md section 8: declared apart from the thin shim
#include <stdexcept>
#include <type_traits>

#include "../internal/Dense.h"
#include "../internal/Device.h"

namespace Spectra {

template <typename Scalar_, int Uplo = Lower, int Flags = ColMajor>
class DenseSymMatProd
{
public:
    using Scalar = Scalar_;

private:
    static_assert(internal::is_device_scalar<Scalar_>::value, "Scalar must be double (or float, widened: the MI355X path computes in fp64)");
    static_assert(Uplo == Lower || Uplo == Upper, "Uplo must be Lower or Upper");
    using Matrix = DenseMatrix<Scalar>;

    internal::CtxPtr m_ctx;
    std::shared_ptr<mispec_dense> m_mat;

    void ingest(const DenseView<Scalar>& A)
    {
        if (A.rows != A.cols)
            throw std::invalid_argument("DenseSymMatProd: matrix must be square");
        if (A.row_major != (Flags == RowMajor))
            throw std::invalid_argument(
                "DenseSymMatProd: the \"Flags\" template parameter does not match the input matrix (ColMajor/RowMajor)");
        mispec_dense* raw = nullptr;
        const Index outer = A.row_major ? A.rows : A.cols, inner = A.row_major ? A.cols : A.rows;
        const internal::WidenedIn<Scalar> values(A.data, static_cast<std::size_t>(outer > 0 ? (outer - 1) * A.ld + inner : 0));
        internal::check(mispec_dense_upload(m_ctx.get(), A.rows, A.cols, values.data(), A.ld, A.row_major ? 1 : 0, Uplo == Lower ? 'L' : 'U', &raw));
        m_mat = std::shared_ptr<mispec_dense>(raw, [](mispec_dense* p) { (void) mispec_dense_destroy(p); });
    }

public:
    // From a dense matrix in host memory.
    explicit DenseSymMatProd(const DenseView<Scalar>& mat, internal::CtxPtr ctx = internal::CtxPtr()) :
        m_ctx(ctx ? ctx : internal::default_context())
    {
        ingest(mat);
    }

#ifdef MISPEC_HAVE_EIGEN
    // The reference's constructor (DenseSymMatProd.h:58-65): an Eigen matrix or Map of matching storage order.
    template <typename Derived>
    DenseSymMatProd(const Eigen::MatrixBase<Derived>& mat) : m_ctx(internal::default_context())
    {
        using Plain = Eigen::Matrix<Scalar, Eigen::Dynamic, Eigen::Dynamic, Flags>;
        static_assert(static_cast<int>(Derived::PlainObject::IsRowMajor) == static_cast<int>(Plain::IsRowMajor),
                      "DenseSymMatProd: the \"Flags\" template parameter does not match the input matrix");
        const Plain tmp(mat);
        ingest(DenseView<Scalar>(tmp.rows(), tmp.cols(), tmp.data(), tmp.outerStride(), Plain::IsRowMajor));
    }
#endif

    Index rows() const { return static_cast<Index>(mispec_dense_rows(m_mat.get())); }
    Index cols() const { return static_cast<Index>(mispec_dense_cols(m_mat.get())); }

    // y_out = A * x_in, host pointers (DenseSymMatProd.h:81-86)
    void perform_op(const Scalar* x_in, Scalar* y_out) const
    {
        const internal::WidenedIn<Scalar> x(x_in, static_cast<std::size_t>(cols()));
        internal::NarrowedOut<Scalar> y(y_out, static_cast<std::size_t>(rows()));
        internal::check(mispec_dense_gemv_host(m_mat.get(), x.data(), y.data()));
        y.store();
    }

    // Y = A * X for a dense block (DenseSymMatProd.h:91-94)
    Matrix operator*(const Matrix& mat_in) const
    {
        Matrix res(rows(), mat_in.cols());
        const internal::WidenedIn<Scalar> in(mat_in.data(), static_cast<std::size_t>(mat_in.rows() * mat_in.cols()));
        internal::NarrowedOut<Scalar> out(res.data(), static_cast<std::size_t>(res.rows() * res.cols()));
        internal::check(mispec_dense_gemm_host(m_mat.get(), in.data(), mat_in.rows(), static_cast<int>(mat_in.cols()), out.data(), res.rows()));
        out.store();
        return res;
    }

    // A(i, j) of the symmetric operator (DenseSymMatProd.h:99-102)
    Scalar operator()(Index i, Index j) const
    {
        double v = 0;
        internal::check(mispec_dense_coeff(m_mat.get(), i, j, &v));
        return static_cast<Scalar>(v);
    }

    // Device binding used by the solvers' fast path.
    mispec_ctx* mispec_context() const { return m_ctx.get(); }
    const mispec_dense* mispec_dense_matrix() const { return m_mat.get(); }
};

}  // namespace Spectra

#endif
