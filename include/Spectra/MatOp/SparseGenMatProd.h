// y = A x for a general real sparse A — the operator GenEigsSolver is normally used with.  Same template
// signature and members as the reference class (MatOp/SparseGenMatProd.h:28-105).  As for
// SparseSymMatProd the matrix is copied to HBM at construction (CSC input is transposed to CSR once)
// and the solvers bind the device matrix directly instead of calling perform_op().
// Scalar = float is accepted at this boundary like the reference's class (test/SparseGenMatProd.cpp:36): widened to the device's
// fp64 on the way in, rounded once on the way out; the solvers themselves require double.
#ifndef MISPEC_SPECTRA_SPARSE_GEN_MAT_PROD_H
#define MISPEC_SPECTRA_SPARSE_GEN_MAT_PROD_H

#include <type_traits>

#include "../internal/Dense.h"
#include "../internal/Device.h"

namespace Spectra {

template <typename Scalar_, int Flags = ColMajor, typename StorageIndex = int>
class SparseGenMatProd
{
public:
    using Scalar = Scalar_;

private:
    static_assert(internal::is_device_scalar<Scalar_>::value, "Scalar must be double (or float, widened: the MI355X path computes in fp64)");
    using Matrix = DenseMatrix<Scalar>;

    internal::CtxPtr m_ctx;
    std::shared_ptr<mispec_csr> m_mat;

    void ingest(const SparseView<Scalar, StorageIndex>& A)
    {
        if (A.row_major != (Flags == RowMajor))
            throw std::invalid_argument(
                "SparseGenMatProd: the \"Flags\" template parameter does not match the input matrix (ColMajor/RowMajor)");
        mispec_csr* raw = nullptr;
        const std::size_t nouter = static_cast<std::size_t>(A.row_major ? A.rows : A.cols);
        const std::size_t nnz = static_cast<std::size_t>(A.outer[nouter]);
        const internal::WidenedIn<Scalar> values(A.values, nnz);
        const internal::Int32Indices<StorageIndex> outer(A.outer, nouter + 1), inner(A.inner, nnz);  // any StorageIndex (reference: :22)
        if (A.row_major)
            internal::check(mispec_csr_upload(m_ctx.get(), A.rows, A.cols, outer.data(), inner.data(), values.data(), &raw));
        else
            internal::check(mispec_csr_from_csc(m_ctx.get(), A.rows, A.cols, outer.data(), inner.data(), values.data(), &raw));
        m_mat = std::shared_ptr<mispec_csr>(raw, [](mispec_csr* p) { (void) mispec_csr_destroy(p); });
    }

public:
    explicit SparseGenMatProd(const SparseView<Scalar, StorageIndex>& mat, internal::CtxPtr ctx = internal::CtxPtr()) :
        m_ctx(ctx ? ctx : internal::default_context())
    {
        ingest(mat);
    }

#ifdef MISPEC_HAVE_EIGEN
    template <typename Derived>
    SparseGenMatProd(const Eigen::SparseMatrixBase<Derived>& mat) : m_ctx(internal::default_context())
    {
        using Plain = Eigen::SparseMatrix<Scalar, Flags, StorageIndex>;
        static_assert(static_cast<int>(Derived::PlainObject::IsRowMajor) == static_cast<int>(Plain::IsRowMajor),
                      "SparseGenMatProd: the \"Flags\" template parameter does not match the input matrix");
        Plain tmp(mat);
        tmp.makeCompressed();
        SparseView<Scalar, StorageIndex> v;
        v.rows = tmp.rows();
        v.cols = tmp.cols();
        v.outer = tmp.outerIndexPtr();
        v.inner = tmp.innerIndexPtr();
        v.values = tmp.valuePtr();
        v.row_major = Plain::IsRowMajor;
        ingest(v);
    }
#endif

    SparseGenMatProd(mispec_ctx* ctx, mispec_csr* device_matrix) :
        m_ctx(internal::borrow_context(ctx)), m_mat(device_matrix, [](mispec_csr*) {})
    {
        if (!ctx || !device_matrix)
            throw std::invalid_argument("SparseGenMatProd: NULL device handle");
    }

    Index rows() const { return static_cast<Index>(mispec_csr_rows(m_mat.get())); }
    Index cols() const { return static_cast<Index>(mispec_csr_cols(m_mat.get())); }

    void perform_op(const Scalar* x_in, Scalar* y_out) const
    {
        const internal::WidenedIn<Scalar> x(x_in, static_cast<std::size_t>(cols()));
        internal::NarrowedOut<Scalar> y(y_out, static_cast<std::size_t>(rows()));
        internal::check(mispec_spmv_host(m_mat.get(), x.data(), y.data()));
        y.store();
    }

    Matrix operator*(const Matrix& mat_in) const
    {
        Matrix res(rows(), mat_in.cols());
        const internal::WidenedIn<Scalar> in(mat_in.data(), static_cast<std::size_t>(mat_in.rows() * mat_in.cols()));
        internal::NarrowedOut<Scalar> out(res.data(), static_cast<std::size_t>(res.rows() * res.cols()));
        internal::check(mispec_spmm_host(m_mat.get(), in.data(), mat_in.rows(), static_cast<int>(mat_in.cols()), out.data(), res.rows()));
        out.store();
        return res;
    }

    Scalar operator()(Index i, Index j) const
    {
        double v = 0;
        internal::check(mispec_csr_coeff(m_mat.get(), i, j, &v));
        return static_cast<Scalar>(v);
    }

    mispec_ctx* mispec_context() const { return m_ctx.get(); }
    const mispec_csr* mispec_matrix() const { return m_mat.get(); }
};

}  // namespace Spectra

#endif
