// y = A x for a general real sparse A — the operator GenEigsSolver is normally used with.  Same template
// signature and members as the reference class (MatOp/SparseGenMatProd.h:28-105).  As for
// SparseSymMatProd the matrix is copied to HBM at construction (CSC input is transposed to CSR once)
// and the solvers bind the device matrix directly instead of calling perform_op().
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
    static_assert(std::is_same<Scalar_, double>::value, "the MI355X path computes in fp64: Scalar must be double");
    static_assert(std::is_same<StorageIndex, int>::value, "sparse indices are int32 on the device");
    using Matrix = DenseMatrix<Scalar>;

    internal::CtxPtr m_ctx;
    std::shared_ptr<mispec_csr> m_mat;

    void ingest(const SparseView<Scalar, StorageIndex>& A)
    {
        if (A.row_major != (Flags == RowMajor))
            throw std::invalid_argument(
                "SparseGenMatProd: the \"Flags\" template parameter does not match the input matrix (ColMajor/RowMajor)");
        mispec_csr* raw = nullptr;
        if (A.row_major)
            internal::check(mispec_csr_upload(m_ctx.get(), A.rows, A.cols, A.outer, A.inner, A.values, &raw));
        else
            internal::check(mispec_csr_from_csc(m_ctx.get(), A.rows, A.cols, A.outer, A.inner, A.values, &raw));
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

    void perform_op(const Scalar* x_in, Scalar* y_out) const { internal::check(mispec_spmv_host(m_mat.get(), x_in, y_out)); }

    Matrix operator*(const Matrix& mat_in) const
    {
        Matrix res(rows(), mat_in.cols());
        internal::check(mispec_spmm_host(m_mat.get(), mat_in.data(), mat_in.rows(), static_cast<int>(mat_in.cols()), res.data(),
                                         res.rows()));
        return res;
    }

    Scalar operator()(Index i, Index j) const
    {
        Scalar v = 0;
        internal::check(mispec_csr_coeff(m_mat.get(), i, j, &v));
        return v;
    }

    mispec_ctx* mispec_context() const { return m_ctx.get(); }
    const mispec_csr* mispec_matrix() const { return m_mat.get(); }
};

}  // namespace Spectra

#endif
