// y = A x for a real symmetric sparse A of which ONE triangle is given — the operator SymEigsSolver is
// normally used with.  Same template signature and members as the reference class
// (MatOp/SparseSymMatProd.h:30-105): Scalar, rows(), cols(), perform_op(), operator*, operator().
//
// Differences that a user can observe:
//   * the matrix is copied to the GPU at construction (the reference keeps an Eigen::Ref): the
//     `Uplo` triangle is mirrored into a full CSR shard in HBM, entries of the other triangle are
//     ignored exactly as selfadjointView<Uplo> ignores them (test/SymEigs.cpp:27-28 relies on it);
//   * perform_op(x_in, y_out) still takes HOST pointers and is then a staged H2D / kernel / D2H
//     round trip; the solvers do not go through it for this class — they bind the device matrix
//     directly (see HermEigsBase.h) and keep the whole Krylov basis in HBM.
// Scalar = float is accepted at this boundary like the reference's class (test/SparseSymMatProd.cpp:37): values are widened to the
// device's fp64 on the way in and results rounded once on the way out; the solvers themselves require double.
#ifndef MISPEC_SPECTRA_SPARSE_SYM_MAT_PROD_H
#define MISPEC_SPECTRA_SPARSE_SYM_MAT_PROD_H

#include <type_traits>
#include <vector>

#include "../internal/Dense.h"
#include "../internal/Device.h"

namespace Spectra {

template <typename Scalar_, int Uplo = Lower, int Flags = ColMajor, typename StorageIndex = int>
class SparseSymMatProd
{
public:
    using Scalar = Scalar_;

private:
    static_assert(internal::is_device_scalar<Scalar_>::value, "Scalar must be double (or float, widened: the MI355X path computes in fp64)");
    static_assert(Uplo == Lower || Uplo == Upper, "Uplo must be Lower or Upper");
    using Matrix = DenseMatrix<Scalar>;

    internal::CtxPtr m_ctx;
    std::shared_ptr<mispec_csr> m_mat;

    void ingest(const SparseView<Scalar, StorageIndex>& A)
    {
        if (A.rows != A.cols)
            throw std::invalid_argument("SparseSymMatProd: matrix must be square");
        if (A.row_major != (Flags == RowMajor))
            throw std::invalid_argument(
                "SparseSymMatProd: the \"Flags\" template parameter does not match the input matrix (ColMajor/RowMajor)");
        mispec_csr* raw = nullptr;
        const std::size_t nnz = static_cast<std::size_t>(A.outer[A.rows]);
        const internal::WidenedIn<Scalar> values(A.values, nnz);
        // (any StorageIndex, as the reference's SparseSymMatProd.h:30: narrowed to the device's int32 where it is not int)
        const internal::Int32Indices<StorageIndex> outer(A.outer, static_cast<std::size_t>(A.rows) + 1), inner(A.inner, nnz);
        internal::check(mispec_csr_from_triangle(m_ctx.get(), A.rows, outer.data(), inner.data(), values.data(), Uplo == Lower ? 'L' : 'U',
                                                 A.row_major ? 1 : 0, &raw));
        m_mat = std::shared_ptr<mispec_csr>(raw, [](mispec_csr* p) { (void) mispec_csr_destroy(p); });
    }

public:
    // From a compressed sparse matrix in host memory.
    explicit SparseSymMatProd(const SparseView<Scalar, StorageIndex>& mat, internal::CtxPtr ctx = internal::CtxPtr()) :
        m_ctx(ctx ? ctx : internal::default_context())
    {
        ingest(mat);
    }

#ifdef MISPEC_HAVE_EIGEN
    // The reference's constructor (SparseSymMatProd.h:58-65): any Eigen sparse expression of matching storage order.
    template <typename Derived>
    SparseSymMatProd(const Eigen::SparseMatrixBase<Derived>& mat) : m_ctx(internal::default_context())
    {
        using Plain = Eigen::SparseMatrix<Scalar, Flags, StorageIndex>;
        static_assert(static_cast<int>(Derived::PlainObject::IsRowMajor) == static_cast<int>(Plain::IsRowMajor),
                      "SparseSymMatProd: the \"Flags\" template parameter does not match the input matrix");
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

    // Adopt a matrix that already lives on the device (generated there, or uploaded through the C ABI).
    // The handle must already be symmetric; it is not owned.
    SparseSymMatProd(mispec_ctx* ctx, mispec_csr* device_matrix) :
        m_ctx(internal::borrow_context(ctx)), m_mat(device_matrix, [](mispec_csr*) {})
    {
        if (!ctx || !device_matrix)
            throw std::invalid_argument("SparseSymMatProd: NULL device handle");
    }

    Index rows() const { return static_cast<Index>(mispec_csr_rows(m_mat.get())); }
    Index cols() const { return static_cast<Index>(mispec_csr_cols(m_mat.get())); }

    // y_out = A * x_in, host pointers (the reference's contract, SparseSymMatProd.h:83-88)
    void perform_op(const Scalar* x_in, Scalar* y_out) const
    {
        const internal::WidenedIn<Scalar> x(x_in, static_cast<std::size_t>(cols()));
        internal::NarrowedOut<Scalar> y(y_out, static_cast<std::size_t>(rows()));
        internal::check(mispec_spmv_host(m_mat.get(), x.data(), y.data()));
        y.store();
    }

    // Y = A * X for a dense block (SparseSymMatProd.h:93-96)
    Matrix operator*(const Matrix& mat_in) const
    {
        Matrix res(rows(), mat_in.cols());
        const internal::WidenedIn<Scalar> in(mat_in.data(), static_cast<std::size_t>(mat_in.rows() * mat_in.cols()));
        internal::NarrowedOut<Scalar> out(res.data(), static_cast<std::size_t>(res.rows() * res.cols()));
        internal::check(mispec_spmm_host(m_mat.get(), in.data(), mat_in.rows(), static_cast<int>(mat_in.cols()), out.data(), res.rows()));
        out.store();
        return res;
    }

    // A(i, j) of the symmetric operator (SparseSymMatProd.h:101-104 returns the stored coefficient of the
    // input; here both triangles answer because the mirror is what is stored).
    Scalar operator()(Index i, Index j) const
    {
        double v = 0;
        internal::check(mispec_csr_coeff(m_mat.get(), i, j, &v));
        return static_cast<Scalar>(v);
    }

    // Device binding used by the solvers' fast path.
    mispec_ctx* mispec_context() const { return m_ctx.get(); }
    const mispec_csr* mispec_matrix() const { return m_mat.get(); }
};

}  // namespace Spectra

#endif
