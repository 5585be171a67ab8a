// The B operator of the generalized solver's Cholesky mode: B = L L' for a sparse positive-definite B, with
// y = L^{-1} x and y = L^{-T} x (reference: MatOp/SparseCholesky.h:36-128 — same members: rows(), cols(), info(),
// lower_triangular_solve(), upper_triangular_solve()).
// The reference factors with Eigen::SimplicialLLT (sparse, fill-reducing permutation).  Here, for n <= 4096 the factor is
// dense and kept, inverted, in HBM (spectra_amd/csrc/cholesky.hip); for larger n B must be banded (half-bandwidth <= 64 since
// round 6, <= 8 before: mass / stiffness matrices of structured problems) and the factor is the one of the partitioned band factorisation
// (shiftsolve.hip) applied in two halves on the device.  Any G with G G' = B yields the same eigenpairs; other large
// sparse B are served by the regular-inverse mode (the constructor throws std::invalid_argument for them).
#ifndef MISPEC_SPECTRA_SPARSE_CHOLESKY_H
#define MISPEC_SPECTRA_SPARSE_CHOLESKY_H

#include <memory>
#include <stdexcept>
#include <type_traits>

#include "../Util/CompInfo.h"
#include "../internal/Dense.h"
#include "../internal/Device.h"

namespace Spectra {

template <typename Scalar_, int Uplo = Lower, int Flags = ColMajor, typename StorageIndex = int>
class SparseCholesky
{
public:
    using Scalar = Scalar_;

private:
    static_assert(std::is_same<Scalar_, double>::value, "the MI355X path computes in fp64: Scalar must be double");
    internal::CtxPtr m_ctx;
    std::shared_ptr<mispec_cholesky> m_chol;

    void ingest(const SparseView<Scalar, StorageIndex>& B)
    {
        if (B.rows != B.cols)
            throw std::invalid_argument("SparseCholesky: matrix must be square");
        if (B.row_major != (Flags == RowMajor))
            throw std::invalid_argument(
                "SparseCholesky: the \"Flags\" template parameter does not match the input matrix (ColMajor/RowMajor)");
        mispec_cholesky* raw = nullptr;
        const std::size_t nnz = static_cast<std::size_t>(B.outer[B.rows]);
        const internal::Int32Indices<StorageIndex> outer(B.outer, static_cast<std::size_t>(B.rows) + 1), inner(B.inner, nnz);
        internal::check(mispec_cholesky_create(m_ctx.get(), B.rows, outer.data(), inner.data(), B.values, Uplo == Lower ? 'L' : 'U',
                                               B.row_major ? 1 : 0, &raw));
        m_chol = std::shared_ptr<mispec_cholesky>(raw, [](mispec_cholesky* p) { (void) mispec_cholesky_destroy(p); });
    }

public:
    explicit SparseCholesky(const SparseView<Scalar, StorageIndex>& mat, internal::CtxPtr ctx = internal::CtxPtr()) :
        m_ctx(ctx ? ctx : internal::default_context())
    {
        ingest(mat);
    }

#ifdef MISPEC_HAVE_EIGEN
    template <typename Derived>
    SparseCholesky(const Eigen::SparseMatrixBase<Derived>& mat) : m_ctx(internal::default_context())
    {
        using Plain = Eigen::SparseMatrix<Scalar, Flags, StorageIndex>;
        static_assert(static_cast<int>(Derived::PlainObject::IsRowMajor) == static_cast<int>(Plain::IsRowMajor),
                      "SparseCholesky: the \"Flags\" template parameter does not match the input matrix");
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

    // adopt a factor created through the C ABI (not owned)
    SparseCholesky(mispec_ctx* ctx, mispec_cholesky* chol) : m_ctx(internal::borrow_context(ctx)), m_chol(chol, [](mispec_cholesky*) {})
    {
        if (!ctx || !chol)
            throw std::invalid_argument("SparseCholesky: NULL device handle");
    }

    Index rows() const { return static_cast<Index>(mispec_cholesky_rows(m_chol.get())); }
    Index cols() const { return rows(); }

    // Successful, or NumericalIssue if B is not positive definite (reference :93-97)
    CompInfo info() const { return mispec_cholesky_info(m_chol.get()) == 0 ? CompInfo::Successful : CompInfo::NumericalIssue; }

    // y_out = inv(L) * x_in, host pointers
    void lower_triangular_solve(const Scalar* x_in, Scalar* y_out) const
    {
        internal::check(mispec_cholesky_lower_solve_host(m_chol.get(), x_in, y_out));
    }
    // y_out = inv(L') * x_in, host pointers
    void upper_triangular_solve(const Scalar* x_in, Scalar* y_out) const
    {
        internal::check(mispec_cholesky_upper_solve_host(m_chol.get(), x_in, y_out));
    }

    mispec_ctx* mispec_context() const { return m_ctx.get(); }
    const mispec_cholesky* mispec_factor() const { return m_chol.get(); }
};

}  // namespace Spectra

#endif
