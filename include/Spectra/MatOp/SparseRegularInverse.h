// The B operator of a generalized problem A x = lambda B x in regular-inverse mode: y = B x and
// y = B^{-1} x for a sparse positive-definite B.  Same template signature and members as the reference class
// (MatOp/SparseRegularInverse.h:33-128): rows(), cols(), info(), solve(), perform_op().
//
// The reference holds an Eigen::ConjugateGradient<SparseMatrix> (lower triangle, diagonal preconditioner,
// tolerance epsilon, at most 2n iterations, start vector 0).  Here the selected triangle of B is mirrored into a
// CSR matrix in HBM and the same iteration runs on the GPU (spectra_amd/csrc/reginv.hip): one CSR-stream SpMV and
// two fused vector kernels per iteration, reproducible two-stage reductions.  The solvers bind the device
// object; solve()/perform_op() keep the host-pointer contract for direct callers.
#ifndef MISPEC_SPECTRA_SPARSE_REGULAR_INVERSE_H
#define MISPEC_SPECTRA_SPARSE_REGULAR_INVERSE_H

#include <memory>
#include <stdexcept>
#include <type_traits>

#include "../Util/CompInfo.h"
#include "../internal/Dense.h"
#include "../internal/Device.h"

namespace Spectra {

template <typename Scalar_, int Uplo = Lower, int Flags = ColMajor, typename StorageIndex = int>
class SparseRegularInverse
{
public:
    using Scalar = Scalar_;

private:
    static_assert(std::is_same<Scalar_, double>::value, "the MI355X path computes in fp64: Scalar must be double");
    internal::CtxPtr m_ctx;
    std::shared_ptr<mispec_reginv> m_B;
    mutable CompInfo m_info = CompInfo::Successful;

    void ingest(const SparseView<Scalar, StorageIndex>& B)
    {
        if (B.rows != B.cols)
            throw std::invalid_argument("SparseRegularInverse: matrix must be square");
        if (B.row_major != (Flags == RowMajor))
            throw std::invalid_argument(
                "SparseRegularInverse: the \"Flags\" template parameter does not match the input matrix (ColMajor/RowMajor)");
        mispec_reginv* raw = nullptr;
        internal::check(
            mispec_reginv_create(m_ctx.get(), B.rows, internal::Int32Indices<StorageIndex>(B.outer, static_cast<std::size_t>(B.rows) + 1).data(),
                                 internal::Int32Indices<StorageIndex>(B.inner, static_cast<std::size_t>(B.outer[B.rows])).data(), B.values, Uplo == Lower ? 'L' : 'U', B.row_major ? 1 : 0, &raw));
        m_B = std::shared_ptr<mispec_reginv>(raw, [](mispec_reginv* p) { (void) mispec_reginv_destroy(p); });
    }

public:
    explicit SparseRegularInverse(const SparseView<Scalar, StorageIndex>& mat, internal::CtxPtr ctx = internal::CtxPtr()) :
        m_ctx(ctx ? ctx : internal::default_context())
    {
        ingest(mat);
    }

#ifdef MISPEC_HAVE_EIGEN
    template <typename Derived>
    SparseRegularInverse(const Eigen::SparseMatrixBase<Derived>& mat) : m_ctx(internal::default_context())
    {
        using Plain = Eigen::SparseMatrix<Scalar, Flags, StorageIndex>;
        static_assert(static_cast<int>(Derived::PlainObject::IsRowMajor) == static_cast<int>(Plain::IsRowMajor),
                      "SparseRegularInverse: the \"Flags\" template parameter does not match the input matrix");
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

    // adopt an operator created through the C ABI (not owned)
    SparseRegularInverse(mispec_ctx* ctx, mispec_reginv* B) : m_ctx(internal::borrow_context(ctx)), m_B(B, [](mispec_reginv*) {})
    {
        if (!ctx || !B)
            throw std::invalid_argument("SparseRegularInverse: NULL device handle");
    }

    Index rows() const { return static_cast<Index>(mispec_reginv_rows(m_B.get())); }
    Index cols() const { return rows(); }

    // Status of the last solve (reference :80)
    CompInfo info() const { return m_info; }

    // y_out = inv(B) * x_in, host pointers; throws std::runtime_error if the CG iteration does not converge (:113-114)
    void solve(const Scalar* x_in, Scalar* y_out) const
    {
        const int rc = mispec_reginv_solve_host(m_B.get(), x_in, y_out);
        m_info = (rc == MISPEC_OK) ? CompInfo::Successful : CompInfo::NotConverging;
        internal::check(rc);
    }

    // y_out = B * x_in, host pointers
    void perform_op(const Scalar* x_in, Scalar* y_out) const { internal::check(mispec_reginv_perform_op_host(m_B.get(), x_in, y_out)); }

    mispec_ctx* mispec_context() const { return m_ctx.get(); }
    const mispec_reginv* mispec_b_operator() const { return m_B.get(); }
};

}  // namespace Spectra

#endif
