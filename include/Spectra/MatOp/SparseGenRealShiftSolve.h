// y = (A - sigma I)^{-1} x for a general real sparse A — the operator of GenEigsRealShiftSolver (reference:
// MatOp/SparseGenRealShiftSolve.h:25-100, which factors with Eigen::SparseLU).  Same members: rows(), cols(),
// set_shift(), perform_op().  The device factorisation is the dense one (LU with partial pivoting, explicit
// inverse in HBM, GEMV per application): n <= 4096.
#ifndef MISPEC_SPECTRA_SPARSE_GEN_REAL_SHIFT_SOLVE_H
#define MISPEC_SPECTRA_SPARSE_GEN_REAL_SHIFT_SOLVE_H

#include <memory>
#include <stdexcept>
#include <type_traits>

#include "../internal/Dense.h"
#include "../internal/Device.h"

namespace Spectra {

template <typename Scalar_, int Flags = ColMajor, typename StorageIndex = int>
class SparseGenRealShiftSolve
{
public:
    using Scalar = Scalar_;

private:
    static_assert(std::is_same<Scalar_, double>::value, "the MI355X path computes in fp64: Scalar must be double");
    internal::CtxPtr m_ctx;
    std::shared_ptr<mispec_symshift> m_solver;

    void ingest(const SparseView<Scalar, StorageIndex>& A)
    {
        if (A.rows != A.cols)
            throw std::invalid_argument("SparseGenRealShiftSolve: matrix must be square");
        if (A.row_major != (Flags == RowMajor))
            throw std::invalid_argument(
                "SparseGenRealShiftSolve: the \"Flags\" template parameter does not match the input matrix (ColMajor/RowMajor)");
        mispec_symshift* raw = nullptr;
        const std::size_t nnz = static_cast<std::size_t>(A.outer[A.rows]);
        const internal::Int32Indices<StorageIndex> outer(A.outer, static_cast<std::size_t>(A.rows) + 1), inner(A.inner, nnz);
        internal::check(mispec_symshift_create_general(m_ctx.get(), A.rows, outer.data(), inner.data(), A.values, A.row_major ? 1 : 0, &raw));
        m_solver = std::shared_ptr<mispec_symshift>(raw, [](mispec_symshift* p) { (void) mispec_symshift_destroy(p); });
    }

public:
    explicit SparseGenRealShiftSolve(const SparseView<Scalar, StorageIndex>& mat, internal::CtxPtr ctx = internal::CtxPtr()) :
        m_ctx(ctx ? ctx : internal::default_context())
    {
        ingest(mat);
    }

#ifdef MISPEC_HAVE_EIGEN
    template <typename Derived>
    SparseGenRealShiftSolve(const Eigen::SparseMatrixBase<Derived>& mat) : m_ctx(internal::default_context())
    {
        using Plain = Eigen::SparseMatrix<Scalar, Flags, StorageIndex>;
        static_assert(static_cast<int>(Derived::PlainObject::IsRowMajor) == static_cast<int>(Plain::IsRowMajor),
                      "SparseGenRealShiftSolve: the \"Flags\" template parameter does not match the input matrix");
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

    // adopt a solver created through the C ABI (not owned)
    SparseGenRealShiftSolve(mispec_ctx* ctx, mispec_symshift* solver) :
        m_ctx(internal::borrow_context(ctx)), m_solver(solver, [](mispec_symshift*) {})
    {
        if (!ctx || !solver)
            throw std::invalid_argument("SparseGenRealShiftSolve: NULL device handle");
    }

    Index rows() const { return static_cast<Index>(mispec_symshift_rows(m_solver.get())); }
    Index cols() const { return rows(); }

    // Factor A - sigma I; throws std::invalid_argument if that fails (reference :84-85)
    void set_shift(const Scalar& sigma) { internal::check(mispec_symshift_set_shift(m_solver.get(), sigma)); }

    // y_out = inv(A - sigma * I) * x_in, host pointers
    void perform_op(const Scalar* x_in, Scalar* y_out) const { internal::check(mispec_symshift_solve_host(m_solver.get(), x_in, y_out)); }

    mispec_ctx* mispec_context() const { return m_ctx.get(); }
    const mispec_symshift* mispec_solver() const { return m_solver.get(); }
};

}  // namespace Spectra

#endif
