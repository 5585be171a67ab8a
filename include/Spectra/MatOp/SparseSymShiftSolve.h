// y = (A - sigma I)^{-1} x for a real symmetric sparse A — the operator SymEigsShiftSolver iterates on.
// Same template signature and members as the reference class (MatOp/SparseSymShiftSolve.h:36-111):
// Scalar, rows(), cols(), set_shift(sigma), perform_op(x_in, y_out).
//
// The reference factors A - sigma I with Eigen::SparseLU (general sparse, CPU).  Here set_shift() builds a
// factorisation whose SOLVE runs on the GPU (spectra_amd/csrc/shiftsolve.hip): a recursive partitioned banded
// LDL' when the half-bandwidth of A — as given, or after the reverse Cuthill-McKee ordering the constructor tries for wider patterns — is <= 8 (any n) or <= 64 with n > 4096 (csrc/shiftsolve.hpp band_path), or a dense inverse when n <= 4096; other patterns throw
// std::invalid_argument at set_shift().  The solver binds the device operator directly; perform_op() keeps the
// reference's host-pointer contract for everybody else.
#ifndef MISPEC_SPECTRA_SPARSE_SYM_SHIFT_SOLVE_H
#define MISPEC_SPECTRA_SPARSE_SYM_SHIFT_SOLVE_H

#include <type_traits>

#include "../internal/Dense.h"
#include "../internal/Device.h"

namespace Spectra {

template <typename Scalar_, int Uplo = Lower, int Flags = ColMajor, typename StorageIndex = int>
class SparseSymShiftSolve
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
            throw std::invalid_argument("SparseSymShiftSolve: matrix must be square");
        if (A.row_major != (Flags == RowMajor))
            throw std::invalid_argument(
                "SparseSymShiftSolve: the \"Flags\" template parameter does not match the input matrix (ColMajor/RowMajor)");
        mispec_symshift* raw = nullptr;
        const std::size_t nnz = static_cast<std::size_t>(A.outer[A.rows]);
        const internal::Int32Indices<StorageIndex> outer(A.outer, static_cast<std::size_t>(A.rows) + 1), inner(A.inner, nnz);
        internal::check(mispec_symshift_create(m_ctx.get(), A.rows, outer.data(), inner.data(), A.values, Uplo == Lower ? 'L' : 'U',
                                               A.row_major ? 1 : 0, &raw));
        m_solver = std::shared_ptr<mispec_symshift>(raw, [](mispec_symshift* p) { (void) mispec_symshift_destroy(p); });
    }

public:
    explicit SparseSymShiftSolve(const SparseView<Scalar, StorageIndex>& mat, internal::CtxPtr ctx = internal::CtxPtr()) :
        m_ctx(ctx ? ctx : internal::default_context())
    {
        ingest(mat);
    }

#ifdef MISPEC_HAVE_EIGEN
    template <typename Derived>
    SparseSymShiftSolve(const Eigen::SparseMatrixBase<Derived>& mat) : m_ctx(internal::default_context())
    {
        using Plain = Eigen::SparseMatrix<Scalar, Flags, StorageIndex>;
        static_assert(static_cast<int>(Derived::PlainObject::IsRowMajor) == static_cast<int>(Plain::IsRowMajor),
                      "SparseSymShiftSolve: the \"Flags\" template parameter does not match the input matrix");
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
    SparseSymShiftSolve(mispec_ctx* ctx, mispec_symshift* solver) :
        m_ctx(internal::borrow_context(ctx)), m_solver(solver, [](mispec_symshift*) {})
    {
        if (!ctx || !solver)
            throw std::invalid_argument("SparseSymShiftSolve: NULL device handle");
    }

    Index rows() const { return static_cast<Index>(mispec_symshift_rows(m_solver.get())); }
    Index cols() const { return rows(); }

    // Factor A - sigma I; throws std::invalid_argument if that fails (reference :93-94).
    void set_shift(const Scalar& sigma) { internal::check(mispec_symshift_set_shift(m_solver.get(), sigma)); }

    // y_out = inv(A - sigma I) * x_in, host pointers
    void perform_op(const Scalar* x_in, Scalar* y_out) const
    {
        internal::check(mispec_symshift_solve_host(m_solver.get(), x_in, y_out));
    }

    mispec_ctx* mispec_context() const { return m_ctx.get(); }
    const mispec_symshift* mispec_solver() const { return m_solver.get(); }
};

}  // namespace Spectra

#endif
