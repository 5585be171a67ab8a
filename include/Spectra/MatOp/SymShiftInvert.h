// y = (A - sigma B)^{-1} x for two sparse symmetric matrices — the operator the shift modes of the generalized
// solver are built on (reference: MatOp/SymShiftInvert.h:120-208, which factors A - sigma B with Eigen::SparseLU or
// a dense Bunch-Kaufman LDLT depending on the storage of A and B).  Here both matrices are sparse and the
// factorisation is the device one of SparseSymShiftSolve (spectra_amd/csrc/shiftsolve.hip): banded A - sigma B with
// half-bandwidth <= 8 (<= 64 beyond n = 4096), or any pattern with n <= 4096.
#ifndef MISPEC_SPECTRA_SYM_SHIFT_INVERT_H
#define MISPEC_SPECTRA_SYM_SHIFT_INVERT_H

#include <memory>
#include <stdexcept>
#include <type_traits>

#include "../internal/Dense.h"
#include "../internal/Device.h"

namespace Spectra {

#ifdef MISPEC_HAVE_EIGEN
using Eigen::Dense;
using Eigen::Sparse;
#else
// storage tags of the reference's template signature (Eigen::Sparse / Eigen::Dense)
struct Sparse
{};
struct Dense
{};
#endif

template <typename Scalar_, typename TypeA = Sparse, typename TypeB = Sparse, int UploA = Lower, int UploB = Lower,
          int FlagsA = ColMajor, int FlagsB = ColMajor, typename StorageIndexA = int, typename StorageIndexB = int>
class SymShiftInvert
{
public:
    using Scalar = Scalar_;

private:
    static_assert(std::is_same<Scalar_, double>::value, "the MI355X path computes in fp64: Scalar must be double");
    static_assert(std::is_same<TypeA, Sparse>::value && std::is_same<TypeB, Sparse>::value,
                  "SymShiftInvert: the device path takes sparse A and B");
    static_assert(std::is_same<StorageIndexA, int>::value && std::is_same<StorageIndexB, int>::value,
                  "sparse indices are int32 on the device");
    internal::CtxPtr m_ctx;
    std::shared_ptr<mispec_symshift> m_solver;

    void ingest(const SparseView<Scalar, int>& A, const SparseView<Scalar, int>& B)
    {
        if (A.rows != A.cols || B.rows != A.rows || B.cols != A.rows)
            throw std::invalid_argument("SymShiftInvert: A and B must be square matrices of the same size");
        if (A.row_major != (FlagsA == RowMajor) || B.row_major != (FlagsB == RowMajor))
            throw std::invalid_argument(
                "SymShiftInvert: the \"FlagsA\" / \"FlagsB\" template parameter does not match the input matrix (ColMajor/RowMajor)");
        mispec_symshift* raw = nullptr;
        internal::check(mispec_symshift_create_pencil(m_ctx.get(), A.rows, A.outer, A.inner, A.values, UploA == Lower ? 'L' : 'U',
                                                      A.row_major ? 1 : 0, B.outer, B.inner, B.values, UploB == Lower ? 'L' : 'U',
                                                      B.row_major ? 1 : 0, &raw));
        m_solver = std::shared_ptr<mispec_symshift>(raw, [](mispec_symshift* p) { (void) mispec_symshift_destroy(p); });
    }

public:
    SymShiftInvert(const SparseView<Scalar, int>& A, const SparseView<Scalar, int>& B, internal::CtxPtr ctx = internal::CtxPtr()) :
        m_ctx(ctx ? ctx : internal::default_context())
    {
        ingest(A, B);
    }

#ifdef MISPEC_HAVE_EIGEN
    template <typename DerivedA, typename DerivedB>
    SymShiftInvert(const Eigen::SparseMatrixBase<DerivedA>& A, const Eigen::SparseMatrixBase<DerivedB>& B) :
        m_ctx(internal::default_context())
    {
        Eigen::SparseMatrix<Scalar, FlagsA, int> a(A);
        Eigen::SparseMatrix<Scalar, FlagsB, int> b(B);
        a.makeCompressed();
        b.makeCompressed();
        SparseView<Scalar, int> va, vb;
        va.rows = a.rows(), va.cols = a.cols(), va.outer = a.outerIndexPtr(), va.inner = a.innerIndexPtr(), va.values = a.valuePtr();
        va.row_major = (FlagsA == RowMajor);
        vb.rows = b.rows(), vb.cols = b.cols(), vb.outer = b.outerIndexPtr(), vb.inner = b.innerIndexPtr(), vb.values = b.valuePtr();
        vb.row_major = (FlagsB == RowMajor);
        ingest(va, vb);
    }
#endif

    // adopt a pencil solver created through the C ABI (not owned)
    SymShiftInvert(mispec_ctx* ctx, mispec_symshift* solver) : m_ctx(internal::borrow_context(ctx)), m_solver(solver, [](mispec_symshift*) {})
    {
        if (!ctx || !solver)
            throw std::invalid_argument("SymShiftInvert: NULL device handle");
    }

    Index rows() const { return static_cast<Index>(mispec_symshift_rows(m_solver.get())); }
    Index cols() const { return rows(); }

    // Factor A - sigma B; throws std::invalid_argument if that fails (reference :189-190)
    void set_shift(const Scalar& sigma) { internal::check(mispec_symshift_set_shift(m_solver.get(), sigma)); }

    // y_out = inv(A - sigma * B) * x_in, host pointers
    void perform_op(const Scalar* x_in, Scalar* y_out) const { internal::check(mispec_symshift_solve_host(m_solver.get(), x_in, y_out)); }

    mispec_ctx* mispec_context() const { return m_ctx.get(); }
    const mispec_symshift* mispec_solver() const { return m_solver.get(); }
};

}  // namespace Spectra

#endif
