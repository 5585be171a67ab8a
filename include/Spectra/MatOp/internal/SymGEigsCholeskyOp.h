// The operator of the generalized eigen solver in Cholesky mode: y = L^{-1} A L^{-T} x with B = L L'
// (reference: MatOp/internal/SymGEigsCholeskyOp.h:27-73; internal use).  perform_op() keeps the host-pointer
// contract; with a device A and a device SparseCholesky the factorisation binds both.
#ifndef MISPEC_SPECTRA_SYM_GEIGS_CHOLESKY_OP_H
#define MISPEC_SPECTRA_SYM_GEIGS_CHOLESKY_OP_H

#include <utility>
#include <vector>

#include "../SparseCholesky.h"
#include "../SparseSymMatProd.h"

namespace Spectra {

template <typename OpType = SparseSymMatProd<double>, typename BOpType = SparseCholesky<double>>
class SymGEigsCholeskyOp
{
public:
    using Scalar = typename OpType::Scalar;

private:
    const OpType& m_op;
    const BOpType& m_Bop;
    mutable std::vector<Scalar> m_cache;

public:
    SymGEigsCholeskyOp(const OpType& op, const BOpType& Bop) : m_op(op), m_Bop(Bop), m_cache(static_cast<std::size_t>(op.rows())) {}
    SymGEigsCholeskyOp(SymGEigsCholeskyOp&& other) : m_op(other.m_op), m_Bop(other.m_Bop) { m_cache.swap(other.m_cache); }

    Index rows() const { return m_Bop.rows(); }
    Index cols() const { return m_Bop.rows(); }

    // y_out = inv(L) * A * inv(L') * x_in
    void perform_op(const Scalar* x_in, Scalar* y_out) const
    {
        m_Bop.upper_triangular_solve(x_in, y_out);
        m_op.perform_op(y_out, m_cache.data());
        m_Bop.lower_triangular_solve(m_cache.data(), y_out);
    }

    // device hooks.  The matrix hook exists only when A is a device CSR operator (SparseSymMatProd): the factorisation
    // then runs y = L^{-1} A L^{-T} x entirely in HBM; any other A (e.g. DenseSymMatProd) goes through perform_op above.
    mispec_ctx* mispec_context() const { return m_op.mispec_context(); }
    template <typename T = OpType>
    auto mispec_geigs_cholesky_matrix() const -> decltype(std::declval<const T&>().mispec_matrix())
    {
        return m_op.mispec_matrix();
    }
    const mispec_cholesky* mispec_geigs_cholesky_factor() const { return m_Bop.mispec_factor(); }
};

}  // namespace Spectra

#endif
