// The Krylov operator of the generalized eigen solver in regular-inverse mode: y = B^{-1} A x
// (reference: MatOp/internal/SymGEigsRegInvOp.h:27-82; internal use).  perform_op() keeps the host-pointer
// contract; when A is a device matrix and B a device SparseRegularInverse the factorisation binds both and
// the whole step — SpMV, conjugate gradient, B-inner products — stays on the GPU.
#ifndef MISPEC_SPECTRA_SYM_GEIGS_REG_INV_OP_H
#define MISPEC_SPECTRA_SYM_GEIGS_REG_INV_OP_H

#include <vector>

#include "../SparseRegularInverse.h"
#include "../SparseSymMatProd.h"

namespace Spectra {

template <typename OpType = SparseSymMatProd<double>, typename BOpType = SparseRegularInverse<double>>
class SymGEigsRegInvOp
{
public:
    using Scalar = typename OpType::Scalar;

private:
    const OpType& m_op;
    const BOpType& m_Bop;
    mutable std::vector<Scalar> m_cache;  // temporary working space

public:
    SymGEigsRegInvOp(const OpType& op, const BOpType& Bop) : m_op(op), m_Bop(Bop), m_cache(static_cast<std::size_t>(op.rows())) {}
    SymGEigsRegInvOp(SymGEigsRegInvOp&& other) : m_op(other.m_op), m_Bop(other.m_Bop) { m_cache.swap(other.m_cache); }

    Index rows() const { return m_Bop.rows(); }
    Index cols() const { return m_Bop.rows(); }

    // y_out = inv(B) * A * x_in
    void perform_op(const Scalar* x_in, Scalar* y_out) const
    {
        m_op.perform_op(x_in, m_cache.data());
        m_Bop.solve(m_cache.data(), y_out);
    }

    // device hooks: the factorisation binds A and B instead of calling perform_op
    mispec_ctx* mispec_context() const { return m_op.mispec_context(); }
    const mispec_csr* mispec_geigs_matrix() const { return m_op.mispec_matrix(); }
    const mispec_reginv* mispec_geigs_b_operator() const { return m_Bop.mispec_b_operator(); }
};

}  // namespace Spectra

#endif
