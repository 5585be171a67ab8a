// The Krylov operators of the generalized solver's shift modes (reference, internal use:
// MatOp/internal/SymGEigsShiftInvertOp.h:27-78, SymGEigsBucklingOp.h:27-60, SymGEigsCayleyOp.h:27-102).
// All three apply inv(A - sigma B) to B x — a SymShiftInvert OpType and a matrix-product BOpType — and the Cayley
// one adds x + 2 sigma * that.  perform_op() keeps the host-pointer contract; with the device operators of this
// library the factorisation binds the pencil solver and B directly and every step stays on the GPU.
#ifndef MISPEC_SPECTRA_SYM_GEIGS_SHIFT_OPS_H
#define MISPEC_SPECTRA_SYM_GEIGS_SHIFT_OPS_H

#include <vector>

#include "../SparseSymMatProd.h"
#include "../SymShiftInvert.h"

namespace Spectra {

namespace internal {
template <typename OpType, typename BOpType, bool Cayley>
class SymGEigsShiftOpBase
{
public:
    using Scalar = typename OpType::Scalar;

protected:
    OpType& m_op;
    const BOpType& m_Bop;
    mutable std::vector<Scalar> m_cache;
    Scalar m_sigma = Scalar(0);

public:
    SymGEigsShiftOpBase(OpType& op, const BOpType& Bop) : m_op(op), m_Bop(Bop), m_cache(static_cast<std::size_t>(op.rows())) {}
    SymGEigsShiftOpBase(SymGEigsShiftOpBase&& other) : m_op(other.m_op), m_Bop(other.m_Bop), m_sigma(other.m_sigma)
    {
        m_cache.swap(other.m_cache);
    }

    Index rows() const { return m_op.rows(); }
    Index cols() const { return m_op.rows(); }

    void set_shift(const Scalar& sigma)
    {
        m_op.set_shift(sigma);
        m_sigma = sigma;
    }

    // y_out = inv(A - sigma B) * B * x_in      (+ Cayley: y_out = x_in + 2 sigma * y_out)
    void perform_op(const Scalar* x_in, Scalar* y_out) const
    {
        m_Bop.perform_op(x_in, m_cache.data());
        m_op.perform_op(m_cache.data(), y_out);
        if (Cayley)
            for (Index i = 0; i < rows(); i++)
                y_out[i] = x_in[i] + (Scalar(2) * m_sigma) * y_out[i];
    }

    // device hooks
    mispec_ctx* mispec_context() const { return m_op.mispec_context(); }
    const mispec_symshift* mispec_geigs_shift_solver() const { return m_op.mispec_solver(); }
    const mispec_csr* mispec_geigs_shift_b() const { return m_Bop.mispec_matrix(); }
    bool mispec_geigs_shift_cayley() const { return Cayley; }
    double mispec_geigs_shift_sigma() const { return m_sigma; }
};
}  // namespace internal

template <typename OpType, typename BOpType>
class SymGEigsShiftInvertOp : public internal::SymGEigsShiftOpBase<OpType, BOpType, false>
{
public:
    using internal::SymGEigsShiftOpBase<OpType, BOpType, false>::SymGEigsShiftOpBase;
};
template <typename OpType, typename BOpType>
class SymGEigsBucklingOp : public internal::SymGEigsShiftOpBase<OpType, BOpType, false>
{
public:
    using internal::SymGEigsShiftOpBase<OpType, BOpType, false>::SymGEigsShiftOpBase;
};
template <typename OpType, typename BOpType>
class SymGEigsCayleyOp : public internal::SymGEigsShiftOpBase<OpType, BOpType, true>
{
public:
    using internal::SymGEigsShiftOpBase<OpType, BOpType, true>::SymGEigsShiftOpBase;
};

}  // namespace Spectra

#endif
