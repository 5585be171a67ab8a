// The wrapper the reference puts between a matrix operator and its factorisation classes (MatOp/internal/ArnoldiOp.h:32-162):
// `Arnoldi<ArnoldiOp<OpType, IdentityBOp>>`, `Lanczos<ArnoldiOp<OpType, IdentityBOp>>` (test/Arnoldi.cpp:94-158 spells them
// that way).  Here the inner products, V^H y and the norms of the B = I case are kernels of the device factorisation, so the
// wrapper only names the operator; LinAlg/Arnoldi.h specialises the factorisation classes for it.  Non-identity B operators
// reach the device through the generalized solvers (SymGEigsSolver.h), not through this class.
#ifndef MISPEC_SPECTRA_ARNOLDI_OP_H
#define MISPEC_SPECTRA_ARNOLDI_OP_H

#include "../../internal/Dense.h"

namespace Spectra {

// Placeholder for the B operator when B = I (ArnoldiOp.h:100-107)
class IdentityBOp
{};

template <typename OpType, typename BOpType>
class ArnoldiOp;

template <typename OpType>
class ArnoldiOp<OpType, IdentityBOp>
{
public:
    using Scalar = typename OpType::Scalar;

private:
    const OpType& m_op;

public:
    ArnoldiOp(const OpType& op, const IdentityBOp& /*Bop*/) : m_op(op) {}

    Index rows() const { return m_op.rows(); }
    // The "A" operator that generates the Krylov subspace (ArnoldiOp.h:158-161)
    void perform_op(const Scalar* x_in, Scalar* y_out) const { m_op.perform_op(x_in, y_out); }
    // the wrapped operator: what the device factorisation binds
    const OpType& op() const { return m_op; }
};

}  // namespace Spectra

#endif
