// Symmetric specialisation of the device factorisation: three-term recurrence, always-on V'f check with
// up to five corrections, tridiagonal H (reference: LinAlg/Lanczos.h:28-217).  Adds the two restart
// primitives that operate on the tridiagonal H entirely on the GPU.
#ifndef MISPEC_SPECTRA_LANCZOS_H
#define MISPEC_SPECTRA_LANCZOS_H

#include "Arnoldi.h"

namespace Spectra {

template <typename OpType>
class Lanczos : public Arnoldi<OpType>
{
    using Base = Arnoldi<OpType>;
    using typename Base::Matrix;
    using typename Base::Vector;
    using Base::m_fac;
    using Base::m_m;

public:
    using Scalar = typename Base::Scalar;

    Lanczos(const OpType& op, Index m) : Base(op, m, true) {}

    // Eigen-decomposition of the projected tridiagonal H (TridiagEigen, LinAlg/TridiagEigen.h:121-210),
    // computed by a single-workgroup LDS kernel.  evals has m entries, evecs is m x m.
    void ritz_pairs(Vector& evals, Matrix& evecs) const
    {
        evals.resize(m_m);
        evecs.resize(m_m, m_m);
        internal::check(mispec_fac_tridiag_eigen(m_fac.get(), evals.data(), evecs.data()));
    }

    // The Ritz values and the last row of their eigenvector matrix only (the convergence test of an iteration needs no more:
    // HermEigsBase.h num_converged); bit-identical to what ritz_pairs returns for them.
    void ritz_values(Vector& evals, Vector& last_row) const
    {
        evals.resize(m_m);
        last_row.resize(m_m);
        internal::check(mispec_fac_ritz_values(m_fac.get(), evals.data(), last_row.data()));
    }

    // Implicit restart with the given shifts, already in the order they are to be applied
    // (HermEigsBase.h:118-147): per shift QR of H - mu I by Givens rotations, Q <- Q Qi, H <- Qi' H Qi —
    // one LDS-resident kernel — then V <- V Q and the new residual (Arnoldi.h:320-340).
    // Afterwards subspace_dim() == m - nshift.
    void restart_with_shifts(const Scalar* shifts, Index nshift)
    {
        internal::check(mispec_fac_restart_sym(m_fac.get(), shifts, static_cast<int>(nshift)));
    }
};

// The reference's spelling, Lanczos<ArnoldiOp<OpType, IdentityBOp>> (Lanczos.h:28-217, test/Arnoldi.cpp:108-158): the symmetric /
// Hermitian factorisation of the wrapped operator.
template <typename OpType>
class Lanczos<ArnoldiOp<OpType, IdentityBOp>> : public Arnoldi<ArnoldiOp<OpType, IdentityBOp>>
{
    using Base = Arnoldi<ArnoldiOp<OpType, IdentityBOp>>;

public:
    using Scalar = typename OpType::Scalar;

    Lanczos(const ArnoldiOp<OpType, IdentityBOp>& op, Index m) : Base(op, m, true) {}
};

}  // namespace Spectra

#endif
