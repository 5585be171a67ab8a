// Symmetric specialisation of the device factorisation: three-term recurrence, always-on V'f check with
// up to five corrections, tridiagonal H (reference: LinAlg/Lanczos.h:28-217).  Adds the two restart
// primitives on the tridiagonal H.  Where their m x m arithmetic runs: on a host core by default — serial chains of
// rotations that a 2.4 GHz lane runs several times slower than a core (measured: profiles/r11g_restart_sweeps_latency.jsonl)
// — with H coming home through pinned memory and Q going back the same way; option small=device keeps it on the GPU
// (spectra_amd/csrc/small.hip: k_tridiag_eigen*, k_restart_sym*, k_restart_pipelined).  The n-sized work (V*Q, the residual)
// is always the device's.
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

    // Eigen-decomposition of the projected tridiagonal H (TridiagEigen, LinAlg/TridiagEigen.h:121-210):
    // internal/SmallDense.h tridiag_eigen on the host core (default) or as a one-workgroup LDS kernel (small=device).
    // evals has m entries, evecs is m x m.
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
    // all shifts as one skewed pipeline (internal/SmallDensePipelined.h; bit-identical to the serial order), on the
    // host core by default — then V <- V Q and the new residual on the device (Arnoldi.h:320-340), with the next
    // sweep of Lanczos steps enqueued behind it.
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
