// Shift-and-invert eigen solver for real symmetric matrices: eigenvalues closest to sigma, found as the
// largest-magnitude eigenvalues nu of (A - sigma I)^{-1} and mapped back by lambda = 1/nu + sigma.
// Same usage as the reference (SymEigsShiftSolver.h:22-200):
//
//     SparseSymShiftSolve<double> op(A);
//     SymEigsShiftSolver<SparseSymShiftSolve<double>> eigs(op, nev, ncv, sigma);   // calls op.set_shift(sigma)
//     eigs.init();
//     eigs.compute(SortRule::LargestMagn);            // "largest" refers to nu = 1 / (lambda - sigma)
//
// OpType needs, besides the usual members, `void set_shift(const Scalar& sigma)` (reference :118, :194).
#ifndef MISPEC_SPECTRA_SYM_EIGS_SHIFT_SOLVER_H
#define MISPEC_SPECTRA_SYM_EIGS_SHIFT_SOLVER_H

#include "HermEigsBase.h"
#include "MatOp/SparseSymShiftSolve.h"

namespace Spectra {

namespace internal {
// set_shift() has to run before the base class binds the operator's device factorisation
template <typename OpType, typename Scalar>
OpType& with_shift(OpType& op, const Scalar& sigma)
{
    op.set_shift(sigma);
    return op;
}
}  // namespace internal

template <typename OpType = SparseSymShiftSolve<double>>
class SymEigsShiftSolver : public HermEigsBase<OpType, IdentityBOp>
{
    using Scalar = typename OpType::Scalar;
    using Base = HermEigsBase<OpType, IdentityBOp>;
    using Base::m_nev;
    using Base::m_ritz_val;
    const Scalar m_sigma;

    // nu -> lambda = 1/nu + sigma, then the usual ordering (reference :163-169)
    void sort_ritzpair(SortRule sort_rule) override
    {
        for (Index i = 0; i < m_nev; i++)
            m_ritz_val[i] = Scalar(1) / m_ritz_val[i] + m_sigma;
        Base::sort_ritzpair(sort_rule);
    }

public:
    SymEigsShiftSolver(OpType& op, Index nev, Index ncv, const Scalar& sigma) :
        Base(internal::with_shift(op, sigma), IdentityBOp(), nev, ncv), m_sigma(sigma)
    {}
};

}  // namespace Spectra

#endif
