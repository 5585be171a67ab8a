// Eigen solver for general real matrices with a real shift: the eigenvalues of A closest to sigma, by the
// implicitly-restarted Arnoldi method on (A - sigma I)^{-1} (reference: GenEigsRealShiftSolver.h:22-86).
//
//     SparseGenRealShiftSolve<double> op(A);                         // factors A - sigma I on the GPU at set_shift()
//     GenEigsRealShiftSolver<SparseGenRealShiftSolve<double>> eigs(op, nev, ncv, sigma);
//     eigs.init();  eigs.compute(SortRule::LargestMagn);             // largest nu = eigenvalues closest to sigma
#ifndef MISPEC_SPECTRA_GEN_EIGS_REAL_SHIFT_SOLVER_H
#define MISPEC_SPECTRA_GEN_EIGS_REAL_SHIFT_SOLVER_H

#include "GenEigsBase.h"
#include "MatOp/SparseGenRealShiftSolve.h"

namespace Spectra {

namespace internal {
// set_shift() has to run before the base class binds the operator's device factorisation
template <typename OpType, typename Scalar>
OpType& gen_with_shift(OpType& op, const Scalar& sigma)
{
    op.set_shift(sigma);
    return op;
}
}  // namespace internal

template <typename OpType = SparseGenRealShiftSolve<double>>
class GenEigsRealShiftSolver : public GenEigsBase<OpType, IdentityBOp>
{
    using Scalar = typename OpType::Scalar;
    using Complex = std::complex<Scalar>;
    using Base = GenEigsBase<OpType, IdentityBOp>;
    using Base::m_nev;
    using Base::m_ritz_val;
    const Scalar m_sigma;

    // nu = 1 / (lambda - sigma)  ->  lambda = 1 / nu + sigma, then the usual ordering (reference :52-58)
    void sort_ritzpair(SortRule sort_rule) override
    {
        for (Index i = 0; i < m_nev; i++)
            m_ritz_val[i] = Complex(1) / m_ritz_val[i] + m_sigma;
        Base::sort_ritzpair(sort_rule);
    }

public:
    GenEigsRealShiftSolver(OpType& op, Index nev, Index ncv, const Scalar& sigma) :
        Base(internal::gen_with_shift(op, sigma), IdentityBOp(), nev, ncv), m_sigma(sigma)
    {}
};

}  // namespace Spectra

#endif
