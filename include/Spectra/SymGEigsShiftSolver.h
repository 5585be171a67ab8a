// Generalized symmetric eigen solver with a spectral transformation (reference: SymGEigsShiftSolver.h:22-210):
// eigenvalues of A x = lambda B x closest to a shift sigma, in three modes
//   ShiftInvert  operator inv(A - sigma B) B,               nu = 1 / (lambda - sigma)
//   Buckling     operator inv(K - sigma KG) K (B-op = K),   nu = lambda / (lambda - sigma)
//   Cayley       operator inv(A - sigma B) (A + sigma B),   nu = (lambda + sigma) / (lambda - sigma)
//
//     SymShiftInvert<double>     op(A, B);          // factors A - sigma B on the GPU at set_shift()
//     SparseSymMatProd<double>   Bop(B);
//     SymGEigsShiftSolver<SymShiftInvert<double>, SparseSymMatProd<double>, GEigsMode::ShiftInvert> eigs(op, Bop, nev, ncv, sigma);
//     eigs.init();  eigs.compute(SortRule::LargestMagn);   // largest nu = eigenvalues closest to sigma
#ifndef MISPEC_SPECTRA_SYM_GEIGS_SHIFT_SOLVER_H
#define MISPEC_SPECTRA_SYM_GEIGS_SHIFT_SOLVER_H

#include "../mispec_extras.h"  // outside the hot path of SURVEY.md section 8: declared apart from the thin shim
#include <stdexcept>
#include <string>
#include <utility>

#include "HermEigsBase.h"
#include "MatOp/internal/SymGEigsShiftOps.h"
#include "Util/GEigsMode.h"

namespace Spectra {

// Empty class template; only the specialisations below exist (reference :31-33)
template <typename OpType, typename BOpType, GEigsMode Mode>
class SymGEigsShiftSolver
{};

namespace internal {
// set_shift() has to run before the base class binds the operator (reference: set_shift_and_move, :52-57)
template <typename ModeMatOp, typename Scalar>
ModeMatOp shifted(ModeMatOp&& op, const Scalar& sigma, bool nonzero_required, const char* mode)
{
    if (nonzero_required && sigma == Scalar(0))
        throw std::invalid_argument(std::string("SymGEigsShiftSolver: sigma cannot be zero in the ") + mode + " mode");
    op.set_shift(sigma);
    return std::move(op);
}
}  // namespace internal

// mode = GEigsMode::ShiftInvert (reference :36-78)
template <typename OpType, typename BOpType>
class SymGEigsShiftSolver<OpType, BOpType, GEigsMode::ShiftInvert> : public HermEigsBase<SymGEigsShiftInvertOp<OpType, BOpType>, BOpType>
{
    using Scalar = typename OpType::Scalar;
    using ModeMatOp = SymGEigsShiftInvertOp<OpType, BOpType>;
    using Base = HermEigsBase<ModeMatOp, BOpType>;
    using Base::m_nev;
    using Base::m_ritz_val;
    const Scalar m_sigma;

    // lambda = 1 / nu + sigma, then the usual ordering
    void sort_ritzpair(SortRule sort_rule) override
    {
        for (Index i = 0; i < m_nev; i++)
            m_ritz_val[i] = Scalar(1) / m_ritz_val[i] + m_sigma;
        Base::sort_ritzpair(sort_rule);
    }

public:
    SymGEigsShiftSolver(OpType& op, BOpType& Bop, Index nev, Index ncv, const Scalar& sigma) :
        Base(internal::shifted(ModeMatOp(op, Bop), sigma, false, "shift-invert"), Bop, nev, ncv), m_sigma(sigma)
    {}
};

// mode = GEigsMode::Buckling (reference :82-129)
template <typename OpType, typename BOpType>
class SymGEigsShiftSolver<OpType, BOpType, GEigsMode::Buckling> : public HermEigsBase<SymGEigsBucklingOp<OpType, BOpType>, BOpType>
{
    using Scalar = typename OpType::Scalar;
    using ModeMatOp = SymGEigsBucklingOp<OpType, BOpType>;
    using Base = HermEigsBase<ModeMatOp, BOpType>;
    using Base::m_nev;
    using Base::m_ritz_val;
    const Scalar m_sigma;

    // lambda = sigma nu / (nu - 1)
    void sort_ritzpair(SortRule sort_rule) override
    {
        for (Index i = 0; i < m_nev; i++)
            m_ritz_val[i] = m_sigma * m_ritz_val[i] / (m_ritz_val[i] - Scalar(1));
        Base::sort_ritzpair(sort_rule);
    }

public:
    SymGEigsShiftSolver(OpType& op, BOpType& Bop, Index nev, Index ncv, const Scalar& sigma) :
        Base(internal::shifted(ModeMatOp(op, Bop), sigma, true, "buckling"), Bop, nev, ncv), m_sigma(sigma)
    {}
};

// mode = GEigsMode::Cayley (reference :133-180)
template <typename OpType, typename BOpType>
class SymGEigsShiftSolver<OpType, BOpType, GEigsMode::Cayley> : public HermEigsBase<SymGEigsCayleyOp<OpType, BOpType>, BOpType>
{
    using Scalar = typename OpType::Scalar;
    using ModeMatOp = SymGEigsCayleyOp<OpType, BOpType>;
    using Base = HermEigsBase<ModeMatOp, BOpType>;
    using Base::m_nev;
    using Base::m_ritz_val;
    const Scalar m_sigma;

    // lambda = sigma (nu + 1) / (nu - 1)
    void sort_ritzpair(SortRule sort_rule) override
    {
        for (Index i = 0; i < m_nev; i++)
            m_ritz_val[i] = m_sigma * (m_ritz_val[i] + Scalar(1)) / (m_ritz_val[i] - Scalar(1));
        Base::sort_ritzpair(sort_rule);
    }

public:
    SymGEigsShiftSolver(OpType& op, BOpType& Bop, Index nev, Index ncv, const Scalar& sigma) :
        Base(internal::shifted(ModeMatOp(op, Bop), sigma, true, "Cayley"), Bop, nev, ncv), m_sigma(sigma)
    {}
};

}  // namespace Spectra

#endif
