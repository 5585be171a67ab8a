// Generalized eigen solver for real symmetric matrices: A x = lambda B x with A symmetric and B positive
// definite (reference: SymGEigsSolver.h:22-242).  Of the reference's modes the **regular-inverse** one runs on
// the device path:
//
//     SparseSymMatProd<double>      op(A);
//     SparseRegularInverse<double>  Bop(B);
//     SymGEigsSolver<SparseSymMatProd<double>, SparseRegularInverse<double>, GEigsMode::RegularInverse> geigs(op, Bop, nev, ncv);
//     geigs.init();  geigs.compute(SortRule::LargestAlge);  geigs.eigenvalues();  geigs.eigenvectors();
//
// The Lanczos process runs on y = B^{-1}(A x) with every inner product taken as x'By, so the Ritz vectors are
// B-orthonormal eigenvectors of the pencil and need no back-transformation (SymGEigsSolver.h:224-238).
// The Cholesky mode needs triangular solves with a sparse Cholesky factor of B and is not provided.
#ifndef MISPEC_SPECTRA_SYM_GEIGS_SOLVER_H
#define MISPEC_SPECTRA_SYM_GEIGS_SOLVER_H

#include "HermEigsBase.h"
#include "MatOp/internal/SymGEigsRegInvOp.h"
#include "Util/GEigsMode.h"

namespace Spectra {

// Empty class template; only the specialisations below exist (reference :31-33)
template <typename OpType, typename BOpType, GEigsMode Mode>
class SymGEigsSolver
{};

// Partial specialization for mode = GEigsMode::RegularInverse (reference :224-238)
template <typename OpType, typename BOpType>
class SymGEigsSolver<OpType, BOpType, GEigsMode::RegularInverse> : public HermEigsBase<SymGEigsRegInvOp<OpType, BOpType>, BOpType>
{
private:
    using ModeMatOp = SymGEigsRegInvOp<OpType, BOpType>;
    using Base = HermEigsBase<ModeMatOp, BOpType>;

public:
    // op: the A operator; Bop: the B operator (perform_op = B x, solve = B^{-1} x); 1 <= nev <= n-1; nev < ncv <= n
    SymGEigsSolver(OpType& op, BOpType& Bop, Index nev, Index ncv) : Base(ModeMatOp(op, Bop), Bop, nev, ncv) {}
};

}  // namespace Spectra

#endif
