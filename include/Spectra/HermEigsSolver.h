// Eigen solver for Hermitian matrices (reference: HermEigsSolver.h:21-160, the complex-capable sibling of SymEigsSolver).
// The MI355X path computes in real fp64 only: with a real operator this class is the same implicitly-restarted Lanczos
// solver as SymEigsSolver; an operator whose Scalar is complex is rejected at compile time (the complex Hermitian
// operators DenseHermMatProd / SparseHermMatProd are not built).
#ifndef MISPEC_SPECTRA_HERM_EIGS_SOLVER_H
#define MISPEC_SPECTRA_HERM_EIGS_SOLVER_H

#include <type_traits>

#include "HermEigsBase.h"
#include "MatOp/SparseSymMatProd.h"

namespace Spectra {

template <typename OpType = SparseSymMatProd<double>>
class HermEigsSolver : public HermEigsBase<OpType, IdentityBOp>
{
    static_assert(std::is_same<typename OpType::Scalar, double>::value,
                  "HermEigsSolver: only real double-precision operators run on the device path (complex Hermitian matrices are not supported)");

public:
    // op: the matrix operator; 1 <= nev <= n-1; nev < ncv <= n.  Throws std::invalid_argument otherwise (reference :150-152)
    HermEigsSolver(OpType& op, Index nev, Index ncv) : HermEigsBase<OpType, IdentityBOp>(op, IdentityBOp(), nev, ncv) {}
};

}  // namespace Spectra

#endif
