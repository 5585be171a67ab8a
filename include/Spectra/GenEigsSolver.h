// Eigen solver for general real matrices: the k eigenvalues of largest / smallest magnitude, real or
// imaginary part, by the implicitly-restarted Arnoldi method on the GPU.  Eigenvalues and eigenvectors
// are complex.  Same usage as the reference (GenEigsSolver.h:24-189):
//
//     SparseGenMatProd<double> op(A);
//     GenEigsSolver<SparseGenMatProd<double>> eigs(op, nev, ncv);   // ncv >= 2 nev + 1 advised
//     eigs.init();
//     int nconv = eigs.compute(SortRule::LargestMagn);
//     if (eigs.info() == CompInfo::Successful) { auto evalues = eigs.eigenvalues(); }
#ifndef MISPEC_SPECTRA_GEN_EIGS_SOLVER_H
#define MISPEC_SPECTRA_GEN_EIGS_SOLVER_H

#include "GenEigsBase.h"
#include "MatOp/SparseGenMatProd.h"

namespace Spectra {

template <typename OpType = SparseGenMatProd<double>>
class GenEigsSolver : public GenEigsBase<OpType, IdentityBOp>
{
public:
    // nev: 1 <= nev <= n-2 ; ncv: nev+2 <= ncv <= n.  Throws std::invalid_argument otherwise.
    GenEigsSolver(OpType& op, Index nev, Index ncv) : GenEigsBase<OpType, IdentityBOp>(op, IdentityBOp(), nev, ncv) {}
};

}  // namespace Spectra

#endif
