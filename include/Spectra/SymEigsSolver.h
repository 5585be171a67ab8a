// Eigen solver for real symmetric matrices: the k largest / smallest eigenvalues of A given through an
// operator object, by the implicitly-restarted Lanczos method on the GPU.
//
// Same usage as the reference (SymEigsSolver.h:31-160):
//
//     SparseSymMatProd<double> op(A);            // or any class with rows(), cols(), perform_op()
//     SymEigsSolver<SparseSymMatProd<double>> eigs(op, nev, ncv);
//     eigs.init();
//     int nconv = eigs.compute(SortRule::LargestAlge);
//     if (eigs.info() == CompInfo::Successful) { auto evalues = eigs.eigenvalues(); auto evecs = eigs.eigenvectors(); }
//
// A user-defined OpType only needs  `using Scalar = double;  Index rows() const;  Index cols() const;
// void perform_op(const double* x_in, double* y_out) const;`  (SymEigsSolver.h:43-51).
#ifndef MISPEC_SPECTRA_SYM_EIGS_SOLVER_H
#define MISPEC_SPECTRA_SYM_EIGS_SOLVER_H

#include "HermEigsBase.h"
#include "MatOp/SparseSymMatProd.h"

namespace Spectra {

template <typename OpType = SparseSymMatProd<double>>
class SymEigsSolver : public HermEigsBase<OpType, IdentityBOp>
{
public:
    // op: the matrix operator; nev: number of eigenvalues wanted, 1 <= nev <= n-1;
    // ncv: Krylov dimension, nev < ncv <= n (ncv >= 2 nev advised).  Throws std::invalid_argument otherwise.
    SymEigsSolver(OpType& op, Index nev, Index ncv) : HermEigsBase<OpType, IdentityBOp>(op, IdentityBOp(), nev, ncv) {}
};

}  // namespace Spectra

#endif
