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
// The **Cholesky** mode (SymGEigsSolver.h:142-208) solves the standard problem L^{-1} A L^{-T} y = lambda y with
// B = L L' and returns x = L^{-T} y; its device B operator holds a dense factor (n <= 4096) or, for a banded B of any
// size, the partitioned band factor:
//
//     SparseCholesky<double> Bop(B);
//     SymGEigsSolver<SparseSymMatProd<double>, SparseCholesky<double>, GEigsMode::Cholesky> geigs(op, Bop, nev, ncv);
#ifndef MISPEC_SPECTRA_SYM_GEIGS_SOLVER_H
#define MISPEC_SPECTRA_SYM_GEIGS_SOLVER_H

#include <vector>

#include "HermEigsBase.h"
#include "MatOp/internal/SymGEigsCholeskyOp.h"
#include "MatOp/internal/SymGEigsRegInvOp.h"
#include "Util/GEigsMode.h"

namespace Spectra {

// Empty class template; only the specialisations below exist (reference :31-33)
template <typename OpType, typename BOpType, GEigsMode Mode>
class SymGEigsSolver
{};

// Partial specialization for mode = GEigsMode::Cholesky (reference :142-208)
template <typename OpType, typename BOpType>
class SymGEigsSolver<OpType, BOpType, GEigsMode::Cholesky> : public HermEigsBase<SymGEigsCholeskyOp<OpType, BOpType>, IdentityBOp>
{
private:
    using Scalar = typename OpType::Scalar;
    using Matrix = DenseMatrix<Scalar>;
    using ModeMatOp = SymGEigsCholeskyOp<OpType, BOpType>;
    using Base = HermEigsBase<ModeMatOp, IdentityBOp>;
    const BOpType& m_Bop;

public:
    SymGEigsSolver(OpType& op, BOpType& Bop, Index nev, Index ncv) : Base(ModeMatOp(op, Bop), IdentityBOp(), nev, ncv), m_Bop(Bop) {}

    // eigenvectors of the pencil: x = L^{-T} y  (reference :186-199)
    Matrix eigenvectors(Index nvec) const override
    {
        Matrix res = Base::eigenvectors(nvec);
        std::vector<Scalar> in(static_cast<std::size_t>(res.rows())), out(static_cast<std::size_t>(res.rows()));
        for (Index i = 0; i < res.cols(); i++)
        {
            for (Index r = 0; r < res.rows(); r++)
                in[static_cast<std::size_t>(r)] = res(r, i);
            m_Bop.upper_triangular_solve(in.data(), out.data());
            for (Index r = 0; r < res.rows(); r++)
                res(r, i) = out[static_cast<std::size_t>(r)];
        }
        return res;
    }
    Matrix eigenvectors() const override { return SymGEigsSolver<OpType, BOpType, GEigsMode::Cholesky>::eigenvectors(this->m_nev); }
    Index eigenvectors_to(Scalar* out_host, Index nvec) const override  // the back-transformed vectors, copied out
    {
        const Matrix res = eigenvectors(nvec);
        for (Index i = 0; i < res.cols(); i++)
            for (Index r = 0; r < res.rows(); r++)
                out_host[static_cast<std::size_t>(i) * static_cast<std::size_t>(res.rows()) + static_cast<std::size_t>(r)] = res(r, i);
        return res.cols();
    }
};

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
