// y = Re((A - sigma I)^{-1} x) for a general real sparse A and a complex shift sigma = sigmar + i sigmai — the operator of
// GenEigsComplexShiftSolver.  Same members as the reference class (MatOp/SparseGenComplexShiftSolve.h:35-113: rows(),
// cols(), set_shift(sigmar, sigmai), perform_op()), which factors a complex Eigen::SparseLU; here the dense device path
// of SparseGenRealShiftSolve is used with the real part of the complex inverse (n <= 4096).
#ifndef MISPEC_SPECTRA_SPARSE_GEN_COMPLEX_SHIFT_SOLVE_H
#define MISPEC_SPECTRA_SPARSE_GEN_COMPLEX_SHIFT_SOLVE_H

#include "../../mispec_extras.h"  // outside the hot path of SURVEY.md section 8: declared apart from the thin shim
#include "SparseGenRealShiftSolve.h"

namespace Spectra {

template <typename Scalar_, int Flags = ColMajor, typename StorageIndex = int>
class SparseGenComplexShiftSolve : public SparseGenRealShiftSolve<Scalar_, Flags, StorageIndex>
{
    using Base = SparseGenRealShiftSolve<Scalar_, Flags, StorageIndex>;

public:
    using Scalar = Scalar_;
    using Base::Base;

    // Factor A - (sigmar + i sigmai) I; throws std::invalid_argument if that fails (reference :96-98)
    void set_shift(const Scalar& sigmar, const Scalar& sigmai)
    {
        internal::check(mispec_symshift_set_shift_complex(const_cast<mispec_symshift*>(this->mispec_solver()), sigmar, sigmai));
    }
};

}  // namespace Spectra

#endif
