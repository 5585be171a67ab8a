// y = Re((A - sigma I)^{-1} x) for a general real dense A and a complex shift — the dense operator of
// GenEigsComplexShiftSolver (reference: MatOp/DenseGenComplexShiftSolve.h:29-104, an Eigen::PartialPivLU of the complex
// matrix).  Here: complex LU on the host, the real part of the inverse in HBM, GEMV kernel; n <= 4096.
#ifndef MISPEC_SPECTRA_DENSE_GEN_COMPLEX_SHIFT_SOLVE_H
#define MISPEC_SPECTRA_DENSE_GEN_COMPLEX_SHIFT_SOLVE_H

#include "../../mispec_extras.h"  // outside the hot path of SURVEY.md section 8: declared apart from the thin shim
#include "DenseGenRealShiftSolve.h"

namespace Spectra {

template <typename Scalar_, int Flags = ColMajor>
class DenseGenComplexShiftSolve : public DenseGenRealShiftSolve<Scalar_, Flags>
{
    using Base = DenseGenRealShiftSolve<Scalar_, Flags>;

public:
    using Scalar = Scalar_;
    using Base::Base;

    void set_shift(const Scalar& sigmar, const Scalar& sigmai)
    {
        internal::check(mispec_symshift_set_shift_complex(const_cast<mispec_symshift*>(this->mispec_solver()), sigmar, sigmai));
    }
};

}  // namespace Spectra

#endif
