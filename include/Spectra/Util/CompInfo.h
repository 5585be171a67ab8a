// Status of a computation — same enumerators, same order as yixuan/spectra v1.2.0
// (include/Spectra/Util/CompInfo.h:17-32), so integer values survive the C ABI.
#ifndef MISPEC_SPECTRA_COMP_INFO_H
#define MISPEC_SPECTRA_COMP_INFO_H

namespace Spectra {

enum class CompInfo
{
    Successful,     // everything converged
    NotComputed,    // compute() has not been called yet
    NotConverging,  // fewer than nev Ritz pairs met the tolerance within maxit restarts
    NumericalIssue  // a factorisation broke down (shift-solve operators)
};

}  // namespace Spectra

#endif
