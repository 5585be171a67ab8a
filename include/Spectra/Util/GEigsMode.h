// Computation modes of the generalized eigen solvers (reference: Util/GEigsMode.h:16-23).
#ifndef MISPEC_SPECTRA_GEIGS_MODE_H
#define MISPEC_SPECTRA_GEIGS_MODE_H

namespace Spectra {

enum class GEigsMode
{
    Cholesky,        // Cholesky decomposition of B
    RegularInverse,  // B^{-1} A with B-inner products — the mode implemented on the device path
    ShiftInvert,
    Buckling,
    Cayley
};

}  // namespace Spectra

#endif
