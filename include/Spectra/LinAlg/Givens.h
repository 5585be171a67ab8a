// Numerically stable Givens rotation used by the shifted-QR sweeps (reference: LinAlg/Givens.h:149-206).
//   G = [ c  s ; -s  c ],  G' [x; y] = [r; 0]  with  r = sqrt(x^2 + y^2) >= 0,  c = x / r,  s = -y / r
// The arithmetic lives in internal/SmallDense.h, shared with the LDS-resident device kernels.
#ifndef MISPEC_SPECTRA_GIVENS_H
#define MISPEC_SPECTRA_GIVENS_H

#include "../internal/SmallDense.h"

namespace Spectra {

template <typename Scalar>
class Givens
{
public:
    static void compute_rotation(const Scalar& x, const Scalar& y, Scalar& r, Scalar& c, Scalar& s)
    {
        double rr, cc, ss;
        mispec::small::givens_rotation(double(x), double(y), rr, cc, ss);
        r = Scalar(rr);
        c = Scalar(cc);
        s = Scalar(ss);
    }
};

}  // namespace Spectra

#endif
