// Numerically stable Givens rotation used by the shifted-QR sweeps (reference: LinAlg/Givens.h:149-206).
//   G = [ c  s ; -s  c ],  G' [x; y] = [r; 0]  with  r = sqrt(x^2 + y^2) >= 0,  c = x / r,  s = -y / r
// The arithmetic lives in internal/SmallDense.h, shared with the LDS-resident device kernels.
#ifndef MISPEC_SPECTRA_GIVENS_H
#define MISPEC_SPECTRA_GIVENS_H

#include <complex>

#include "../internal/SmallDenseComplex.h"

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

// Complex operands: G = [c s; -conj(s) c] with real c, G^H [x; y] = [r; 0]  (reference: LinAlg/Givens.h:236-339)
template <typename RealScalar>
class Givens<std::complex<RealScalar>>
{
    using Scalar = std::complex<RealScalar>;

public:
    static void compute_rotation(const Scalar& x, const Scalar& y, Scalar& r, RealScalar& c, Scalar& s)
    {
        std::complex<double> rr, ss;
        double cc;
        mispec::small::givens_rotation_complex(std::complex<double>(x), std::complex<double>(y), rr, cc, ss);
        r = Scalar(rr);
        c = RealScalar(cc);
        s = Scalar(ss);
    }
};

}  // namespace Spectra

#endif
