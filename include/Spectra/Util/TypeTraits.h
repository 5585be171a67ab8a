// Floating-point limits used by every threshold of the solver (reference: Util/TypeTraits.h:37-88).
#ifndef MISPEC_SPECTRA_TYPE_TRAITS_H
#define MISPEC_SPECTRA_TYPE_TRAITS_H

#include <complex>
#include <limits>

namespace Spectra {

template <typename Scalar>
struct TypeTraits
{
    static constexpr Scalar epsilon() { return std::numeric_limits<Scalar>::epsilon(); }
    // smallest positive normal value; "near_0" in the solver is 10 * min()
    static constexpr Scalar(min)() { return (std::numeric_limits<Scalar>::min)(); }
};

// ElemType<double> = double, ElemType<std::complex<double>> = double
template <typename T>
struct ElemTypeOf
{
    using type = T;
};
template <typename T>
struct ElemTypeOf<std::complex<T>>
{
    using type = T;
};
template <typename T>
using ElemType = typename ElemTypeOf<T>::type;

}  // namespace Spectra

#endif
