// TEST INFRASTRUCTURE — NOT EIGEN.  See ../Core.  makeHouseholder with the algorithm Eigen 3.4.0 publishes in
// Eigen/src/Householder/Householder.h, and one-coefficient "packets" for the code that names Eigen::internal::p* directly.
#ifndef ORACLE_EIGEN_SHIM_HOUSEHOLDER_H
#define ORACLE_EIGEN_SHIM_HOUSEHOLDER_H

namespace Eigen {

// v.makeHouseholder(essential, tau, beta): H = I - tau [1; essential] [1; essential]^*, H v = [beta; 0]
template <typename Derived, typename EssentialPart>
void make_householder(const MatrixBase<Derived>& v, EssentialPart& essential, typename internal::traits<Derived>::Scalar& tau,
                      typename NumTraits<typename internal::traits<Derived>::Scalar>::Real& beta)
{
    typedef typename internal::traits<Derived>::Scalar Scalar;
    typedef typename NumTraits<Scalar>::Real RealScalar;
    using std::sqrt;
    const Index n = v.size();
    RealScalar tailSqNorm = RealScalar(0);
    for (Index i = 1; i < n; i++)
        tailSqNorm += numext::abs2(v.coeff(i));
    const Scalar c0 = v.coeff(0);
    const RealScalar tol = (std::numeric_limits<RealScalar>::min)();
    if (tailSqNorm <= tol && numext::abs2(numext::imag(c0)) <= tol)
    {
        tau = Scalar(RealScalar(0));
        beta = numext::real(c0);
        essential.setZero();
    }
    else
    {
        beta = sqrt(numext::abs2(c0) + tailSqNorm);
        if (numext::real(c0) >= RealScalar(0))
            beta = -beta;
        for (Index i = 1; i < n; i++)
            essential.coeffRef(i - 1) = v.coeff(i) / (c0 - beta);
        tau = numext::conj((beta - c0) / beta);
    }
}

template <typename Derived>
template <typename EssentialPart>
void MatrixBase<Derived>::makeHouseholder(EssentialPart& essential, Scalar& tau, RealScalar& beta) const
{
    make_householder(*this, essential, tau, beta);
}

namespace internal {
template <typename T>
struct packet_traits
{
    typedef T type;
    enum
    {
        size = 1,
        Vectorizable = 0
    };
};
template <typename P>
inline P ploadu(const P* p) { return *p; }
template <typename P>
inline void pstoreu(P* p, const P& v) { *p = v; }
template <typename P>
inline P pset1(const P& v) { return v; }
template <typename P>
inline P padd(const P& a, const P& b) { return a + b; }
template <typename P>
inline P psub(const P& a, const P& b) { return a - b; }
template <typename P>
inline P pmul(const P& a, const P& b) { return a * b; }
}  // namespace internal

}  // namespace Eigen

#endif
