// TEST INFRASTRUCTURE — NOT EIGEN.  See ../Core.  JacobiRotation with the scalar algorithms Eigen 3.4.0 publishes in
// Eigen/src/Jacobi/Jacobi.h (makeGivens, real and complex branches); the plane rotations themselves are members of LvalueBase.
#ifndef ORACLE_EIGEN_SHIM_JACOBI_H
#define ORACLE_EIGEN_SHIM_JACOBI_H

namespace Eigen {

template <typename Scalar>
class JacobiRotation
{
    Scalar m_c, m_s;
    typedef typename NumTraits<Scalar>::Real RealScalar;

public:
    JacobiRotation() : m_c(Scalar(1)), m_s(Scalar(0)) {}
    JacobiRotation(const Scalar& c, const Scalar& s) : m_c(c), m_s(s) {}
    Scalar& c() { return m_c; }
    Scalar c() const { return m_c; }
    Scalar& s() { return m_s; }
    Scalar s() const { return m_s; }
    JacobiRotation operator*(const JacobiRotation& other)
    {
        return JacobiRotation(m_c * other.m_c - numext::conj(m_s) * other.m_s,
                              numext::conj(m_c * numext::conj(other.m_s) + numext::conj(m_s) * numext::conj(other.m_c)));
    }
    JacobiRotation transpose() const { return JacobiRotation(m_c, -numext::conj(m_s)); }
    JacobiRotation adjoint() const { return JacobiRotation(numext::conj(m_c), -m_s); }

    void makeGivens(const Scalar& p, const Scalar& q, Scalar* r = 0)
    {
        make_givens(p, q, r, typename internal::is_complex<Scalar>::type());
    }
    // J = [c s; -s c] such that J^* [x y; y z] J is diagonal (real symmetric 2x2), Jacobi.h makeJacobi
    bool makeJacobi(const RealScalar& x, const Scalar& y, const RealScalar& z)
    {
        const RealScalar deno = RealScalar(2) * numext::abs(y);
        if (deno < (std::numeric_limits<RealScalar>::min)())
        {
            m_c = Scalar(1);
            m_s = Scalar(0);
            return false;
        }
        const RealScalar tau = (x - z) / deno;
        const RealScalar w = std::sqrt(numext::abs2(tau) + RealScalar(1));
        const RealScalar t = tau > RealScalar(0) ? RealScalar(1) / (tau + w) : RealScalar(1) / (tau - w);
        const RealScalar sign_t = t > RealScalar(0) ? RealScalar(1) : RealScalar(-1);
        const RealScalar n = RealScalar(1) / std::sqrt(numext::abs2(t) + RealScalar(1));
        m_s = -sign_t * (numext::conj(y) / numext::abs(y)) * numext::abs(t) * n;
        m_c = n;
        return true;
    }

private:
    void make_givens(const Scalar& p, const Scalar& q, Scalar* r, std::false_type)
    {
        using std::abs;
        using std::sqrt;
        if (q == Scalar(0))
        {
            m_c = p < Scalar(0) ? Scalar(-1) : Scalar(1);
            m_s = Scalar(0);
            if (r)
                *r = abs(p);
        }
        else if (p == Scalar(0))
        {
            m_c = Scalar(0);
            m_s = q < Scalar(0) ? Scalar(1) : Scalar(-1);
            if (r)
                *r = abs(q);
        }
        else if (abs(p) > abs(q))
        {
            Scalar t = q / p;
            Scalar u = sqrt(Scalar(1) + numext::abs2(t));
            if (p < Scalar(0))
                u = -u;
            m_c = Scalar(1) / u;
            m_s = -t * m_c;
            if (r)
                *r = p * u;
        }
        else
        {
            Scalar t = p / q;
            Scalar u = sqrt(Scalar(1) + numext::abs2(t));
            if (q < Scalar(0))
                u = -u;
            m_s = -Scalar(1) / u;
            m_c = -t * m_s;
            if (r)
                *r = q * u;
        }
    }
    void make_givens(const Scalar& p, const Scalar& q, Scalar* r, std::true_type)
    {
        using std::abs;
        using std::sqrt;
        if (q == Scalar(0))
        {
            m_c = numext::real(p) < 0 ? Scalar(-1) : Scalar(1);
            m_s = 0;
            if (r)
                *r = m_c * p;
        }
        else if (p == Scalar(0))
        {
            m_c = 0;
            m_s = -q / abs(q);
            if (r)
                *r = abs(q);
        }
        else
        {
            RealScalar p1 = numext::norm1(p);
            RealScalar q1 = numext::norm1(q);
            if (p1 >= q1)
            {
                Scalar ps = p / p1;
                RealScalar p2 = numext::abs2(ps);
                Scalar qs = q / p1;
                RealScalar q2 = numext::abs2(qs);
                RealScalar u = sqrt(RealScalar(1) + q2 / p2);
                if (numext::real(p) < RealScalar(0))
                    u = -u;
                m_c = Scalar(1) / u;
                m_s = -qs * numext::conj(ps) * (m_c / p2);
                if (r)
                    *r = p * u;
            }
            else
            {
                Scalar ps = p / q1;
                RealScalar p2 = numext::abs2(ps);
                Scalar qs = q / q1;
                RealScalar q2 = numext::abs2(qs);
                RealScalar u = q1 * sqrt(p2 + q2);
                if (numext::real(p) < RealScalar(0))
                    u = -u;
                p1 = abs(p);
                ps = p / p1;
                m_c = p1 / u;
                m_s = -numext::conj(ps) * (q / u);
                if (r)
                    *r = ps * u;
            }
        }
    }
};

}  // namespace Eigen

#endif
