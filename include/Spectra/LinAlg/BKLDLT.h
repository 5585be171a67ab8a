// Bunch-Kaufman LDL' factorisation of a real symmetric (possibly indefinite) matrix A - shift*I, P (A - sI) P' = L D L'
// with 1x1 and 2x2 diagonal blocks, and the solver built on it.  Same class name and public members as the reference
// (LinAlg/BKLDLT.h:394-522: BKLDLT(mat, uplo, shift), compute(), solve_inplace(), solve(), info()); the reference uses it
// inside DenseSymShiftSolve.  Host code (the device shift-and-invert operators have their own factorisations); this is
// the partial-pivoting scheme of Bunch & Kaufman (1977) in the unblocked right-looking form: interchanges act on the
// trailing matrix only and are replayed during the solve.  Only the `uplo` triangle of the input is read.
#ifndef MISPEC_SPECTRA_BK_LDLT_H
#define MISPEC_SPECTRA_BK_LDLT_H

#include <cmath>
#include <stdexcept>
#include <vector>

#include "../Util/CompInfo.h"
#include "../internal/Dense.h"

namespace Spectra {

template <typename Scalar = double>
class BKLDLT
{
    using Matrix = DenseMatrix<Scalar>;
    using Vector = DenseVector<Scalar>;

    Index m_n;
    std::vector<Scalar> m_a;    // n x n column-major; the lower triangle holds L (unit diagonal implied) and D
    std::vector<Index> m_piv;   // per column: the row it was interchanged with; for a 2x2 block both entries are -(row + 1)
    bool m_computed;
    CompInfo m_info;

    Scalar& a(Index i, Index j) { return m_a[static_cast<std::size_t>(j) * m_n + i]; }
    const Scalar& a(Index i, Index j) const { return m_a[static_cast<std::size_t>(j) * m_n + i]; }

    // symmetric interchange of rows/columns p < q inside the trailing matrix that starts at column k (lower triangle only)
    void interchange(Index k, Index p, Index q)
    {
        if (p == q)
            return;
        using std::swap;
        for (Index j = k; j < p; j++)  // row segments left of p
            swap(a(p, j), a(q, j));
        for (Index i = p + 1; i < q; i++)  // the part between the two pivots: column p <-> row q
            swap(a(i, p), a(q, i));
        for (Index i = q + 1; i < m_n; i++)  // column segments below q
            swap(a(i, p), a(i, q));
        swap(a(p, p), a(q, q));
    }

public:
    BKLDLT() : m_n(0), m_computed(false), m_info(CompInfo::NotComputed) {}

    template <typename MatType>
    BKLDLT(const MatType& mat, int uplo = Lower, const Scalar& shift = Scalar(0)) : m_n(0), m_computed(false), m_info(CompInfo::NotComputed)
    {
        compute(mat, uplo, shift);
    }

    template <typename MatType>
    void compute(const MatType& mat, int uplo = Lower, const Scalar& shift = Scalar(0))
    {
        using std::abs;
        m_n = static_cast<Index>(mat.rows());
        if (m_n != static_cast<Index>(mat.cols()))
            throw std::invalid_argument("BKLDLT: matrix must be square");
        m_a.assign(static_cast<std::size_t>(m_n) * static_cast<std::size_t>(m_n), Scalar(0));
        for (Index j = 0; j < m_n; j++)
            for (Index i = j; i < m_n; i++)
                a(i, j) = (uplo == Lower) ? mat(i, j) : mat(j, i);
        for (Index i = 0; i < m_n; i++)
            a(i, i) -= shift;
        m_piv.assign(static_cast<std::size_t>(m_n), 0);
        m_computed = false;
        m_info = CompInfo::Successful;

        const Scalar alpha = (Scalar(1) + std::sqrt(Scalar(17))) / Scalar(8);
        Index k = 0;
        while (k < m_n)
        {
            // pivot choice
            const Scalar absakk = abs(a(k, k));
            Index imax = k;
            Scalar colmax = 0;
            for (Index i = k + 1; i < m_n; i++)
                if (abs(a(i, k)) > colmax)
                {
                    colmax = abs(a(i, k));
                    imax = i;
                }
            Index kstep = 1, kp = k;
            if (!(absakk > Scalar(0)) && !(colmax > Scalar(0)))
            {
                m_info = CompInfo::NumericalIssue;  // a zero pivot column: A - shift*I is singular
                return;
            }
            if (absakk < alpha * colmax)
            {
                Scalar rowmax = 0;
                for (Index j = k; j < imax; j++)
                    rowmax = abs(a(imax, j)) > rowmax ? abs(a(imax, j)) : rowmax;
                for (Index i = imax + 1; i < m_n; i++)
                    rowmax = abs(a(i, imax)) > rowmax ? abs(a(i, imax)) : rowmax;
                if (absakk >= alpha * colmax * (colmax / rowmax))
                    kp = k;
                else if (abs(a(imax, imax)) >= alpha * rowmax)
                    kp = imax;
                else
                {
                    kp = imax;
                    kstep = 2;
                }
            }
            const Index kk = k + kstep - 1;
            interchange(k, kk, kp);
            if (kstep == 1)
            {
                const Scalar d = a(k, k);
                if (d == Scalar(0))
                {
                    m_info = CompInfo::NumericalIssue;
                    return;
                }
                for (Index j = k + 1; j < m_n; j++)
                {
                    const Scalar w = a(j, k) / d;
                    if (w != Scalar(0))
                        for (Index i = j; i < m_n; i++)
                            a(i, j) -= a(i, k) * w;
                }
                for (Index i = k + 1; i < m_n; i++)
                    a(i, k) /= d;
                m_piv[static_cast<std::size_t>(k)] = kp;
            }
            else
            {
                // D = [d11 d21; d21 d22]; W = A(k+2:, k:k+1) * inv(D); trailing -= W * A(k+2:, k:k+1)'
                const Scalar d11 = a(k, k), d21 = a(k + 1, k), d22 = a(k + 1, k + 1);
                const Scalar det = d11 * d22 - d21 * d21;
                if (det == Scalar(0))
                {
                    m_info = CompInfo::NumericalIssue;
                    return;
                }
                for (Index j = k + 2; j < m_n; j++)
                {
                    const Scalar w1 = (a(j, k) * d22 - a(j, k + 1) * d21) / det;
                    const Scalar w2 = (a(j, k + 1) * d11 - a(j, k) * d21) / det;
                    for (Index i = j; i < m_n; i++)
                        a(i, j) -= a(i, k) * w1 + a(i, k + 1) * w2;
                    m_w1.resize(static_cast<std::size_t>(m_n));
                    m_w2.resize(static_cast<std::size_t>(m_n));
                    m_w1[static_cast<std::size_t>(j)] = w1;
                    m_w2[static_cast<std::size_t>(j)] = w2;
                }
                for (Index j = k + 2; j < m_n; j++)
                {
                    a(j, k) = m_w1[static_cast<std::size_t>(j)];
                    a(j, k + 1) = m_w2[static_cast<std::size_t>(j)];
                }
                m_piv[static_cast<std::size_t>(k)] = -(kp + 1);
                m_piv[static_cast<std::size_t>(k) + 1] = -(kp + 1);
            }
            k += kstep;
        }
        m_computed = true;
    }

    // b <- inv(A - shift*I) b
    void solve_inplace(Scalar* b) const
    {
        if (!m_computed)
            throw std::logic_error("BKLDLT: need to call compute() first");
        using std::swap;
        // forward: L and the interchanges, then the block diagonal
        Index k = 0;
        while (k < m_n)
        {
            if (m_piv[static_cast<std::size_t>(k)] >= 0)
            {
                swap(b[k], b[m_piv[static_cast<std::size_t>(k)]]);
                for (Index i = k + 1; i < m_n; i++)
                    b[i] -= a(i, k) * b[k];
                b[k] /= a(k, k);
                k += 1;
            }
            else
            {
                swap(b[k + 1], b[-m_piv[static_cast<std::size_t>(k)] - 1]);
                for (Index i = k + 2; i < m_n; i++)
                    b[i] -= a(i, k) * b[k] + a(i, k + 1) * b[k + 1];
                const Scalar d11 = a(k, k), d21 = a(k + 1, k), d22 = a(k + 1, k + 1);
                const Scalar det = d11 * d22 - d21 * d21;
                const Scalar x1 = (b[k] * d22 - b[k + 1] * d21) / det;
                const Scalar x2 = (b[k + 1] * d11 - b[k] * d21) / det;
                b[k] = x1;
                b[k + 1] = x2;
                k += 2;
            }
        }
        // backward: L' and the interchanges in reverse order
        k = m_n - 1;
        while (k >= 0)
        {
            if (m_piv[static_cast<std::size_t>(k)] >= 0)
            {
                Scalar s = 0;
                for (Index i = k + 1; i < m_n; i++)
                    s += a(i, k) * b[i];
                b[k] -= s;
                swap(b[k], b[m_piv[static_cast<std::size_t>(k)]]);
                k -= 1;
            }
            else
            {
                Scalar s1 = 0, s2 = 0;
                for (Index i = k + 1; i < m_n; i++)
                {
                    s1 += a(i, k - 1) * b[i];
                    s2 += a(i, k) * b[i];
                }
                b[k - 1] -= s1;
                b[k] -= s2;
                swap(b[k], b[-m_piv[static_cast<std::size_t>(k)] - 1]);
                k -= 2;
            }
        }
    }
    void solve_inplace(Vector& b) const { solve_inplace(b.data()); }

    Vector solve(const Vector& b) const
    {
        Vector res = b;
        solve_inplace(res);
        return res;
    }

    CompInfo info() const { return m_info; }

private:
    std::vector<Scalar> m_w1, m_w2;  // workspace of the 2x2 elimination
};

}  // namespace Spectra

#endif
