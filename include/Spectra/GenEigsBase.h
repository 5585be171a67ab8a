// Implicitly-restarted Arnoldi driver for general real matrices, GPU-resident factorisation.
//
// Public surface and semantics follow the reference's GenEigsBase (GenEigsBase.h:140-611): constructor
// checks (:419-423), init() / init(v0) (:442-476), compute(selection, maxit, tol, sorting) (:501-525),
// info(), num_iterations(), num_operations(), eigenvalues() and eigenvectors() returning complex data
// (:548-610), and the virtual sort_ritzpair hook (:345-401).
//
// Where things run: the Arnoldi factorisation (SpMV, h = V'w, f = w - Vh, re-orthogonalisation), V <- VQ
// and the eigenvector products V*Re(Y), V*Im(Y) are HIP kernels on HBM-resident data (LinAlg/Arnoldi.h);
// the ncv x ncv Hessenberg work of a restart (real shifts by Givens QR, conjugate pairs by a Francis
// double-shift step, Ritz pairs through the real Schur form) runs on the host on ~1e5 flops and hands
// Q (ncv^2 doubles) to the device — see internal/SmallDenseGen.h.
#ifndef MISPEC_SPECTRA_GEN_EIGS_BASE_H
#define MISPEC_SPECTRA_GEN_EIGS_BASE_H

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <string>
#include <stdexcept>
#include <vector>

#include "HermEigsBase.h"  // IdentityBOp
#include "LinAlg/Arnoldi.h"
#include "LinAlg/UpperHessenbergEigen.h"
#include "Util/CompInfo.h"
#include "Util/SelectionRule.h"
#include "Util/TypeTraits.h"
#include "internal/Dense.h"
#include "internal/SmallDenseGen.h"

namespace Spectra {

template <typename OpType, typename BOpType = IdentityBOp>
class GenEigsBase
{
    static_assert(std::is_same<BOpType, IdentityBOp>::value, "only standard problems (B = I) run on the device path");

private:
    using Scalar = typename OpType::Scalar;
    using RealScalar = ElemType<Scalar>;
    using Complex = std::complex<RealScalar>;
    using Matrix = DenseMatrix<Scalar>;
    using ComplexMatrix = DenseMatrix<Complex>;
    using ComplexVector = DenseVector<Complex>;
    using ArnoldiFac = Arnoldi<OpType>;

protected:
    OpType& m_op;
    const Index m_n;
    const Index m_nev;
    const Index m_ncv;
    Index m_nmatop;
    Index m_niter;
    ArnoldiFac m_fac;
    ComplexVector m_ritz_val;  // Ritz values, wanted ones first
    ComplexMatrix m_ritz_vec;  // Ritz vectors of H for the nev wanted values
    ComplexVector m_ritz_est;  // last row of the eigenvector matrix of H

private:
    std::vector<char> m_ritz_conv;
    CompInfo m_info;

    // Real Ritz values come with an exactly zero imaginary part and complex ones in exact conjugate
    // pairs (that is how the Schur-based decomposition builds them), so exact tests are right here.
    static bool is_complex(const Complex& v) { return v.imag() != RealScalar(0); }
    static bool is_conj(const Complex& a, const Complex& b) { return a == std::conj(b); }

    // LIMIT of this implementation (not of the reference): the device factorisation holds at most 1024 basis vectors, so a
    // solver constructed with ncv > 1024 throws std::invalid_argument from the factorisation's constructor
    // (mispec_fac_create: "ncv <= 1024"); the reference accepts any nev < ncv <= n.
    static Index check_args(Index n, Index nev, Index ncv)
    {
        if (nev < 1 || nev > n - 2)
            throw std::invalid_argument("nev must satisfy 1 <= nev <= n - 2, n is the size of matrix");
        if (ncv < nev + 2 || ncv > n)
            throw std::invalid_argument("ncv must satisfy nev + 2 <= ncv <= n, n is the size of matrix");
        return ncv;
    }

    // Ritz pairs of the Hessenberg H, wanted ones first (reference :280-340)
    void retrieve_ritzpair(SortRule selection)
    {
        UpperHessenbergEigen<RealScalar> decomp(m_fac.matrix_H());
        const ComplexVector& evals = decomp.eigenvalues();
        const ComplexMatrix evecs = decomp.eigenvectors();
        const std::vector<Index> ind = argsort(selection, evals.data(), m_ncv);
        for (Index i = 0; i < m_ncv; i++)
        {
            m_ritz_val[i] = evals[ind[i]];
            m_ritz_est[i] = evecs(m_ncv - 1, ind[i]);
        }
        for (Index i = 0; i < m_nev; i++)
            for (Index r = 0; r < m_ncv; r++)
                m_ritz_vec(r, i) = evecs(r, ind[i]);
    }

    Index num_converged(RealScalar tol)  // reference :225-242
    {
        const RealScalar eps23 = std::pow(TypeTraits<RealScalar>::epsilon(), RealScalar(2) / 3);
        const RealScalar fnorm = m_fac.f_norm();
        Index count = 0;
        for (Index i = 0; i < m_nev; i++)
        {
            const RealScalar thresh = tol * (std::max)(eps23, std::abs(m_ritz_val[i]));
            m_ritz_conv[static_cast<std::size_t>(i)] = (std::abs(m_ritz_est[i]) * fnorm < thresh) ? 1 : 0;
            count += m_ritz_conv[static_cast<std::size_t>(i)];
        }
        return count;
    }

    // ARPACK dnaup2 heuristic; never splits a conjugate pair (reference :245-277)
    Index nev_adjusted(Index nconv)
    {
        const RealScalar near_0 = TypeTraits<RealScalar>::min() * RealScalar(10);
        Index nev_new = m_nev;
        for (Index i = m_nev; i < m_ncv; i++)
            if (std::abs(m_ritz_est[i]) < near_0)
                nev_new++;
        nev_new += (std::min)(nconv, (m_ncv - nev_new) / 2);
        if (nev_new == 1 && m_ncv >= 6)
            nev_new = m_ncv / 2;
        else if (nev_new == 1 && m_ncv > 3)
            nev_new = 2;
        if (nev_new > m_ncv - 2)
            nev_new = m_ncv - 2;
        if (is_complex(m_ritz_val[nev_new - 1]) && is_conj(m_ritz_val[nev_new - 1], m_ritz_val[nev_new]))
            nev_new++;
        return nev_new;
    }

    // One implicit restart keeping k Ritz pairs: shifts are the unwanted Ritz values in their sorted
    // order, conjugate pairs applied together (reference :204-222 and RestartArnoldi :60-107)
    void restart(Index k, SortRule selection)
    {
        if (k >= m_ncv)
            return;
        const int m = static_cast<int>(m_ncv);
        // Where the ncv x ncv sweeps run: the option small=device (mispec_set_option) applies the whole shift list in one LDS-resident kernel
        // (ncv <= 96, spectra_amd/csrc/small.hip k_hess_restart; H and Q never leave the device between the sweeps and
        // V <- V Q); the default is the host (same arithmetic, internal/SmallDenseGen.h) because one wavefront stepping
        // through a serial chain of reflectors is slower than one host core at these sizes (DESIGN.md 3.4 has the numbers).
        const char* where = mispec_get_option("small");
        if (where && std::string(where) == "device" && m <= 96)
        {
            std::vector<int> kind;
            std::vector<double> sa, sb;
            Index kk = m_ncv;
            for (Index i = k; i < m_ncv; i++)
            {
                if (is_complex(m_ritz_val[i]) && i + 1 < m_ncv && is_conj(m_ritz_val[i], m_ritz_val[i + 1]))
                {
                    kind.push_back(1);
                    sa.push_back(RealScalar(2) * m_ritz_val[i].real());
                    sb.push_back(std::norm(m_ritz_val[i]));
                    kk -= 2;
                    i++;
                }
                else
                {
                    kind.push_back(0);
                    sa.push_back(m_ritz_val[i].real());
                    sb.push_back(0.0);
                    kk -= 1;
                }
            }
            m_fac.restart_gen(kind, sa, sb, kk);
            m_fac.factorize_from(k, m_ncv, m_nmatop);
            retrieve_ritzpair(selection);
            return;
        }
        Matrix H = m_fac.matrix_H();
        Matrix Q(m_ncv, m_ncv);
        for (Index j = 0; j < m_ncv; j++)
            for (Index i = 0; i < m_ncv; i++)
                Q(i, j) = (i == j) ? Scalar(1) : Scalar(0);
        std::vector<double> work(static_cast<std::size_t>(2 * m));
        Index kk = m_ncv;  // subspace dimension after the shifts applied so far
        for (Index i = k; i < m_ncv; i++)
        {
            if (is_complex(m_ritz_val[i]) && i + 1 < m_ncv && is_conj(m_ritz_val[i], m_ritz_val[i + 1]))
            {
                // (H - mu I)(H - conj(mu) I) = H^2 - 2 Re(mu) H + |mu|^2 I, real arithmetic throughout
                const RealScalar s = RealScalar(2) * m_ritz_val[i].real();
                const RealScalar t = std::norm(m_ritz_val[i]);
                mispec::small::DoubleShiftStep(m, H.data(), m, s, t, Q.data(), m, m);
                kk -= 2;
                i++;
            }
            else
            {
                mispec::small::hess_shifted_qr(m, H.data(), m, m_ritz_val[i].real(), Q.data(), m, m, work.data());
                kk -= 1;
            }
        }
        m_fac.compress_V(Q, H, kk);                // V <- V Q and the new residual, on the device
        m_fac.factorize_from(k, m_ncv, m_nmatop);  // back to an ncv-step factorisation
        retrieve_ritzpair(selection);
    }

protected:
    virtual void sort_ritzpair(SortRule sort_rule)  // reference :345-401
    {
        std::vector<Index> ind;
        try
        {
            ind = argsort(sort_rule, m_ritz_val.data(), m_nev);
        }
        catch (const std::invalid_argument&)
        {
            throw std::invalid_argument("unsupported sorting rule");
        }
        ComplexVector new_val(m_ncv);
        ComplexMatrix new_vec(m_ncv, m_nev);
        std::vector<char> new_conv(static_cast<std::size_t>(m_nev), 0);
        for (Index i = 0; i < m_ncv; i++)
            new_val[i] = Complex(0, 0);
        for (Index i = 0; i < m_nev; i++)
        {
            new_val[i] = m_ritz_val[ind[i]];
            for (Index r = 0; r < m_ncv; r++)
                new_vec(r, i) = m_ritz_vec(r, ind[i]);
            new_conv[static_cast<std::size_t>(i)] = m_ritz_conv[static_cast<std::size_t>(ind[i])];
        }
        m_ritz_val = new_val;
        m_ritz_vec = new_vec;
        m_ritz_conv.swap(new_conv);
    }

    Index num_flagged() const
    {
        Index c = 0;
        for (char b : m_ritz_conv)
            c += b;
        return c;
    }

public:
    GenEigsBase(OpType& op, const BOpType& /*Bop*/, Index nev, Index ncv) :
        m_op(op),
        m_n(op.rows()),
        m_nev(nev),
        m_ncv(check_args(op.rows(), nev, ncv)),
        m_nmatop(0),
        m_niter(0),
        m_fac(op, m_ncv),
        m_info(CompInfo::NotComputed)
    {}
    virtual ~GenEigsBase() {}

    void init(const Scalar* init_resid)
    {
        reset();
        m_fac.init(init_resid, m_nmatop);
    }
    void init()  // SimpleRandom(0), generated on the device (reference :471-476)
    {
        reset();
        m_fac.init_random(0, m_nmatop);
    }

    Index compute(SortRule selection = SortRule::LargestMagn, Index maxit = 1000, RealScalar tol = 1e-10,
                  SortRule sorting = SortRule::LargestMagn)
    {
        m_fac.factorize_from(1, m_ncv, m_nmatop);
        retrieve_ritzpair(selection);
        Index i, nconv = 0;
        for (i = 0; i < maxit; i++)
        {
            nconv = num_converged(tol);
            if (nconv >= m_nev)
                break;
            restart(nev_adjusted(nconv), selection);
        }
        sort_ritzpair(sorting);
        m_niter += i + 1;
        m_info = (nconv >= m_nev) ? CompInfo::Successful : CompInfo::NotConverging;
        return (std::min)(m_nev, nconv);
    }

    CompInfo info() const { return m_info; }
    Index num_iterations() const { return m_niter; }
    Index num_operations() const { return m_nmatop; }

    ComplexVector eigenvalues() const
    {
        ComplexVector res(num_flagged());
        Index j = 0;
        for (Index i = 0; i < m_nev; i++)
            if (m_ritz_conv[static_cast<std::size_t>(i)])
                res[j++] = m_ritz_val[i];
        return res;
    }

    // n x nvec complex: V * Y with Y complex = two real products on the device (reference :578-602)
    ComplexMatrix eigenvectors(Index nvec) const
    {
        nvec = (std::min)(nvec, num_flagged());
        Matrix Yre(m_ncv, nvec), Yim(m_ncv, nvec);
        Index j = 0;
        for (Index i = 0; i < m_nev && j < nvec; i++)
        {
            if (!m_ritz_conv[static_cast<std::size_t>(i)])
                continue;
            for (Index r = 0; r < m_ncv; r++)
            {
                Yre(r, j) = m_ritz_vec(r, i).real();
                Yim(r, j) = m_ritz_vec(r, i).imag();
            }
            j++;
        }
        const Matrix Xre = m_fac.ritz_vectors(Yre);
        const Matrix Xim = m_fac.ritz_vectors(Yim);
        ComplexMatrix res(Xre.rows(), nvec);
        for (Index c = 0; c < nvec; c++)
            for (Index r = 0; r < Xre.rows(); r++)
                res(r, c) = Complex(Xre(r, c), Xim(r, c));
        return res;
    }
    ComplexMatrix eigenvectors() const { return eigenvectors(m_nev); }

    const ArnoldiFac& factorization() const { return m_fac; }

    // Device-side extra (no counterpart in the reference): ||A x - lambda x||_2 / ||x||_2 of the converged
    // pairs, formed and reduced on the GPU without bringing the eigenvectors to the host.
    DenseVector<RealScalar> residuals() const
    {
        const Index nconv = num_flagged();
        DenseVector<RealScalar> res(nconv);
        if (nconv == 0)
            return res;
        Matrix Yre(m_ncv, nconv), Yim(m_ncv, nconv);
        std::vector<RealScalar> lam(static_cast<std::size_t>(2 * nconv));
        Index j = 0;
        for (Index i = 0; i < m_nev; i++)
        {
            if (!m_ritz_conv[static_cast<std::size_t>(i)])
                continue;
            for (Index r = 0; r < m_ncv; r++)
            {
                Yre(r, j) = m_ritz_vec(r, i).real();
                Yim(r, j) = m_ritz_vec(r, i).imag();
            }
            lam[static_cast<std::size_t>(2 * j)] = m_ritz_val[i].real();
            lam[static_cast<std::size_t>(2 * j + 1)] = m_ritz_val[i].imag();
            j++;
        }
        internal::check(mispec_fac_residuals_complex(m_fac.handle(), Yre.data(), Yim.data(), lam.data(), static_cast<int>(nconv),
                                                     res.data()));
        return res;
    }

private:
    void reset()
    {
        m_ritz_val.resize(m_ncv);
        m_ritz_vec.resize(m_ncv, m_nev);
        m_ritz_est.resize(m_ncv);
        m_ritz_conv.assign(static_cast<std::size_t>(m_nev), 0);
        for (Index i = 0; i < m_ncv; i++)
        {
            m_ritz_val[i] = Complex(0, 0);
            m_ritz_est[i] = Complex(0, 0);
        }
        for (Index c = 0; c < m_nev; c++)
            for (Index r = 0; r < m_ncv; r++)
                m_ritz_vec(r, c) = Complex(0, 0);
        m_nmatop = 0;
        m_niter = 0;
    }
};

}  // namespace Spectra

#endif
