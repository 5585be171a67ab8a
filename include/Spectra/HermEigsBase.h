// Implicitly-restarted Lanczos driver for real symmetric problems, GPU-resident.
//
// Public surface and semantics follow the reference's HermEigsBase (HermEigsBase.h:44-478, which
// SymEigsSolver and SymEigsShiftSolver derive from): constructor checks (:257-272), init() / init(v0)
// (:309-342), compute(selection, maxit, tol, sorting) (:366-390), info(), num_iterations(),
// num_operations(), eigenvalues(), eigenvectors([nvec]) (:395-478) and the virtual sort_ritzpair hook
// (:229-251).  The restart loop below is the same algorithm (Ritz pairs -> convergence test -> ARPACK
// nev adjustment -> exact shifts, largest magnitude first -> compress -> re-factorise); what is
// different is that every length-n operation and the m x m QR sweeps run in HIP kernels and the host
// only ever sees m-sized data (Ritz values, last row of the Ritz vectors, H).
#ifndef MISPEC_SPECTRA_HERM_EIGS_BASE_H
#define MISPEC_SPECTRA_HERM_EIGS_BASE_H

#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <utility>
#include <vector>

#include "LinAlg/Lanczos.h"
#include "Util/CompInfo.h"
#include "Util/SelectionRule.h"
#include "Util/SimpleRandom.h"
#include "Util/TypeTraits.h"
#include "internal/Dense.h"

namespace Spectra {

// IdentityBOp, the placeholder for B = I in A x = lambda B x, comes from MatOp/internal/ArnoldiOp.h (through LinAlg/Arnoldi.h), where
// the reference declares it (MatOp/internal/ArnoldiOp.h:105-106).

template <typename OpType, typename BOpType = IdentityBOp>
class HermEigsBase
{
    // BOpType other than IdentityBOp: the generalized problem in regular-inverse mode — OpType is then a
    // SymGEigsRegInvOp, whose device hooks tell the factorisation to use the B-inner product
    // (reference: ArnoldiOp<Scalar, OpType, BOpType>, MatOp/internal/ArnoldiOp.h:36-101).

private:
    using Scalar = typename OpType::Scalar;
    using RealScalar = ElemType<Scalar>;
    using Matrix = DenseMatrix<Scalar>;
    using Vector = DenseVector<Scalar>;
    using RealMatrix = DenseMatrix<RealScalar>;
    using RealVector = DenseVector<RealScalar>;
    using LanczosFac = Lanczos<OpType>;

protected:
    // An operator passed as an rvalue is moved into this container and m_op refers to it (reference :73-79)
    std::vector<OpType> m_op_container;
    const OpType& m_op;   // matrix operator for A
    const Index m_n;      // dimension of A
    const Index m_nev;    // number of eigenvalues requested
    const Index m_ncv;    // dimension of the Krylov subspace
    Index m_nmatop;       // number of operator applications
    Index m_niter;        // number of restarts
    LanczosFac m_fac;     // device-resident factorisation
    RealVector m_ritz_val;  // Ritz values, wanted ones first

private:
    RealMatrix m_ritz_vec;          // Ritz vectors of H for the nev wanted values (ncv x nev)
    RealVector m_ritz_est;          // last row of the eigenvector matrix of H
    std::vector<char> m_ritz_conv;  // convergence flags of the wanted values
    bool m_ritz_vec_current = false;  // m_ritz_vec belongs to the current H (it is formed after the last iteration only)
    CompInfo m_info;

    // LIMIT of this implementation (not of the reference): the device factorisation holds at most 1024 basis vectors, so a
    // solver constructed with ncv > 1024 throws std::invalid_argument from the factorisation's constructor
    // (mispec_fac_create: "ncv <= 1024"); the reference accepts any nev < ncv <= n.
    static Index check_ncv(Index ncv, Index n) { return ncv > n ? n : ncv; }
    static std::vector<OpType> create_op_container(OpType&& rval)
    {
        std::vector<OpType> container;
        container.emplace_back(std::move(rval));
        return container;
    }

    // Ritz pairs of H, wanted ones first (reference :205-224)
    // with_vectors == false (the iterations of compute()): the Ritz values and the last components of their vectors — all that
    // the convergence test and the choice of the shifts need; the m x nev matrix of Ritz vectors is formed once, after the last
    // iteration, from the same H (same values, same order: the eigen-decomposition is deterministic).
    void retrieve_ritzpair(SortRule selection, bool with_vectors = true)
    {
        RealVector evals;
        if (!with_vectors)
        {
            RealVector last_row;
            m_fac.ritz_values(evals, last_row);
            const std::vector<Index> ind = argsort(selection, evals.data(), m_ncv);
            for (Index i = 0; i < m_ncv; i++)
            {
                m_ritz_val[i] = evals[ind[i]];
                m_ritz_est[i] = last_row[ind[i]];
            }
            m_ritz_vec_current = false;
            return;
        }
        RealMatrix evecs;
        m_fac.ritz_pairs(evals, evecs);
        const std::vector<Index> ind = argsort(selection, evals.data(), m_ncv);
        for (Index i = 0; i < m_ncv; i++)
        {
            m_ritz_val[i] = evals[ind[i]];
            m_ritz_est[i] = evecs(m_ncv - 1, ind[i]);
        }
        for (Index i = 0; i < m_nev; i++)
            for (Index r = 0; r < m_ncv; r++)
                m_ritz_vec(r, i) = evecs(r, ind[i]);
        m_ritz_vec_current = true;
    }

    // |last component| * |f| < tol * max(eps^(2/3), |theta|)  (reference :158-175)
    Index num_converged(RealScalar tol)
    {
        const RealScalar eps23 = std::pow(TypeTraits<RealScalar>::epsilon(), RealScalar(2) / 3);
        const RealScalar fnorm = m_fac.f_norm();
        Index count = 0;
        for (Index i = 0; i < m_nev; i++)
        {
            const RealScalar thresh = tol * (std::max)(eps23, std::abs(m_ritz_val[i]));
            const RealScalar resid = std::abs(m_ritz_est[i]) * fnorm;
            m_ritz_conv[i] = (resid < thresh) ? 1 : 0;
            count += m_ritz_conv[i];
        }
        return count;
    }

    // How many Ritz pairs to keep at this restart: ARPACK's dsaup2 heuristic (reference :178-202)
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
        else if (nev_new == 1 && m_ncv > 2)
            nev_new = 2;
        if (nev_new > m_ncv - 1)
            nev_new = m_ncv - 1;
        return nev_new;
    }

    // One implicit restart keeping k Ritz pairs (reference :105-155)
    void restart(Index k, SortRule selection)
    {
        if (k >= m_ncv)
            return;
        const Index nshift = m_ncv - k;
        // the unwanted Ritz values are the shifts; large magnitudes first
        std::vector<RealScalar> shifts(static_cast<std::size_t>(nshift));
        for (Index i = 0; i < nshift; i++)
            shifts[static_cast<std::size_t>(i)] = m_ritz_val[k + i];
        std::sort(shifts.begin(), shifts.end(), [](const RealScalar& a, const RealScalar& b) { return std::abs(a) > std::abs(b); });
        // shifted QR sweeps on H, Q accumulation, V <- VQ and the new residual: all on the device
        m_fac.restart_with_shifts(shifts.data(), nshift);
        // back to an ncv-step factorisation
        m_fac.factorize_from(k, m_ncv, m_nmatop);
        retrieve_ritzpair(selection, false);
    }

protected:
    // Final ordering of the nev wanted pairs (reference :229-251); shift-and-invert solvers override
    // this to map the Ritz values back first.
    virtual void sort_ritzpair(SortRule sort_rule)
    {
        if (sort_rule != SortRule::LargestAlge && sort_rule != SortRule::LargestMagn && sort_rule != SortRule::SmallestAlge &&
            sort_rule != SortRule::SmallestMagn)
            throw std::invalid_argument("unsupported sorting rule");
        const std::vector<Index> ind = argsort(sort_rule, m_ritz_val.data(), m_nev);
        RealVector new_val(m_ncv);
        RealMatrix new_vec(m_ncv, m_nev);
        std::vector<char> new_conv(static_cast<std::size_t>(m_nev), 0);
        for (Index i = 0; i < m_ncv; i++)
            new_val[i] = RealScalar(0);
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
    HermEigsBase(OpType& op, const BOpType& /*Bop*/, Index nev, Index ncv) :
        m_op(op),
        m_n(op.rows()),
        m_nev(nev),
        m_ncv(check_args(op.rows(), nev, ncv)),
        m_nmatop(0),
        m_niter(0),
        m_fac(op, m_ncv),
        m_info(CompInfo::NotComputed)
    {}

    // Same for an operator object the solver is to own (the generalized solvers build theirs on the fly; reference :275-288)
    HermEigsBase(OpType&& op, const BOpType& /*Bop*/, Index nev, Index ncv) :
        m_op_container(create_op_container(std::move(op))),
        m_op(m_op_container.front()),
        m_n(m_op.rows()),
        m_nev(nev),
        m_ncv(check_args(m_op.rows(), nev, ncv)),
        m_nmatop(0),
        m_niter(0),
        m_fac(m_op, m_ncv),
        m_info(CompInfo::NotComputed)
    {}

    virtual ~HermEigsBase() {}

    // Start from a user-supplied residual vector (n entries, host memory).
    void init(const Scalar* init_resid)
    {
        reset();
        m_fac.init(init_resid, m_nmatop);
    }

    // Start from the reference's default vector: SimpleRandom(0), i.i.d. U(-0.5, 0.5), generated on the device.
    void init()
    {
        reset();
        m_fac.init_random(0, m_nmatop);
    }

    // selection: which end of the spectrum; maxit: restart limit; tol: relative residual tolerance;
    // sorting: order of the returned pairs.  Returns the number of converged eigenvalues.
    Index compute(SortRule selection = SortRule::LargestMagn, Index maxit = 1000, RealScalar tol = 1e-10,
                  SortRule sorting = SortRule::LargestAlge)
    {
        m_fac.factorize_from(1, m_ncv, m_nmatop);
        retrieve_ritzpair(selection, false);
        Index i, nconv = 0;
        for (i = 0; i < maxit; i++)
        {
            nconv = num_converged(tol);
            if (nconv >= m_nev)
                break;
            restart(nev_adjusted(nconv), selection);
        }
        if (!m_ritz_vec_current)
        {
            // The Ritz vectors of the final H (what eigenvectors() multiplies V by): a second decomposition of the SAME host copy
            // of H the values-only call above read — nothing between num_converged() and this line may change it (f_norm() only
            // reads beta; accessors that resolve pending device state, get_H / get_f, are not called here).  m_ritz_conv comes
            // from the first call, the vectors from this one: the order must agree, i.e. the values must be the same bits.
#ifndef NDEBUG
            const RealVector before = m_ritz_val;
#endif
            retrieve_ritzpair(selection, true);
#ifndef NDEBUG
            for (Index j = 0; j < m_nev; j++)
                if (!(before[j] == m_ritz_val[j]))
                    throw std::logic_error("HermEigsBase: H changed between the convergence test and the Ritz vectors");
#endif
        }
        sort_ritzpair(sorting);
        m_niter += i + 1;
        m_info = (nconv >= m_nev) ? CompInfo::Successful : CompInfo::NotConverging;
        return (std::min)(m_nev, nconv);
    }

    CompInfo info() const { return m_info; }
    Index num_iterations() const { return m_niter; }
    Index num_operations() const { return m_nmatop; }

    // Converged eigenvalues, in the order requested by `sorting`.
    RealVector eigenvalues() const
    {
        const Index nconv = num_flagged();
        RealVector res(nconv);
        Index j = 0;
        for (Index i = 0; i < m_nev; i++)
            if (m_ritz_conv[static_cast<std::size_t>(i)])
                res[j++] = m_ritz_val[i];
        return res;
    }

    // Eigenvectors of the converged eigenvalues: V * (Ritz vectors of H), formed on the device and
    // copied to the host (n x nvec).  On a row-sharded run each rank gets its own rows.
    virtual Matrix eigenvectors(Index nvec) const { return m_fac.ritz_vectors(converged_ritz_vectors(nvec)); }
    virtual Matrix eigenvectors() const { return eigenvectors(m_nev); }

    // Device-side extra (no counterpart in the reference, whose eigenvectors() returns an n x nev host
    // matrix — 1.6 GB at n = 1e7): form the same eigenvectors but leave them in HBM.  Returns the number
    // of columns; *dev / *ld receive the device pointer and leading dimension (valid until the next call).
    Index eigenvectors_on_device(Index nvec, const Scalar** dev = nullptr, Index* ld = nullptr) const
    {
        const RealMatrix Y = converged_ritz_vectors(nvec);
        const Scalar* X = m_fac.ritz_vectors_device(Y, ld);
        if (dev)
            *dev = X;
        return Y.cols();
    }

    // The eigenvectors written straight into caller-provided host memory (n x min(nvec, nconv), column-major): what
    // eigenvectors() returns, without the intermediate matrix (1.6 GB at n = 1e7).  Returns the number of columns.
    virtual Index eigenvectors_to(Scalar* out_host, Index nvec) const
    {
        const RealMatrix Y = converged_ritz_vectors(nvec);
        m_fac.ritz_vectors_into(Y, out_host);
        return Y.cols();
    }

    // Device-side extras (no counterpart in the reference): the factorisation handle, e.g. for
    // mispec_fac_residuals() / profiling through the C ABI.
    const LanczosFac& factorization() const { return m_fac; }

    // Device-side extra (no counterpart in the reference): the one-sweep orthogonalisation of the Lanczos steps
    // (include/mispec.h mispec_fac_set_orth_mode) is the default; false selects the reference's two-pass control flow
    // (also: MISPEC_ORTH=reference in the environment).  Call before init().
    void set_onesweep_orthogonalization(bool on)
    {
        internal::check(mispec_fac_set_orth_mode(m_fac.handle(), on ? MISPEC_ORTH_ONESWEEP : MISPEC_ORTH_REFERENCE));
    }
    // ... with ONE global reduction per step (the default; include/mispec.h MISPEC_ORTH_ONE_REDUCTION) or with the separate
    // reduction of alpha = <v, w> before the pass over V (also: MISPEC_ONE_REDUCTION=0 in the environment).  Call before init().
    void set_onesweep_orthogonalization(bool on, bool one_reduction)
    {
        const int flags = one_reduction ? MISPEC_ORTH_ONE_REDUCTION : MISPEC_ORTH_TWO_REDUCTIONS;
        internal::check(mispec_fac_set_orth_mode(m_fac.handle(), on ? (MISPEC_ORTH_ONESWEEP | flags) : MISPEC_ORTH_REFERENCE));
    }

private:
    // The Ritz vectors of H belonging to the converged wanted values, as columns (reference :455-465)
    RealMatrix converged_ritz_vectors(Index nvec) const
    {
        const Index nconv = num_flagged();
        nvec = (std::min)(nvec, nconv);
        RealMatrix Y(m_ncv, nvec);
        Index j = 0;
        for (Index i = 0; i < m_nev && j < nvec; i++)
        {
            if (!m_ritz_conv[static_cast<std::size_t>(i)])
                continue;
            for (Index r = 0; r < m_ncv; r++)
                Y(r, j) = m_ritz_vec(r, i);
            j++;
        }
        return Y;
    }

    static Index check_args(Index n, Index nev, Index ncv)
    {
        if (nev < 1 || nev > n - 1)
            throw std::invalid_argument("nev must satisfy 1 <= nev <= n - 1, n is the size of matrix");
        if (ncv <= nev || ncv > n)
            throw std::invalid_argument("ncv must satisfy nev < ncv <= n, n is the size of matrix");
        return check_ncv(ncv, n);
    }

    void reset()
    {
        m_ritz_val.resize(m_ncv);
        m_ritz_vec.resize(m_ncv, m_nev);
        m_ritz_est.resize(m_ncv);
        m_ritz_conv.assign(static_cast<std::size_t>(m_nev), 0);
        for (Index i = 0; i < m_ncv; i++)
        {
            m_ritz_val[i] = RealScalar(0);
            m_ritz_est[i] = RealScalar(0);
        }
        for (Index c = 0; c < m_nev; c++)
            for (Index r = 0; r < m_ncv; r++)
                m_ritz_vec(r, c) = RealScalar(0);
        m_nmatop = 0;
        m_niter = 0;
    }
};

}  // namespace Spectra

#endif
