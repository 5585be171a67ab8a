// Eigen solver for general real matrices with a complex shift: eigenvalues of A closest to sigma = sigmar + i sigmai,
// by the implicitly-restarted Arnoldi method on the REAL operator x -> Re((A - sigma I)^{-1} x)
// (reference: GenEigsComplexShiftSolver.h:20-150).
//
//     SparseGenComplexShiftSolve<double> op(A);
//     GenEigsComplexShiftSolver<SparseGenComplexShiftSolve<double>> eigs(op, nev, ncv, sigmar, sigmai);
//     eigs.init();  eigs.compute(SortRule::LargestMagn);
//
// The iteration sees nu = (1/(lambda - sigma) + 1/(lambda - conj(sigma))) / 2, which has two pre-images
// lambda = sigmar + (1 +- sqrt(1 - 4 nu^2 sigmai^2)) / (2 nu); as in the reference the right one is found by applying the
// operator once more with a real probe shift r and comparing op(v) with v / (lambda - r) for both candidates.
#ifndef MISPEC_SPECTRA_GEN_EIGS_COMPLEX_SHIFT_SOLVER_H
#define MISPEC_SPECTRA_GEN_EIGS_COMPLEX_SHIFT_SOLVER_H

#include "../mispec_extras.h"  // outside the hot path of SURVEY.md section 8: declared apart from the thin shim
#include <cmath>
#include <complex>
#include <vector>

#include "GenEigsBase.h"
#include "MatOp/SparseGenComplexShiftSolve.h"
#include "Util/SimpleRandom.h"
#include "Util/TypeTraits.h"

namespace Spectra {

namespace internal {
// set_shift() has to run before the base class binds the operator's device factorisation
template <typename OpType, typename Scalar>
OpType& gen_with_complex_shift(OpType& op, const Scalar& sigmar, const Scalar& sigmai)
{
    op.set_shift(sigmar, sigmai);
    return op;
}
}  // namespace internal

template <typename OpType = SparseGenComplexShiftSolve<double>>
class GenEigsComplexShiftSolver : public GenEigsBase<OpType, IdentityBOp>
{
    using Scalar = typename OpType::Scalar;
    using Complex = std::complex<Scalar>;
    using Matrix = DenseMatrix<Scalar>;
    using Base = GenEigsBase<OpType, IdentityBOp>;
    using Base::m_fac;
    using Base::m_n;
    using Base::m_ncv;
    using Base::m_nev;
    using Base::m_op;
    using Base::m_ritz_val;
    using Base::m_ritz_vec;

    const Scalar m_sigmar;
    const Scalar m_sigmai;

    // reference :39-123
    void sort_ritzpair(SortRule sort_rule) override
    {
        // probe shift (reference :69-72): real, from the deterministic generator
        SimpleRandom<Scalar> rng(0);
        const Scalar shiftr = rng.random() * m_sigmar + rng.random();
        const Complex shift(shiftr, Scalar(0));
        m_op.set_shift(shiftr, Scalar(0));

        const Scalar eps = TypeTraits<Scalar>::epsilon();
        std::vector<Scalar> op_re(static_cast<std::size_t>(m_n)), op_im(static_cast<std::size_t>(m_n));
        Matrix Y(m_ncv, 2);
        for (Index i = 0; i < m_nev; i++)
        {
            // v = V y_i (real and imaginary part), op(v) at the probe shift
            for (Index r = 0; r < m_ncv; r++)
            {
                Y(r, 0) = m_ritz_vec(r, i).real();
                Y(r, 1) = m_ritz_vec(r, i).imag();
            }
            const Matrix X = m_fac.ritz_vectors(Y);
            m_op.perform_op(&X(0, 0), op_re.data());
            m_op.perform_op(&X(0, 1), op_im.data());

            // the two roots of the quadratic (reference :85-90)
            const Complex nu = m_ritz_val[i];
            const Complex part1 = m_sigmar + Scalar(0.5) / nu;
            const Complex part2 = Scalar(0.5) * std::sqrt(Scalar(1) - Scalar(4) * m_sigmai * m_sigmai * (nu * nu)) / nu;
            const Complex root1 = part1 + part2, root2 = part1 - part2;
            Scalar err1 = 0, err2 = 0;
            for (Index k = 0; k < m_n; k++)
            {
                const Complex v(X(k, 0), X(k, 1));
                const Complex opv(op_re[static_cast<std::size_t>(k)], op_im[static_cast<std::size_t>(k)]);
                err1 += std::norm(opv - v / (root1 - shift));
                err2 += std::norm(opv - v / (root2 - shift));
            }
            const Complex lambdaj = (err1 < err2) ? root1 : root2;
            m_ritz_val[i] = lambdaj;
            if (std::abs(lambdaj.imag()) > eps)  // the conjugate follows (reference :110-114)
            {
                if (i + 1 < m_ncv)
                    m_ritz_val[i + 1] = std::conj(lambdaj);
                i++;
            }
            else
                m_ritz_val[i] = Complex(lambdaj.real(), Scalar(0));
        }
        Base::sort_ritzpair(sort_rule);
    }

public:
    GenEigsComplexShiftSolver(OpType& op, Index nev, Index ncv, const Scalar& sigmar, const Scalar& sigmai) :
        Base(internal::gen_with_complex_shift(op, sigmar, sigmai), IdentityBOp(), nev, ncv), m_sigmar(sigmar), m_sigmai(sigmai)
    {}
};

}  // namespace Spectra

#endif
