// Block Davidson eigen solver for real symmetric matrices with the diagonal-preconditioned-residual (DPR)
// correction.  Same constructor signatures, member names and defaults as the reference class
// (DavidsonSymEigsSolver.h:18-90 with its base JDSymEigsBase.h:28-187): compute(), compute_with_guess(),
// eigenvalues(), eigenvectors(), info(), num_iterations(), set_max_search_space_size(), set_correction_size(),
// set_initial_search_space_size().
//
// The search space, its image under A and the Ritz vectors live in HBM; this class is a thin owner of a
// mispec_davidson handle (csrc/davidson.hip).  Operators: the device matrix classes (SparseSymMatProd,
// DenseSymMatProd) or a user class with perform_op_device() plus operator()(i, i) for the diagonal.
// Differences a user can observe: the device search space holds 256 vectors (a larger nvec_max is lowered to
// 256 - correction size: earlier restarts, same results); new directions are orthonormalised by two
// Gram-Schmidt passes per vector instead of a block projection + Householder QR (same space, vectors equal up to
// sign); the base class JDSymEigsBase is not exposed as a customisation point.
#ifndef MISPEC_SPECTRA_DAVIDSON_SYM_EIGS_SOLVER_H
#define MISPEC_SPECTRA_DAVIDSON_SYM_EIGS_SOLVER_H

#include "../mispec_extras.h"  // outside the hot path of SURVEY.md section 8: declared apart from the thin shim
#include <memory>
#include <stdexcept>
#include <type_traits>
#include <vector>

#include "LinAlg/Arnoldi.h"  // operator detection traits
#include "Util/CompInfo.h"
#include "Util/SelectionRule.h"
#include "internal/Dense.h"
#include "internal/Device.h"

namespace Spectra {

template <typename OpType>
class DavidsonSymEigsSolver
{
public:
    using Scalar = typename OpType::Scalar;

private:
    static_assert(std::is_same<Scalar, double>::value, "the MI355X path computes in fp64: Scalar must be double");
    using Matrix = DenseMatrix<Scalar>;
    using Vector = DenseVector<Scalar>;

    const OpType& m_matrix_operator;
    const Index m_number_eigenvalues;
    internal::CtxPtr m_ctx;
    std::shared_ptr<mispec_davidson> m_solver;

    static int call_user_device_op(void* user, const double* x_dev, double* y_dev, void* hip_stream)
    {
        try
        {
            static_cast<const OpType*>(user)->perform_op_device(x_dev, y_dev, hip_stream);
            return 0;
        }
        catch (...)
        {
            return 1;
        }
    }

    void adopt(mispec_davidson* raw) { m_solver = std::shared_ptr<mispec_davidson>(raw, [](mispec_davidson* p) { (void) mispec_davidson_destroy(p); }); }

    template <typename T = OpType>
    typename std::enable_if<internal::has_device_matrix<T>::value>::type bind(Index nvec_init, Index nvec_max)
    {
        m_ctx = internal::borrow_context(m_matrix_operator.mispec_context());
        mispec_davidson* raw = nullptr;
        internal::check(mispec_davidson_create(m_ctx.get(), m_matrix_operator.mispec_matrix(), m_number_eigenvalues, nvec_init, nvec_max, &raw));
        adopt(raw);
    }
    template <typename T = OpType>
    typename std::enable_if<internal::has_device_dense<T>::value>::type bind(Index nvec_init, Index nvec_max)
    {
        m_ctx = internal::borrow_context(m_matrix_operator.mispec_context());
        mispec_davidson* raw = nullptr;
        internal::check(mispec_davidson_create_dense(m_ctx.get(), m_matrix_operator.mispec_dense_matrix(), m_number_eigenvalues, nvec_init,
                                                     nvec_max, &raw));
        adopt(raw);
    }
    template <typename T = OpType>
    typename std::enable_if<internal::has_device_perform_op<T>::value && !internal::has_device_matrix<T>::value &&
                            !internal::has_device_dense<T>::value>::type
    bind(Index nvec_init, Index nvec_max)
    {
        m_ctx = internal::context_of(m_matrix_operator);
        const Index n = m_matrix_operator.rows();
        std::vector<double> diag(static_cast<std::size_t>(n));
        for (Index i = 0; i < n; i++)
            diag[static_cast<std::size_t>(i)] = m_matrix_operator(i, i);  // DavidsonSymEigsSolver.h:33-38
        mispec_davidson* raw = nullptr;
        internal::check(mispec_davidson_create_device_op(m_ctx.get(), &DavidsonSymEigsSolver::call_user_device_op,
                                                         const_cast<OpType*>(&m_matrix_operator), n, diag.data(), m_number_eigenvalues,
                                                         nvec_init, nvec_max, &raw));
        adopt(raw);
    }

    Index run(SortRule selection, Index maxit, Scalar tol, const Scalar* guess, Index guess_cols, Index ldg)
    {
        int64_t nconv = 0;
        internal::check(mispec_davidson_compute(m_solver.get(), static_cast<int>(selection), maxit, tol, guess, guess_cols, ldg, &nconv));
        return static_cast<Index>(nconv);
    }

public:
    DavidsonSymEigsSolver(OpType& op, Index nev, Index nvec_init, Index nvec_max) : m_matrix_operator(op), m_number_eigenvalues(nev)
    {
        bind(nvec_init, nvec_max);
    }
    DavidsonSymEigsSolver(OpType& op, Index nev) : DavidsonSymEigsSolver(op, nev, 2 * nev, 10 * nev) {}

    void set_max_search_space_size(Index max_search_space_size) { internal::check(mispec_davidson_set_sizes(m_solver.get(), -1, max_search_space_size, -1)); }
    void set_correction_size(Index correction_size) { internal::check(mispec_davidson_set_sizes(m_solver.get(), -1, -1, correction_size)); }
    void set_initial_search_space_size(Index initial_search_space_size)
    {
        internal::check(mispec_davidson_set_sizes(m_solver.get(), initial_search_space_size, -1, -1));
    }

    CompInfo info() const { return static_cast<CompInfo>(mispec_davidson_info(m_solver.get())); }
    Index num_iterations() const { return static_cast<Index>(mispec_davidson_num_iterations(m_solver.get())); }

    Vector eigenvalues() const
    {
        Vector res(m_number_eigenvalues);
        internal::check(mispec_davidson_eigenvalues(m_solver.get(), res.data()));
        return res;
    }
    Matrix eigenvectors() const
    {
        Matrix res(m_matrix_operator.rows(), m_number_eigenvalues);
        internal::check(mispec_davidson_eigenvectors(m_solver.get(), res.data(), res.rows()));
        return res;
    }

    // tol default: 100 * Eigen::NumTraits<double>::dummy_precision() = 1e-10 (JDSymEigsBase.h:121-122)
    Index compute(SortRule selection = SortRule::LargestMagn, Index maxit = 100, Scalar tol = 1e-10)
    {
        return run(selection, maxit, tol, nullptr, 0, 0);
    }
    // initial_space: n x p column-major block in host memory (the reference takes an Eigen::Ref<const Matrix>)
    Index compute_with_guess(const Matrix& initial_space, SortRule selection = SortRule::LargestMagn, Index maxit = 100, Scalar tol = 1e-10)
    {
        if (initial_space.rows() != m_matrix_operator.rows())
            throw std::invalid_argument("compute_with_guess: the initial space must have n rows");
        return run(selection, maxit, tol, initial_space.data(), initial_space.cols(), initial_space.rows());
    }
};

}  // namespace Spectra

#endif
