// Orthonormalisation routines for the columns of a dense host matrix — the function set of the reference's
// LinAlg/Orthogonalization.h:20-137 (used by its Jacobi-Davidson search space): MGS / GS / QR / subspace projection /
// Jens Wehner's projection + QR / "twice is enough".  `left_cols_to_skip` leading columns are taken as already
// orthonormal and left untouched.  Host code for user programs; the device Davidson solver orthogonalises with the
// factorisation kernels instead (csrc/davidson.hip).  Matrix: any type with rows(), cols() and operator()(i, j).
#ifndef MISPEC_SPECTRA_ORTHOGONALIZATION_H
#define MISPEC_SPECTRA_ORTHOGONALIZATION_H

#include <cassert>
#include <cmath>
#include <vector>

#include "../internal/Dense.h"

namespace Spectra {

template <typename Matrix>
void assert_left_cols_to_skip(Matrix& in_output, Index left_cols_to_skip)
{
    assert(static_cast<Index>(in_output.cols()) > left_cols_to_skip && "left_cols_to_skip is larger than columns of matrix");
    assert(left_cols_to_skip >= 0 && "left_cols_to_skip is negative");
    (void) in_output;
    (void) left_cols_to_skip;
}

namespace internal {
template <typename Matrix>
double col_dot(const Matrix& M, Index a, Index b)
{
    double s = 0.0;
    for (Index i = 0; i < static_cast<Index>(M.rows()); i++)
        s += M(i, a) * M(i, b);
    return s;
}
template <typename Matrix>
void col_normalize(Matrix& M, Index j)
{
    const double nrm = std::sqrt(col_dot(M, j, j));
    if (nrm > 0.0)
        for (Index i = 0; i < static_cast<Index>(M.rows()); i++)
            M(i, j) /= nrm;
}
}  // namespace internal

// If nothing is to be skipped: normalise the first column and skip it (reference :32-41)
template <typename Matrix>
Index treat_first_col(Matrix& in_output, Index left_cols_to_skip)
{
    if (left_cols_to_skip == 0)
    {
        internal::col_normalize(in_output, 0);
        left_cols_to_skip = 1;
    }
    return left_cols_to_skip;
}

// Columns [first, cols) <- the thin Q of their Householder QR (reference :45-57 for the whole matrix)
template <typename Matrix>
void QR_orthogonalisation(Matrix& in_output, Index first = 0)
{
    const Index n = static_cast<Index>(in_output.rows());
    const Index p = static_cast<Index>(in_output.cols()) - first;
    const Index r = p < n ? p : n;  // number of reflectors
    std::vector<std::vector<double>> vs(static_cast<std::size_t>(r));
    for (Index k = 0; k < r; k++)
    {
        std::vector<double>& v = vs[static_cast<std::size_t>(k)];
        v.assign(static_cast<std::size_t>(n), 0.0);
        double nrm2 = 0.0;
        for (Index i = k; i < n; i++)
        {
            v[static_cast<std::size_t>(i)] = in_output(i, first + k);
            nrm2 += v[static_cast<std::size_t>(i)] * v[static_cast<std::size_t>(i)];
        }
        const double nrm = std::sqrt(nrm2);
        if (nrm == 0.0)
            continue;  // H = I
        v[static_cast<std::size_t>(k)] += (v[static_cast<std::size_t>(k)] >= 0.0) ? nrm : -nrm;
        double vn2 = 0.0;
        for (Index i = k; i < n; i++)
            vn2 += v[static_cast<std::size_t>(i)] * v[static_cast<std::size_t>(i)];
        const double vn = std::sqrt(vn2);
        for (Index i = k; i < n; i++)
            v[static_cast<std::size_t>(i)] /= vn;
        for (Index j = k; j < p; j++)  // apply H = I - 2 v v' to the remaining columns
        {
            double s = 0.0;
            for (Index i = k; i < n; i++)
                s += v[static_cast<std::size_t>(i)] * in_output(i, first + j);
            for (Index i = k; i < n; i++)
                in_output(i, first + j) -= 2.0 * s * v[static_cast<std::size_t>(i)];
        }
    }
    // Q = H_0 H_1 ... H_{r-1} applied to the first p columns of the identity
    for (Index j = 0; j < p; j++)
        for (Index i = 0; i < n; i++)
            in_output(i, first + j) = (i == j) ? 1.0 : 0.0;
    for (Index k = r - 1; k >= 0; k--)
    {
        const std::vector<double>& v = vs[static_cast<std::size_t>(k)];
        if (v.empty())
            continue;
        bool zero = true;
        for (Index i = k; i < n && zero; i++)
            zero = (v[static_cast<std::size_t>(i)] == 0.0);
        if (zero)
            continue;
        for (Index j = 0; j < p; j++)
        {
            double s = 0.0;
            for (Index i = k; i < n; i++)
                s += v[static_cast<std::size_t>(i)] * in_output(i, first + j);
            for (Index i = k; i < n; i++)
                in_output(i, first + j) -= 2.0 * s * v[static_cast<std::size_t>(i)];
        }
    }
}

// Modified Gram-Schmidt (reference :60-76)
template <typename Matrix>
void MGS_orthogonalisation(Matrix& in_output, Index left_cols_to_skip = 0)
{
    assert_left_cols_to_skip(in_output, left_cols_to_skip);
    left_cols_to_skip = treat_first_col(in_output, left_cols_to_skip);
    const Index n = static_cast<Index>(in_output.rows());
    for (Index k = left_cols_to_skip; k < static_cast<Index>(in_output.cols()); ++k)
    {
        for (Index j = 0; j < k; j++)
        {
            const double c = internal::col_dot(in_output, j, k);
            for (Index i = 0; i < n; i++)
                in_output(i, k) -= c * in_output(i, j);
        }
        internal::col_normalize(in_output, k);
    }
}

// Classical Gram-Schmidt: all coefficients of a column from the unmodified column (reference :79-95)
template <typename Matrix>
void GS_orthogonalisation(Matrix& in_output, Index left_cols_to_skip = 0)
{
    assert_left_cols_to_skip(in_output, left_cols_to_skip);
    left_cols_to_skip = treat_first_col(in_output, left_cols_to_skip);
    const Index n = static_cast<Index>(in_output.rows());
    std::vector<double> c;
    for (Index j = left_cols_to_skip; j < static_cast<Index>(in_output.cols()); ++j)
    {
        c.assign(static_cast<std::size_t>(j), 0.0);
        for (Index l = 0; l < j; l++)
            c[static_cast<std::size_t>(l)] = internal::col_dot(in_output, l, j);
        for (Index l = 0; l < j; l++)
            for (Index i = 0; i < n; i++)
                in_output(i, j) -= c[static_cast<std::size_t>(l)] * in_output(i, l);
        internal::col_normalize(in_output, j);
    }
}

// Right block -= Left (Left' Right): the new columns are made orthogonal to the old space, not to each other (reference :98-115)
template <typename Matrix>
void subspace_orthogonalisation(Matrix& in_output, Index left_cols_to_skip)
{
    assert_left_cols_to_skip(in_output, left_cols_to_skip);
    if (left_cols_to_skip == 0)
        return;
    const Index n = static_cast<Index>(in_output.rows());
    std::vector<double> c(static_cast<std::size_t>(left_cols_to_skip));
    for (Index j = left_cols_to_skip; j < static_cast<Index>(in_output.cols()); ++j)
    {
        for (Index l = 0; l < left_cols_to_skip; l++)
            c[static_cast<std::size_t>(l)] = internal::col_dot(in_output, l, j);
        for (Index l = 0; l < left_cols_to_skip; l++)
            for (Index i = 0; i < n; i++)
                in_output(i, j) -= c[static_cast<std::size_t>(l)] * in_output(i, l);
    }
}

// J. Wehner's scheme: project on the complement of the old space, then QR of the new block (reference :118-129)
template <typename Matrix>
void JensWehner_orthogonalisation(Matrix& in_output, Index left_cols_to_skip = 0)
{
    assert_left_cols_to_skip(in_output, left_cols_to_skip);
    subspace_orthogonalisation(in_output, left_cols_to_skip);
    QR_orthogonalisation(in_output, left_cols_to_skip);
}

// ... applied twice: the second pass removes what rounding left of the old space (reference :132-137)
template <typename Matrix>
void twice_is_enough_orthogonalisation(Matrix& in_output, Index left_cols_to_skip = 0)
{
    JensWehner_orthogonalisation(in_output, left_cols_to_skip);
    JensWehner_orthogonalisation(in_output, left_cols_to_skip);
}

}  // namespace Spectra

#endif
