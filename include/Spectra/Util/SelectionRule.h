// Selection / sorting rules of the wanted eigenvalues (reference: Util/SelectionRule.h:33-58 for the
// enumeration — same order, so integer values survive the C ABI — and :195-287 for argsort).
#ifndef MISPEC_SPECTRA_SELECTION_RULE_H
#define MISPEC_SPECTRA_SELECTION_RULE_H

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstddef>
#include <stdexcept>
#include <utility>
#include <vector>

namespace Spectra {

enum class SortRule
{
    LargestMagn,   // largest |lambda|            (symmetric and general solvers)
    LargestReal,   // largest Re(lambda)          (general solvers)
    LargestImag,   // largest |Im(lambda)|        (general solvers)
    LargestAlge,   // largest lambda, signed      (symmetric solvers)
    SmallestMagn,  // smallest |lambda|
    SmallestReal,  // smallest Re(lambda)
    SmallestImag,  // smallest |Im(lambda)|
    SmallestAlge,  // smallest lambda, signed
    BothEnds       // alternately from the top and the bottom of the spectrum (symmetric solvers)
};

namespace internal {

// The key std::sort orders by (ascending); "largest" rules negate.
inline double sort_key(SortRule rule, double v)
{
    switch (rule)
    {
        case SortRule::LargestMagn:
            return -std::abs(v);
        case SortRule::LargestAlge:
        case SortRule::BothEnds:
            return -v;
        case SortRule::SmallestMagn:
            return std::abs(v);
        case SortRule::SmallestAlge:
            return v;
        default:
            throw std::invalid_argument("unsupported selection rule");
    }
}
inline double sort_key(SortRule rule, const std::complex<double>& v)
{
    switch (rule)
    {
        case SortRule::LargestMagn:
            return -std::abs(v);
        case SortRule::LargestReal:
            return -v.real();
        case SortRule::LargestImag:
            return -std::abs(v.imag());
        case SortRule::SmallestMagn:
            return std::abs(v);
        case SortRule::SmallestReal:
            return v.real();
        case SortRule::SmallestImag:
            return std::abs(v.imag());
        default:
            throw std::invalid_argument("unsupported selection rule");
    }
}

}  // namespace internal

// Indices that put values[0..len) in the order of `selection`.  std::sort (not stable), as in the
// reference, so ties break the same way under the same standard library.  BothEnds interleaves
// largest, smallest, 2nd largest, 2nd smallest, ... so that any leading k entries are the wanted set.
template <typename T>
std::vector<std::ptrdiff_t> argsort(SortRule selection, const T* values, std::ptrdiff_t len)
{
    (void) internal::sort_key(selection, T());  // reject rules that do not apply to T before sorting
    std::vector<std::ptrdiff_t> ind(static_cast<std::size_t>(len));
    for (std::ptrdiff_t i = 0; i < len; i++)
        ind[static_cast<std::size_t>(i)] = i;
    std::sort(ind.begin(), ind.end(), [&](std::ptrdiff_t a, std::ptrdiff_t b) {
        return internal::sort_key(selection, values[a]) < internal::sort_key(selection, values[b]);
    });
    if (selection == SortRule::BothEnds)
    {
        const std::vector<std::ptrdiff_t> sorted(ind);
        for (std::ptrdiff_t i = 0; i < len; i++)
            ind[static_cast<std::size_t>(i)] = (i % 2 == 0) ? sorted[static_cast<std::size_t>(i / 2)]
                                                            : sorted[static_cast<std::size_t>(len - 1 - i / 2)];
    }
    return ind;
}

// The compile-time form of the reference (Util/SelectionRule.h:195-224): SortEigenvalue<T, Rule>(ptr, n).index() is
// the permutation that sorts the n values by `Rule` (BothEnds sorts like LargestAlge here; the interleaving is
// argsort's job, as upstream).  Kept for user code that names the class; the solvers use argsort.
template <typename T, SortRule Rule>
class SortEigenvalue
{
    std::vector<std::ptrdiff_t> m_index;

public:
    SortEigenvalue(const T* start, std::ptrdiff_t size) : m_index(static_cast<std::size_t>(size))
    {
        (void) internal::sort_key(Rule, T());
        for (std::ptrdiff_t i = 0; i < size; i++)
            m_index[static_cast<std::size_t>(i)] = i;
        std::sort(m_index.begin(), m_index.end(), [start](std::ptrdiff_t a, std::ptrdiff_t b) {
            return internal::sort_key(Rule, start[a]) < internal::sort_key(Rule, start[b]);
        });
    }
    std::vector<std::ptrdiff_t> index() const { return m_index; }
    void swap(std::vector<std::ptrdiff_t>& other) { m_index.swap(other); }
};

// Container forms (anything with data() and size(): Eigen vectors, std::vector, DenseVector)
template <typename Vec, typename = decltype(std::declval<const Vec&>().data())>
std::vector<std::ptrdiff_t> argsort(SortRule selection, const Vec& values, std::ptrdiff_t len)
{
    return argsort(selection, values.data(), len);
}
template <typename Vec, typename = decltype(std::declval<const Vec&>().data())>
std::vector<std::ptrdiff_t> argsort(SortRule selection, const Vec& values)
{
    return argsort(selection, values.data(), static_cast<std::ptrdiff_t>(values.size()));
}

}  // namespace Spectra

#endif
