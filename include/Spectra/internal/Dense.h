// Dense value types handed back by eigenvalues() / eigenvectors().
//
// The reference returns Eigen::Matrix / Eigen::Vector by value (HermEigsBase.h:417, :447).  Eigen 3.4.0
// is the reference's only dependency and is NOT vendored with it; when <Eigen/Core> is on the include
// path these aliases ARE the Eigen types, so user code written against Spectra compiles unchanged.
// Without Eigen (this repository's own build and tests) a minimal column-major container with the
// same element access (operator(), operator[], data(), rows(), cols(), size(), col()) stands in.
#ifndef MISPEC_SPECTRA_DENSE_H
#define MISPEC_SPECTRA_DENSE_H

#include <cstddef>
#include <type_traits>
#include <stdexcept>
#include <vector>

#if !defined(MISPEC_NO_EIGEN) && defined(__has_include)
#if __has_include(<Eigen/Core>)
#define MISPEC_HAVE_EIGEN 1
#include <Eigen/Core>
#include <Eigen/SparseCore>
#endif
#endif

namespace Spectra {

using Index = std::ptrdiff_t;  // Eigen::Index

// Values of Eigen::Lower / Eigen::Upper / Eigen::ColMajor / Eigen::RowMajor, for the Uplo / Flags
// template parameters of the matrix operators when Eigen itself is not there.
constexpr int Lower = 1;
constexpr int Upper = 2;
constexpr int ColMajor = 0;
constexpr int RowMajor = 1;

namespace internal {

template <typename T>
class PlainVector
{
    std::vector<T> m_data;

public:
    PlainVector() {}
    explicit PlainVector(Index n) : m_data(static_cast<std::size_t>(n)) {}
    Index size() const { return static_cast<Index>(m_data.size()); }
    Index rows() const { return size(); }
    Index cols() const { return 1; }
    void resize(Index n) { m_data.assign(static_cast<std::size_t>(n), T()); }
    T& operator[](Index i) { return m_data[static_cast<std::size_t>(i)]; }
    const T& operator[](Index i) const { return m_data[static_cast<std::size_t>(i)]; }
    T& operator()(Index i) { return (*this)[i]; }
    const T& operator()(Index i) const { return (*this)[i]; }
    T* data() { return m_data.data(); }
    const T* data() const { return m_data.data(); }
    const T* begin() const { return m_data.data(); }
    const T* end() const { return m_data.data() + m_data.size(); }
};

template <typename T>
class PlainMatrix  // column-major
{
    Index m_rows = 0, m_cols = 0;
    std::vector<T> m_data;

public:
    PlainMatrix() {}
    PlainMatrix(Index r, Index c) : m_rows(r), m_cols(c), m_data(static_cast<std::size_t>(r) * static_cast<std::size_t>(c)) {}
    Index rows() const { return m_rows; }
    Index cols() const { return m_cols; }
    Index size() const { return m_rows * m_cols; }
    void resize(Index r, Index c)
    {
        m_rows = r;
        m_cols = c;
        m_data.assign(static_cast<std::size_t>(r) * static_cast<std::size_t>(c), T());
    }
    T& operator()(Index i, Index j) { return m_data[static_cast<std::size_t>(j) * m_rows + i]; }
    const T& operator()(Index i, Index j) const { return m_data[static_cast<std::size_t>(j) * m_rows + i]; }
    T* data() { return m_data.data(); }
    const T* data() const { return m_data.data(); }
    T* col(Index j) { return m_data.data() + static_cast<std::size_t>(j) * m_rows; }
    const T* col(Index j) const { return m_data.data() + static_cast<std::size_t>(j) * m_rows; }
};

}  // namespace internal

// fp32 at the operator boundary.  The device computes in fp64 only; the matrix operators also accept Scalar = float, as the
// reference's do (test/SparseSymMatProd.cpp:37, TEMPLATE_TEST_CASE over float and double): the values are widened on the way in
// (exact) and the result is rounded once on the way out.  For Scalar = double these are the caller's own pointers, no copy.
namespace internal {
template <typename T>
struct is_device_scalar
{
    static constexpr bool value = false;
};
template <>
struct is_device_scalar<double>
{
    static constexpr bool value = true;
};
template <>
struct is_device_scalar<float>
{
    static constexpr bool value = true;
};

template <typename T>
class WidenedIn;  // const T[count] seen as const double*
template <>
class WidenedIn<double>
{
    const double* m_p;

public:
    WidenedIn(const double* p, std::size_t) : m_p(p) {}
    const double* data() const { return m_p; }
};
template <>
class WidenedIn<float>
{
    std::vector<double> m_buf;

public:
    WidenedIn(const float* p, std::size_t count) : m_buf(p, p + (p ? count : 0)) {}
    const double* data() const { return m_buf.data(); }
};

// The index arrays of a compressed sparse matrix seen as const int32_t* — what the device stores (mispec_csr: int32 indices,
// n < 2^31).  StorageIndex = int: the caller's arrays themselves; any other integer type (the reference takes any StorageIndex,
// MatOp/SparseSymMatProd.h:30): a narrowed copy, std::invalid_argument if an index does not fit.
template <typename I>
class Int32Indices
{
    std::vector<int> m_buf;
    const int* m_p;

public:
    Int32Indices(const I* p, std::size_t count) : m_p(nullptr)
    {
        static_assert(std::is_integral<I>::value, "StorageIndex must be an integer type");
        m_buf.resize(p ? count : 0);
        for (std::size_t i = 0; i < m_buf.size(); i++)
        {
            if (p[i] < I(0) || static_cast<unsigned long long>(p[i]) > 2147483647ull)
                throw std::invalid_argument("sparse matrix: an index does not fit the device's int32 indices");
            m_buf[i] = static_cast<int>(p[i]);
        }
        m_p = m_buf.data();
    }
    const int* data() const { return m_p; }
};
template <>
class Int32Indices<int>
{
    const int* m_p;

public:
    Int32Indices(const int* p, std::size_t) : m_p(p) {}
    const int* data() const { return m_p; }
};

template <typename T>
class NarrowedOut;  // T[count] written through a double*; store() rounds into the destination
template <>
class NarrowedOut<double>
{
    double* m_p;

public:
    NarrowedOut(double* p, std::size_t) : m_p(p) {}
    double* data() { return m_p; }
    void store() {}
};
template <>
class NarrowedOut<float>
{
    float* m_dst;
    std::vector<double> m_buf;

public:
    NarrowedOut(float* p, std::size_t count) : m_dst(p), m_buf(count) {}
    double* data() { return m_buf.data(); }
    void store()
    {
        for (std::size_t i = 0; i < m_buf.size(); i++)
            m_dst[i] = static_cast<float>(m_buf[i]);
    }
};
}  // namespace internal

#ifdef MISPEC_HAVE_EIGEN
template <typename T>
using DenseVector = Eigen::Matrix<T, Eigen::Dynamic, 1>;
template <typename T>
using DenseMatrix = Eigen::Matrix<T, Eigen::Dynamic, Eigen::Dynamic>;
#else
template <typename T>
using DenseVector = internal::PlainVector<T>;
template <typename T>
using DenseMatrix = internal::PlainMatrix<T>;
#endif

// A borrowed view of a compressed sparse matrix (what an Eigen::SparseMatrix in compressed mode holds):
// outer[outer_size+1], inner[nnz], values[nnz]; row_major tells whether "outer" runs over rows (CSR)
// or columns (CSC, Eigen's default).
template <typename Scalar, typename StorageIndex = int>
struct SparseView
{
    Index rows = 0, cols = 0;
    const StorageIndex* outer = nullptr;
    const StorageIndex* inner = nullptr;
    const Scalar* values = nullptr;
    bool row_major = false;
};

// A borrowed view of a dense matrix in host memory: rows x cols with leading dimension ld, column-major
// (Eigen's default) unless row_major.
template <typename Scalar>
struct DenseView
{
    Index rows = 0, cols = 0;
    const Scalar* data = nullptr;
    Index ld = 0;
    bool row_major = false;

    DenseView() {}
    DenseView(Index r, Index c, const Scalar* d, Index ld_, bool rm = false) : rows(r), cols(c), data(d), ld(ld_), row_major(rm) {}
    // the stand-in matrix type (column-major, contiguous)
    DenseView(const internal::PlainMatrix<Scalar>& m) : rows(m.rows()), cols(m.cols()), data(m.data()), ld(m.rows()), row_major(false) {}
#ifdef MISPEC_HAVE_EIGEN
    // a plain (contiguous) Eigen matrix of either storage order
    template <int Options>
    DenseView(const Eigen::Matrix<Scalar, Eigen::Dynamic, Eigen::Dynamic, Options>& m) :
        rows(m.rows()), cols(m.cols()), data(m.data()), ld(m.outerStride()), row_major((Options & Eigen::RowMajorBit) != 0)
    {}
#endif
};

}  // namespace Spectra

#endif
