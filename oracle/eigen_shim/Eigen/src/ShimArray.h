// TEST INFRASTRUCTURE — NOT EIGEN.  See ../Core.  Coefficient-wise "array" objects: eager (every operation returns an owned
// Array); the reference uses them on ncv-sized data only.
#ifndef ORACLE_EIGEN_SHIM_ARRAY_H
#define ORACLE_EIGEN_SHIM_ARRAY_H

namespace Eigen {

namespace internal {
template <typename T, int R, int C, int O>
struct traits<Array<T, R, C, O>>
{
    typedef T Scalar;
    enum
    {
        Rows = R,
        Cols = C,
        RowMajor = 0,
        Direct = 1
    };
};
template <typename D>
struct traits<ArrayLvalue<D>>
{
    typedef typename traits<D>::Scalar Scalar;
    enum
    {
        Rows = traits<D>::Rows,
        Cols = traits<D>::Cols,
        RowMajor = 0,
        Direct = 0
    };
};
template <typename D>
struct lv_nested
{
    typedef D type;  // views: a copy of the handle
};
template <typename T, int R, int C, int O>
struct lv_nested<Matrix<T, R, C, O>>
{
    typedef Matrix<T, R, C, O>& type;
};
}  // namespace internal

template <typename Derived>
class ArrayBase
{
public:
    typedef internal::traits<Derived> Traits;
    typedef typename Traits::Scalar Scalar;
    typedef typename NumTraits<Scalar>::Real RealScalar;
    typedef Array<Scalar, Traits::Rows, Traits::Cols> PlainObject;
    typedef Array<RealScalar, Traits::Rows, Traits::Cols> RealPlain;
    typedef Array<bool, Traits::Rows, Traits::Cols> BoolPlain;
    const Derived& derived() const { return *static_cast<const Derived*>(this); }
    Derived& derived() { return *static_cast<Derived*>(this); }
    Index rows() const { return derived().rows(); }
    Index cols() const { return derived().cols(); }
    Index size() const { return rows() * cols(); }
    Scalar coeff(Index i, Index j) const { return derived().coeff(i, j); }
    Scalar coeff(Index i) const
    {
        const Index r = rows();
        return r == 1 ? derived().coeff(0, i) : derived().coeff(i % r, i / r);
    }
    Scalar operator()(Index i, Index j) const { return coeff(i, j); }
    Scalar operator()(Index i) const { return coeff(i); }
    Scalar operator[](Index i) const { return coeff(i); }

    template <typename F>
    Array<typename std::decay<decltype(std::declval<F>()(std::declval<Scalar>()))>::type, Traits::Rows, Traits::Cols> unary(F f) const
    {
        typedef typename std::decay<decltype(f(std::declval<Scalar>()))>::type S2;
        Array<S2, Traits::Rows, Traits::Cols> r(rows(), cols());
        for (Index j = 0; j < cols(); j++)
            for (Index i = 0; i < rows(); i++)
                r.coeffRef(i, j) = f(coeff(i, j));
        return r;
    }
    template <typename O, typename F>
    Array<typename std::decay<decltype(std::declval<F>()(std::declval<Scalar>(), std::declval<typename internal::traits<O>::Scalar>()))>::type, Traits::Rows, Traits::Cols> binary(const ArrayBase<O>& o, F f) const
    {
        typedef typename std::decay<decltype(f(std::declval<Scalar>(), std::declval<typename internal::traits<O>::Scalar>()))>::type S2;
        Array<S2, Traits::Rows, Traits::Cols> r(rows(), cols());
        for (Index j = 0; j < cols(); j++)
            for (Index i = 0; i < rows(); i++)
                r.coeffRef(i, j) = f(coeff(i, j), o.coeff(i, j));
        return r;
    }
    RealPlain abs() const { return unary([](const Scalar& a) { return numext::abs(a); }); }
    RealPlain abs2() const { return unary([](const Scalar& a) { return numext::abs2(a); }); }
    PlainObject sqrt() const { return unary([](const Scalar& a) { return Scalar(std::sqrt(a)); }); }
    PlainObject square() const { return unary([](const Scalar& a) { return Scalar(a * a); }); }
    PlainObject inverse() const { return unary([](const Scalar& a) { return Scalar(Scalar(1) / a); }); }
    PlainObject exp() const { return unary([](const Scalar& a) { return Scalar(std::exp(a)); }); }
    PlainObject log() const { return unary([](const Scalar& a) { return Scalar(std::log(a)); }); }
    PlainObject log10() const { return unary([](const Scalar& a) { return Scalar(std::log10(a)); }); }
    PlainObject pow(const RealScalar& p) const { return unary([p](const Scalar& a) { return Scalar(std::pow(a, p)); }); }
    RealPlain real() const { return unary([](const Scalar& a) { return numext::real(a); }); }
    RealPlain imag() const { return unary([](const Scalar& a) { return numext::imag(a); }); }
    PlainObject operator-() const { return unary([](const Scalar& a) { return Scalar(-a); }); }
    PlainObject max(const Scalar& s) const { return unary([s](const Scalar& a) { return numext::maxi(a, s); }); }
    PlainObject min(const Scalar& s) const { return unary([s](const Scalar& a) { return numext::mini(a, s); }); }
    template <typename O>
    PlainObject max(const ArrayBase<O>& o) const { return binary(o, [](const Scalar& a, const Scalar& b) { return numext::maxi(a, b); }); }
    template <typename O>
    PlainObject min(const ArrayBase<O>& o) const { return binary(o, [](const Scalar& a, const Scalar& b) { return numext::mini(a, b); }); }
    template <typename U>
    Array<U, Traits::Rows, Traits::Cols> cast() const { return unary([](const Scalar& a) { return internal::op_cast<U>()(a); }); }
    BoolPlain isFinite() const { return unary([](const Scalar& a) { return bool(numext::isfinite(a)); }); }
    BoolPlain operator!() const { return unary([](const Scalar& a) { return !a; }); }

    Scalar sum() const
    {
        Scalar s = Scalar(0);
        for (Index j = 0; j < cols(); j++)
            for (Index i = 0; i < rows(); i++)
                s += coeff(i, j);
        return s;
    }
    Scalar prod() const
    {
        Scalar s = Scalar(1);
        for (Index j = 0; j < cols(); j++)
            for (Index i = 0; i < rows(); i++)
                s *= coeff(i, j);
        return s;
    }
    Scalar mean() const { return sum() / Scalar(RealScalar(size())); }
    Scalar maxCoeff() const
    {
        Scalar b = coeff(0, 0);
        for (Index j = 0; j < cols(); j++)
            for (Index i = 0; i < rows(); i++)
                if (coeff(i, j) > b)
                    b = coeff(i, j);
        return b;
    }
    Scalar minCoeff() const
    {
        Scalar b = coeff(0, 0);
        for (Index j = 0; j < cols(); j++)
            for (Index i = 0; i < rows(); i++)
                if (coeff(i, j) < b)
                    b = coeff(i, j);
        return b;
    }
    Index count() const
    {
        Index n = 0;
        for (Index j = 0; j < cols(); j++)
            for (Index i = 0; i < rows(); i++)
                if (coeff(i, j))
                    n++;
        return n;
    }
    bool all() const { return count() == size(); }
    bool any() const { return count() > 0; }
    Matrix<Scalar, Traits::Rows, Traits::Cols> matrix() const
    {
        Matrix<Scalar, Traits::Rows, Traits::Cols> m(rows(), cols());
        for (Index j = 0; j < cols(); j++)
            for (Index i = 0; i < rows(); i++)
                m.coeffRef(i, j) = coeff(i, j);
        return m;
    }
    PlainObject eval() const { return PlainObject(derived()); }
    // rowwise() / colwise() broadcasting of a vector over the rows / columns (eager): a.rowwise() / r divides row i by r entrywise
    template <bool ByRow>
    class Broadcast
    {
        const Derived& m_a;

    public:
        explicit Broadcast(const Derived& a) : m_a(a) {}
        template <typename F, typename O>
        Array<Scalar, Dynamic, Dynamic> apply(const ArrayBase<O>& v, F f) const
        {
            Array<Scalar, Dynamic, Dynamic> r(m_a.rows(), m_a.cols());
            for (Index j = 0; j < m_a.cols(); j++)
                for (Index i = 0; i < m_a.rows(); i++)
                    r.coeffRef(i, j) = f(m_a.coeff(i, j), v.coeff(ByRow ? j : i));
            return r;
        }
        template <typename O>
        Array<Scalar, Dynamic, Dynamic> operator/(const ArrayBase<O>& v) const
        {
            return apply(v, [](const Scalar& a, const Scalar& b) { return a / b; });
        }
        template <typename O>
        Array<Scalar, Dynamic, Dynamic> operator*(const ArrayBase<O>& v) const
        {
            return apply(v, [](const Scalar& a, const Scalar& b) { return a * b; });
        }
        template <typename O>
        Array<Scalar, Dynamic, Dynamic> operator+(const ArrayBase<O>& v) const
        {
            return apply(v, [](const Scalar& a, const Scalar& b) { return a + b; });
        }
        template <typename O>
        Array<Scalar, Dynamic, Dynamic> operator-(const ArrayBase<O>& v) const
        {
            return apply(v, [](const Scalar& a, const Scalar& b) { return a - b; });
        }
    };
    Broadcast<true> rowwise() const { return Broadcast<true>(derived()); }
    Broadcast<false> colwise() const { return Broadcast<false>(derived()); }
    PlainObject segment(Index start, Index k) const
    {
        PlainObject r(rows() == 1 && Traits::Cols != 1 ? 1 : k, rows() == 1 && Traits::Cols != 1 ? k : 1);
        for (Index i = 0; i < k; i++)
            r.coeffRef(i) = coeff(start + i);
        return r;
    }
    PlainObject head(Index k) const { return segment(0, k); }
    PlainObject tail(Index k) const { return segment(size() - k, k); }
    template <typename Then, typename Else>
    typename Then::PlainObject select(const ArrayBase<Then>& a, const ArrayBase<Else>& b) const
    {
        typename Then::PlainObject r(rows(), cols());
        for (Index j = 0; j < cols(); j++)
            for (Index i = 0; i < rows(); i++)
                r.coeffRef(i, j) = coeff(i, j) ? a.coeff(i, j) : b.coeff(i, j);
        return r;
    }
};

// ---- owned array -----------------------------------------------------------------------------------------------------------------
template <typename T, int Rows, int Cols, int Options>
class Array : public ArrayBase<Array<T, Rows, Cols, Options>>
{
    Index m_rows, m_cols;
    internal::Buf<T> m_data;
    static Index dim0(int d) { return d == Dynamic ? 0 : d; }

public:
    typedef T Scalar;
    typedef ArrayBase<Array> Base;
    Array() : m_rows(dim0(Rows)), m_cols(dim0(Cols)), m_data(static_cast<std::size_t>(dim0(Rows) * dim0(Cols))) {}
    template <typename I, typename = typename std::enable_if<std::is_integral<I>::value>::type>
    explicit Array(I n) : m_rows(Rows == 1 ? 1 : Index(n)), m_cols(Rows == 1 ? Index(n) : 1), m_data(static_cast<std::size_t>(n))
    {}
    Array(Index r, Index c) : m_rows(r), m_cols(c), m_data(static_cast<std::size_t>(r) * static_cast<std::size_t>(c)) {}
    Array(const Array&) = default;
    Array(Array&&) = default;
    template <typename D>
    Array(const ArrayBase<D>& o) : m_rows(0), m_cols(0)
    {
        *this = o;
    }
    // from a matrix expression (Eigen allows the assignment; a vector of the other orientation is transposed, as Eigen does:
    // RitzPairs.h:75 initialises a column Array from colwise().norm(), a row vector)
    template <typename D>
    Array(const MatrixBase<D>& o) : m_rows(o.rows()), m_cols(o.cols()), m_data(static_cast<std::size_t>(o.rows() * o.cols()))
    {
        const bool flip = (Cols == 1 && Rows != 1 && o.rows() == 1 && o.cols() != 1) || (Rows == 1 && Cols != 1 && o.cols() == 1 && o.rows() != 1);
        if (flip)
            std::swap(m_rows, m_cols);
        for (Index j = 0; j < o.cols(); j++)
            for (Index i = 0; i < o.rows(); i++)
                (flip ? coeffRef(j, i) : coeffRef(i, j)) = o.derived().coeff(i, j);
    }
    Array& operator=(const Array&) = default;
    Array& operator=(Array&&) = default;
    template <typename D>
    Array& operator=(const ArrayBase<D>& o)
    {
        resize(o.rows(), o.cols());
        for (Index j = 0; j < m_cols; j++)
            for (Index i = 0; i < m_rows; i++)
                coeffRef(i, j) = T(o.coeff(i, j));
        return *this;
    }
    Index rows() const { return m_rows; }
    Index cols() const { return m_cols; }
    Index size() const { return m_rows * m_cols; }
    void resize(Index n)
    {
        if (Rows == 1)
            resize(1, n);
        else
            resize(n, 1);
    }
    void resize(Index r, Index c)
    {
        if (r == m_rows && c == m_cols)
            return;
        m_data.assign_size(static_cast<std::size_t>(r) * static_cast<std::size_t>(c));
        m_rows = r;
        m_cols = c;
    }
    T* data() { return m_data.data(); }
    const T* data() const { return m_data.data(); }
    T coeff(Index i, Index j) const { return m_data.data()[j * m_rows + i]; }
    T coeff(Index i) const { return m_data.data()[i]; }
    T& coeffRef(Index i, Index j) { return m_data.data()[j * m_rows + i]; }
    T& coeffRef(Index i) { return m_data.data()[i]; }
    T& operator()(Index i, Index j) { return coeffRef(i, j); }
    const T& operator()(Index i, Index j) const { return m_data.data()[j * m_rows + i]; }
    T& operator()(Index i) { return coeffRef(i); }
    const T& operator()(Index i) const { return m_data.data()[i]; }
    T& operator[](Index i) { return coeffRef(i); }
    const T& operator[](Index i) const { return m_data.data()[i]; }
    Array& setConstant(const T& v)
    {
        std::fill(m_data.data(), m_data.data() + m_data.size(), v);
        return *this;
    }
    Array& setZero() { return setConstant(T(0)); }
    Array& setOnes() { return setConstant(T(1)); }
    Array& fill(const T& v) { return setConstant(v); }
    void swap(Array& o)
    {
        std::swap(m_rows, o.m_rows);
        std::swap(m_cols, o.m_cols);
        m_data.swap(o.m_data);
    }
    static Array Zero(Index n)
    {
        Array a(n);
        a.setZero();
        return a;
    }
    static Array Zero(Index r, Index c)
    {
        Array a(r, c);
        a.setZero();
        return a;
    }
    static Array Constant(Index n, const T& v)
    {
        Array a(n);
        a.setConstant(v);
        return a;
    }
    static Array Constant(Index r, Index c, const T& v)
    {
        Array a(r, c);
        a.setConstant(v);
        return a;
    }
    static Array Ones(Index n) { return Constant(n, T(1)); }
    static Array LinSpaced(Index n, const T& lo, const T& hi)
    {
        Array a(n);
        for (Index i = 0; i < n; i++)
            a[i] = n == 1 ? hi : lo + T(i) * (hi - lo) / T(n - 1);
        return a;
    }
#define ESHIM_ARRAY_COMPOUND(OP)                          \
    template <typename D>                                 \
    Array& operator OP(const ArrayBase<D>& o)             \
    {                                                     \
        for (Index j = 0; j < m_cols; j++)                \
            for (Index i = 0; i < m_rows; i++)            \
                coeffRef(i, j) OP o.coeff(i, j);          \
        return *this;                                     \
    }                                                     \
    Array& operator OP(const T& s)                        \
    {                                                     \
        for (Index i = 0; i < size(); i++)                \
            coeffRef(i) OP s;                             \
        return *this;                                     \
    }
    ESHIM_ARRAY_COMPOUND(+=)
    ESHIM_ARRAY_COMPOUND(-=)
    ESHIM_ARRAY_COMPOUND(*=)
    ESHIM_ARRAY_COMPOUND(/=)
#undef ESHIM_ARRAY_COMPOUND
};

typedef Array<double, Dynamic, 1> ArrayXd;
typedef Array<double, Dynamic, Dynamic> ArrayXXd;
typedef Array<int, Dynamic, 1> ArrayXi;
typedef Array<bool, Dynamic, 1> ArrayXb;

// ---- a matrix lvalue seen as an array: M.array() += s ------------------------------------------------------------------------------
template <typename D>
class ArrayLvalue : public ArrayBase<ArrayLvalue<D>>
{
    typename internal::lv_nested<D>::type m_x;

public:
    typedef typename internal::traits<D>::Scalar Scalar;
    explicit ArrayLvalue(D& x) : m_x(x) {}
    Index rows() const { return m_x.rows(); }
    Index cols() const { return m_x.cols(); }
    Scalar coeff(Index i, Index j) const { return m_x.coeff(i, j); }
    using ArrayBase<ArrayLvalue>::coeff;
    template <typename O>
    ArrayLvalue& operator=(const ArrayBase<O>& o)
    {
        for (Index j = 0; j < cols(); j++)
            for (Index i = 0; i < rows(); i++)
                m_x.coeffRef(i, j) = o.coeff(i, j);
        return *this;
    }
    ArrayLvalue& operator=(const ArrayLvalue& o) { return this->template operator=<ArrayLvalue>(o); }
#define ESHIM_ARRAYLV_COMPOUND(OP)                        \
    template <typename O>                                 \
    ArrayLvalue& operator OP(const ArrayBase<O>& o)       \
    {                                                     \
        for (Index j = 0; j < cols(); j++)                \
            for (Index i = 0; i < rows(); i++)            \
                m_x.coeffRef(i, j) OP o.coeff(i, j);      \
        return *this;                                     \
    }                                                     \
    ArrayLvalue& operator OP(const Scalar& s)             \
    {                                                     \
        for (Index j = 0; j < cols(); j++)                \
            for (Index i = 0; i < rows(); i++)            \
                m_x.coeffRef(i, j) OP s;                  \
        return *this;                                     \
    }
    ESHIM_ARRAYLV_COMPOUND(+=)
    ESHIM_ARRAYLV_COMPOUND(-=)
    ESHIM_ARRAYLV_COMPOUND(*=)
    ESHIM_ARRAYLV_COMPOUND(/=)
#undef ESHIM_ARRAYLV_COMPOUND
};

// ---- binary operators ----------------------------------------------------------------------------------------------------------------
#define ESHIM_ARRAY_BINARY(OP)                                                                                                     \
    template <typename A, typename B>                                                                                              \
    Array<typename internal::promote<typename internal::traits<A>::Scalar, typename internal::traits<B>::Scalar>::type, internal::traits<A>::Rows, internal::traits<A>::Cols> \
    operator OP(const ArrayBase<A>& a, const ArrayBase<B>& b)                                                                      \
    {                                                                                                                              \
        typedef typename internal::traits<A>::Scalar SA;                                                                           \
        typedef typename internal::traits<B>::Scalar SB;                                                                           \
        typedef typename internal::promote<SA, SB>::type S;                                                                        \
        return a.binary(b, [](const SA& x, const SB& y) { return S(x OP y); });                                                    \
    }                                                                                                                              \
    template <typename A>                                                                                                          \
    typename ArrayBase<A>::PlainObject operator OP(const ArrayBase<A>& a, const typename internal::traits<A>::Scalar& s)           \
    {                                                                                                                              \
        typedef typename internal::traits<A>::Scalar SA;                                                                           \
        return a.unary([s](const SA& x) { return SA(x OP s); });                                                                   \
    }                                                                                                                              \
    template <typename A>                                                                                                          \
    typename ArrayBase<A>::PlainObject operator OP(const typename internal::traits<A>::Scalar& s, const ArrayBase<A>& a)           \
    {                                                                                                                              \
        typedef typename internal::traits<A>::Scalar SA;                                                                           \
        return a.unary([s](const SA& x) { return SA(s OP x); });                                                                   \
    }
ESHIM_ARRAY_BINARY(+)
ESHIM_ARRAY_BINARY(-)
ESHIM_ARRAY_BINARY(*)
ESHIM_ARRAY_BINARY(/)
#undef ESHIM_ARRAY_BINARY
#define ESHIM_ARRAY_COMPARE(OP)                                                                                                    \
    template <typename A, typename B>                                                                                              \
    typename ArrayBase<A>::BoolPlain operator OP(const ArrayBase<A>& a, const ArrayBase<B>& b)                                     \
    {                                                                                                                              \
        typedef typename internal::traits<A>::Scalar SA;                                                                           \
        typedef typename internal::traits<B>::Scalar SB;                                                                           \
        return a.binary(b, [](const SA& x, const SB& y) { return bool(x OP y); });                                                 \
    }                                                                                                                              \
    template <typename A>                                                                                                          \
    typename ArrayBase<A>::BoolPlain operator OP(const ArrayBase<A>& a, const typename internal::traits<A>::Scalar& s)             \
    {                                                                                                                              \
        typedef typename internal::traits<A>::Scalar SA;                                                                           \
        return a.unary([s](const SA& x) { return bool(x OP s); });                                                                 \
    }                                                                                                                              \
    template <typename A>                                                                                                          \
    typename ArrayBase<A>::BoolPlain operator OP(const typename internal::traits<A>::Scalar& s, const ArrayBase<A>& a)             \
    {                                                                                                                              \
        typedef typename internal::traits<A>::Scalar SA;                                                                           \
        return a.unary([s](const SA& x) { return bool(s OP x); });                                                                 \
    }
ESHIM_ARRAY_COMPARE(<)
ESHIM_ARRAY_COMPARE(<=)
ESHIM_ARRAY_COMPARE(>)
ESHIM_ARRAY_COMPARE(>=)
ESHIM_ARRAY_COMPARE(==)
ESHIM_ARRAY_COMPARE(!=)
ESHIM_ARRAY_COMPARE(&&)
ESHIM_ARRAY_COMPARE(||)
#undef ESHIM_ARRAY_COMPARE

template <typename D>
std::ostream& operator<<(std::ostream& os, const ArrayBase<D>& a)
{
    return os << a.matrix();
}

// ---- matrix objects written from arrays -------------------------------------------------------------------------------------------
template <typename T, int R, int C, int O>
template <typename D2>
Matrix<T, R, C, O>::Matrix(const ArrayBase<D2>& a) : m_rows(0), m_cols(0)
{
    *this = a;
}
template <typename T, int R, int C, int O>
template <typename D2>
Matrix<T, R, C, O>& Matrix<T, R, C, O>::operator=(const ArrayBase<D2>& a)
{
    resize(a.rows(), a.cols());
    for (Index j = 0; j < m_cols; j++)
        for (Index i = 0; i < m_rows; i++)
            coeffRef(i, j) = a.coeff(i, j);
    return *this;
}
template <typename T, int R, int C, bool RM>
template <typename D2>
View<T, R, C, RM>& View<T, R, C, RM>::operator=(const ArrayBase<D2>& a)
{
    for (Index j = 0; j < m_cols; j++)
        for (Index i = 0; i < m_rows; i++)
            this->coeffRef(i, j) = a.coeff(i, j);
    return *this;
}

}  // namespace Eigen

#endif
