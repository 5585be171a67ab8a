// The factorisation  A V = V H + f e'  held in GPU memory.
//
// Counterpart of the reference's Arnoldi<ArnoldiOpType> (LinAlg/Arnoldi.h:32-343): same life cycle —
// init(), factorize_from(from_k, to_m, op_counter), compress, matrix_H(), f_norm(), subspace_dim() —
// but V (n x m), f and the work vectors never leave HBM.  The class is a thin owner of a `mispec_fac`
// handle (include/mispec.h); the arithmetic is in spectra_amd/csrc/{csr,krylov,small,fac}.hip.
//
// Operators.  If OpType exposes a device matrix (SparseSymMatProd / SparseGenMatProd do, through
// mispec_matrix()), the factorisation binds it and each step is: all-gather (sharded runs only) ->
// fused SpMV -> one or two passes over V.  Any other OpType is used through the reference's own
// contract, perform_op(const Scalar* x_in, Scalar* y_out) on HOST pointers: x is copied out of HBM, the
// user's code runs, y is copied back in (2 x 8n bytes over PCIe per step) — everything else stays on
// the device.
#ifndef MISPEC_SPECTRA_ARNOLDI_H
#define MISPEC_SPECTRA_ARNOLDI_H

#include <complex>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <type_traits>
#include <utility>
#include <vector>

#include "../MatOp/internal/ArnoldiOp.h"
#include "../internal/ComplexDense.h"
#include "../internal/Dense.h"
#include "../internal/Device.h"

namespace Spectra {

namespace internal {
// std::void_t for C++11/14 (the reference only requires C++11)
template <typename... Ts>
struct make_void
{
    typedef void type;
};
template <typename... Ts>
using void_t = typename make_void<Ts...>::type;

template <typename T, typename = void>
struct has_device_matrix : std::false_type
{};
template <typename T>
struct has_device_matrix<T, void_t<decltype(std::declval<const T&>().mispec_matrix())>> : std::true_type
{};
template <typename T, typename = void>
struct has_device_solver : std::false_type
{};
template <typename T>
struct has_device_solver<T, void_t<decltype(std::declval<const T&>().mispec_solver())>> : std::true_type
{};
template <typename T, typename = void>
struct has_device_product : std::false_type
{};
template <typename T>
struct has_device_product<T, void_t<decltype(std::declval<const T&>().mispec_product_second())>> : std::true_type
{};
template <typename T, typename = void>
struct has_device_geigs : std::false_type
{};
template <typename T>
struct has_device_geigs<T, void_t<decltype(std::declval<const T&>().mispec_geigs_b_operator())>> : std::true_type
{};
template <typename T, typename = void>
struct has_device_geigs_cholesky : std::false_type
{};
template <typename T>
struct has_device_geigs_cholesky<T, void_t<decltype(std::declval<const T&>().mispec_geigs_cholesky_matrix())>> : std::true_type
{};
template <typename T, typename = void>
struct has_device_geigs_shift : std::false_type
{};
template <typename T>
struct has_device_geigs_shift<T, void_t<decltype(std::declval<const T&>().mispec_geigs_shift_solver())>> : std::true_type
{};
// dense matrix in HBM (DenseSymMatProd / DenseGenMatProd)
template <typename T, typename = void>
struct has_device_dense : std::false_type
{};
template <typename T>
struct has_device_dense<T, void_t<decltype(std::declval<const T&>().mispec_dense_matrix())>> : std::true_type
{};
// user operator that works on device pointers:
//   void perform_op_device(const Scalar* x_dev, Scalar* y_dev, void* hip_stream) const
// It ENQUEUES y = Op(x) on hip_stream (a hipStream_t) and returns; the steps of a sweep are enqueued ahead of their execution,
// so x_dev is valid in stream order only (mispec.h, mispec_device_op_fn).
template <typename T, typename = void>
struct has_device_perform_op : std::false_type
{};
template <typename T>
struct has_device_perform_op<T, void_t<decltype(std::declval<const T&>().perform_op_device(
                                    static_cast<const double*>(nullptr), static_cast<double*>(nullptr), static_cast<void*>(nullptr)))>>
    : std::true_type
{};
template <typename T, typename = void>
struct has_device_context : std::false_type
{};
template <typename T>
struct has_device_context<T, void_t<decltype(std::declval<const T&>().mispec_context())>> : std::true_type
{};
// The context a host-pointer operator wants its Krylov basis on: its own, if it names one.
template <typename T>
typename std::enable_if<has_device_context<T>::value, CtxPtr>::type context_of(const T& op)
{
    return borrow_context(op.mispec_context());
}
template <typename T>
typename std::enable_if<!has_device_context<T>::value, CtxPtr>::type context_of(const T&)
{
    return default_context();
}
}  // namespace internal

template <typename OpType>
class Arnoldi
{
public:
    using Scalar = typename OpType::Scalar;

protected:
    static_assert(std::is_same<Scalar, double>::value, "the MI355X path computes in fp64: Scalar must be double");
    using Matrix = DenseMatrix<Scalar>;
    using Vector = DenseVector<Scalar>;

    const OpType& m_op;
    const Index m_n;  // dimension of A
    const Index m_m;  // maximum dimension of the Krylov subspace
    internal::CtxPtr m_ctx;
    std::shared_ptr<mispec_fac> m_fac;

    // perform_op of a user operator, called back from inside the library with pinned host buffers.
    static int call_user_op(void* user, const double* x_in, double* y_out)
    {
        try
        {
            static_cast<const OpType*>(user)->perform_op(x_in, y_out);
            return 0;
        }
        catch (...)
        {
            return 1;
        }
    }

    // perform_op_device of a user operator: device pointers, work enqueued on the factorisation's stream
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

    template <typename T = OpType>
    typename std::enable_if<internal::has_device_dense<T>::value>::type bind(bool symmetric)
    {
        m_ctx = internal::borrow_context(m_op.mispec_context());
        mispec_fac* raw = nullptr;
        internal::check(mispec_fac_create_dense(m_ctx.get(), m_op.mispec_dense_matrix(), static_cast<int>(m_m), symmetric ? 1 : 0, &raw));
        m_fac = std::shared_ptr<mispec_fac>(raw, [](mispec_fac* p) { (void) mispec_fac_destroy(p); });
    }
    template <typename T = OpType>
    typename std::enable_if<internal::has_device_perform_op<T>::value && !internal::has_device_dense<T>::value &&
                            !internal::has_device_matrix<T>::value && !internal::has_device_solver<T>::value>::type
    bind(bool symmetric)
    {
        m_ctx = internal::context_of(m_op);
        mispec_fac* raw = nullptr;
        internal::check(mispec_fac_create_device_op(m_ctx.get(), &Arnoldi::call_user_device_op, const_cast<OpType*>(&m_op), m_n,
                                                    static_cast<int>(m_m), symmetric ? 1 : 0, &raw));
        m_fac = std::shared_ptr<mispec_fac>(raw, [](mispec_fac* p) { (void) mispec_fac_destroy(p); });
    }
    template <typename T = OpType>
    typename std::enable_if<internal::has_device_matrix<T>::value>::type bind(bool symmetric)
    {
        m_ctx = internal::borrow_context(m_op.mispec_context());
        mispec_fac* raw = nullptr;
        internal::check(mispec_fac_create(m_ctx.get(), m_op.mispec_matrix(), nullptr, nullptr, m_n, static_cast<int>(m_m),
                                          symmetric ? 1 : 0, &raw));
        m_fac = std::shared_ptr<mispec_fac>(raw, [](mispec_fac* p) { (void) mispec_fac_destroy(p); });
    }
    // y = A2 (A x) with both factors in HBM (contrib/PartialSVDSolver.h's A'A / AA' operators)
    template <typename T = OpType>
    typename std::enable_if<internal::has_device_product<T>::value>::type bind(bool symmetric)
    {
        if (!symmetric)
            throw std::invalid_argument("Arnoldi: product operators are symmetric (Lanczos) only");
        m_ctx = internal::borrow_context(m_op.mispec_context());
        mispec_fac* raw = nullptr;
        internal::check(mispec_fac_create_product(m_ctx.get(), m_op.mispec_product_first(), m_op.mispec_product_second(),
                                                  static_cast<int>(m_m), &raw));
        m_fac = std::shared_ptr<mispec_fac>(raw, [](mispec_fac* p) { (void) mispec_fac_destroy(p); });
    }
    // generalized problem, regular-inverse mode: y = B^{-1}(A x) with B-inner products, A and B on the device
    template <typename T = OpType>
    typename std::enable_if<internal::has_device_geigs<T>::value>::type bind(bool symmetric)
    {
        if (!symmetric)
            throw std::invalid_argument("Arnoldi: the generalized regular-inverse operator is symmetric (Lanczos) only");
        m_ctx = internal::borrow_context(m_op.mispec_context());
        mispec_fac* raw = nullptr;
        internal::check(mispec_fac_create_geigs_reginv(m_ctx.get(), m_op.mispec_geigs_matrix(), m_op.mispec_geigs_b_operator(),
                                                       static_cast<int>(m_m), &raw));
        m_fac = std::shared_ptr<mispec_fac>(raw, [](mispec_fac* p) { (void) mispec_fac_destroy(p); });
    }
    // generalized problem, Cholesky mode: y = L^{-1} A L^{-T} x, plain inner products
    template <typename T = OpType>
    typename std::enable_if<internal::has_device_geigs_cholesky<T>::value>::type bind(bool symmetric)
    {
        if (!symmetric)
            throw std::invalid_argument("Arnoldi: the generalized Cholesky operator is symmetric (Lanczos) only");
        m_ctx = internal::borrow_context(m_op.mispec_context());
        mispec_fac* raw = nullptr;
        internal::check(mispec_fac_create_geigs_cholesky(m_ctx.get(), m_op.mispec_geigs_cholesky_matrix(),
                                                         m_op.mispec_geigs_cholesky_factor(), static_cast<int>(m_m), &raw));
        m_fac = std::shared_ptr<mispec_fac>(raw, [](mispec_fac* p) { (void) mispec_fac_destroy(p); });
    }
    // generalized problem, shift modes: y = (A - sigma B)^{-1} B x (Cayley: x + 2 sigma * that), B-inner products
    template <typename T = OpType>
    typename std::enable_if<internal::has_device_geigs_shift<T>::value>::type bind(bool symmetric)
    {
        if (!symmetric)
            throw std::invalid_argument("Arnoldi: the generalized shift operators are symmetric (Lanczos) only");
        m_ctx = internal::borrow_context(m_op.mispec_context());
        mispec_fac* raw = nullptr;
        internal::check(mispec_fac_create_geigs_shift(m_ctx.get(), m_op.mispec_geigs_shift_solver(), m_op.mispec_geigs_shift_b(),
                                                      m_op.mispec_geigs_shift_cayley() ? 1 : 0, m_op.mispec_geigs_shift_sigma(),
                                                      static_cast<int>(m_m), &raw));
        m_fac = std::shared_ptr<mispec_fac>(raw, [](mispec_fac* p) { (void) mispec_fac_destroy(p); });
    }
    template <typename T = OpType>
    typename std::enable_if<internal::has_device_solver<T>::value>::type bind(bool symmetric)
    {
        m_ctx = internal::borrow_context(m_op.mispec_context());
        mispec_fac* raw = nullptr;
        internal::check(
            mispec_fac_create_shiftsolve(m_ctx.get(), m_op.mispec_solver(), static_cast<int>(m_m), symmetric ? 1 : 0, &raw));
        m_fac = std::shared_ptr<mispec_fac>(raw, [](mispec_fac* p) { (void) mispec_fac_destroy(p); });
    }
    template <typename T = OpType>
    typename std::enable_if<!internal::has_device_matrix<T>::value && !internal::has_device_solver<T>::value &&
                            !internal::has_device_product<T>::value && !internal::has_device_geigs<T>::value &&
                            !internal::has_device_geigs_shift<T>::value && !internal::has_device_geigs_cholesky<T>::value &&
                            !internal::has_device_dense<T>::value && !internal::has_device_perform_op<T>::value>::type
    bind(bool symmetric)
    {
        m_ctx = internal::context_of(m_op);
        mispec_fac* raw = nullptr;
        internal::check(mispec_fac_create(m_ctx.get(), nullptr, &Arnoldi::call_user_op, const_cast<OpType*>(&m_op), m_n,
                                          static_cast<int>(m_m), symmetric ? 1 : 0, &raw));
        m_fac = std::shared_ptr<mispec_fac>(raw, [](mispec_fac* p) { (void) mispec_fac_destroy(p); });
    }

    Arnoldi(const OpType& op, Index m, bool symmetric) : m_op(op), m_n(op.rows()), m_m(m) { bind(symmetric); }

public:
    // General (non-symmetric) factorisation: full Gram-Schmidt against V every step (Arnoldi.h:198-295).
    Arnoldi(const OpType& op, Index m) : Arnoldi(op, m, false) {}
    virtual ~Arnoldi() {}

    // v <- A v0 / |A v0|, H(0,0), f  (Arnoldi.h:136-195).  v0 has n entries, host memory.
    void init(const Scalar* v0, Index& op_counter)
    {
        std::int64_t cnt = op_counter;
        internal::check(mispec_fac_init(m_fac.get(), v0, &cnt));
        op_counter = static_cast<Index>(cnt);
    }
    // Same with v0 = SimpleRandom(seed) generated on the device (HermEigsBase.h:337-342 uses seed 0).
    void init_random(unsigned long seed, Index& op_counter)
    {
        std::int64_t cnt = op_counter;
        internal::check(mispec_fac_init_random(m_fac.get(), seed, &cnt));
        op_counter = static_cast<Index>(cnt);
    }

    // Extend the k-step factorisation to to_m steps.
    virtual void factorize_from(Index from_k, Index to_m, Index& op_counter)
    {
        std::int64_t cnt = op_counter;
        internal::check(mispec_fac_factorize(m_fac.get(), static_cast<int>(from_k), static_cast<int>(to_m), &cnt));
        op_counter = static_cast<Index>(cnt);
    }

    Index subspace_dim() const { return mispec_fac_subspace_dim(m_fac.get()); }
    Scalar f_norm() const
    {
        double b = 0;
        internal::check(mispec_fac_f_norm(m_fac.get(), &b));
        return b;
    }
    Matrix matrix_H() const
    {
        Matrix H(m_m, m_m);
        internal::check(mispec_fac_get_H(m_fac.get(), H.data()));
        return H;
    }
    // Downloads: V is local_rows x m (the whole matrix on an unsharded run).
    Matrix matrix_V() const
    {
        Matrix V(local_rows(), m_m);
        internal::check(mispec_fac_get_V(m_fac.get(), static_cast<int>(m_m), V.data()));
        return V;
    }
    Vector vector_f() const
    {
        Vector f(local_rows());
        internal::check(mispec_fac_get_f(m_fac.get(), f.data()));
        return f;
    }
    Index local_rows() const { return static_cast<Index>(mispec_fac_local_rows(m_fac.get())); }

    // V[:, :k+1] <- V Q and the matching update of f (Arnoldi.h:320-340) after the caller compressed H on the host.
    void compress_V(const Matrix& Q, const Matrix& H_compressed, Index new_k)
    {
        internal::check(mispec_fac_compress_V(m_fac.get(), Q.data(), H_compressed.data(), static_cast<int>(new_k)));
    }

    // The shift list of one general restart applied on the device (k_hess_restart), then V <- V Q and the update of f.
    void restart_gen(const std::vector<int>& kind, const std::vector<double>& a, const std::vector<double>& b, Index new_k)
    {
        internal::check(mispec_fac_restart_gen(m_fac.get(), kind.data(), a.data(), b.data(), static_cast<int>(kind.size()),
                                               static_cast<int>(new_k)));
    }

    // X = V * Y, Y is m x ncols (HermEigsBase.h:467 / GenEigsBase.h:600); returned on the host.
    Matrix ritz_vectors(const Matrix& Y) const
    {
        Matrix X(local_rows(), Y.cols());
        if (Y.cols() > 0)
            internal::check(mispec_fac_ritz_vectors(m_fac.get(), Y.data(), static_cast<int>(Y.cols()), X.data(), nullptr));
        return X;
    }

    // The same product written into caller-provided host memory (local_rows() x Y.cols(), column-major, no intermediate matrix)
    void ritz_vectors_into(const Matrix& Y, Scalar* X_host) const
    {
        if (Y.cols() > 0)
            internal::check(mispec_fac_ritz_vectors(m_fac.get(), Y.data(), static_cast<int>(Y.cols()), X_host, nullptr));
    }

    // Same product left in HBM only (valid until the next call); returns the device pointer, leading dimension in *ld.
    const Scalar* ritz_vectors_device(const Matrix& Y, Index* ld = nullptr) const
    {
        const double* X = nullptr;
        if (Y.cols() > 0)
            internal::check(mispec_fac_ritz_vectors(m_fac.get(), Y.data(), static_cast<int>(Y.cols()), nullptr, &X));
        if (ld)
        {
            std::int64_t l = 0;
            (void) mispec_fac_V_dev(m_fac.get(), &l);
            *ld = static_cast<Index>(l);
        }
        return X;
    }

    mispec_fac* handle() const { return m_fac.get(); }
    mispec_ctx* context() const { return m_ctx.get(); }
};

// ---- complex scalars (outside the hot path of SURVEY.md section 8) ---------------------------------------------------------
namespace internal {
template <typename T, typename = void>
struct has_device_zdense : std::false_type
{};
template <typename T>
struct has_device_zdense<T, void_t<decltype(std::declval<const T&>().mispec_zdense_matrix())>> : std::true_type
{};

// The factorisation for Scalar = std::complex<double>: basis, residual and (for the dense operators) the matrix in HBM behind a
// mispec_zfac handle (include/mispec_extras.h, csrc/zfac.hip); host-driven steps in the reference's order (Arnoldi.h:136-295,
// Lanczos.h:62-187).  Any other operator is used through perform_op on host pointers.
template <typename OpType>
class ComplexArnoldi
{
public:
    using Scalar = typename OpType::Scalar;

protected:
    static_assert(std::is_same<Scalar, std::complex<double>>::value, "complex factorisation: Scalar must be std::complex<double>");
    using Matrix = DenseMatrix<Scalar>;
    using Vector = DenseVector<Scalar>;

    const OpType& m_op;
    const Index m_n;
    const Index m_m;
    CtxPtr m_ctx;
    std::shared_ptr<mispec_zfac> m_zfac;

    static int call_user_op(void* user, const double* x_in, double* y_out)
    {
        try
        {
            static_cast<const OpType*>(user)->perform_op(reinterpret_cast<const Scalar*>(x_in), reinterpret_cast<Scalar*>(y_out));
            return 0;
        }
        catch (...)
        {
            return 1;
        }
    }
    template <typename T = OpType>
    typename std::enable_if<has_device_zdense<T>::value>::type bind(bool hermitian)
    {
        m_ctx = borrow_context(m_op.mispec_context());
        mispec_zfac* raw = nullptr;
        check(mispec_zfac_create_dense(m_ctx.get(), m_op.mispec_zdense_matrix(), static_cast<int>(m_m), hermitian ? 1 : 0, &raw));
        m_zfac = std::shared_ptr<mispec_zfac>(raw, [](mispec_zfac* p) { (void) mispec_zfac_destroy(p); });
    }
    template <typename T = OpType>
    typename std::enable_if<!has_device_zdense<T>::value>::type bind(bool hermitian)
    {
        m_ctx = context_of(m_op);
        mispec_zfac* raw = nullptr;
        check(mispec_zfac_create_op(m_ctx.get(), &ComplexArnoldi::call_user_op, const_cast<OpType*>(&m_op), m_n, static_cast<int>(m_m),
                                    hermitian ? 1 : 0, &raw));
        m_zfac = std::shared_ptr<mispec_zfac>(raw, [](mispec_zfac* p) { (void) mispec_zfac_destroy(p); });
    }

    ComplexArnoldi(const OpType& op, Index m, bool hermitian) : m_op(op), m_n(op.rows()), m_m(m) { bind(hermitian); }

public:
    ComplexArnoldi(const OpType& op, Index m) : ComplexArnoldi(op, m, false) {}
    virtual ~ComplexArnoldi() {}

    void init(const Scalar* v0, Index& op_counter)
    {
        std::int64_t cnt = op_counter;
        check(mispec_zfac_init(m_zfac.get(), reinterpret_cast<const double*>(v0), &cnt));
        op_counter = static_cast<Index>(cnt);
    }
    virtual void factorize_from(Index from_k, Index to_m, Index& op_counter)
    {
        std::int64_t cnt = op_counter;
        check(mispec_zfac_factorize(m_zfac.get(), static_cast<int>(from_k), static_cast<int>(to_m), &cnt));
        op_counter = static_cast<Index>(cnt);
    }
    Index subspace_dim() const { return mispec_zfac_subspace_dim(m_zfac.get()); }
    double f_norm() const
    {
        double b = 0;
        check(mispec_zfac_f_norm(m_zfac.get(), &b));
        return b;
    }
    Matrix matrix_H() const
    {
        Matrix H(m_m, m_m);
        check(mispec_zfac_get_H(m_zfac.get(), reinterpret_cast<double*>(H.data())));
        return H;
    }
    Matrix matrix_V() const
    {
        Matrix V(m_n, m_m);
        check(mispec_zfac_get_V(m_zfac.get(), static_cast<int>(m_m), reinterpret_cast<double*>(V.data())));
        return V;
    }
    Vector vector_f() const
    {
        Vector f(m_n);
        check(mispec_zfac_get_f(m_zfac.get(), reinterpret_cast<double*>(f.data())));
        return f;
    }
    mispec_zfac* handle() const { return m_zfac.get(); }
    mispec_ctx* context() const { return m_ctx.get(); }
};

// double -> the device factorisation above, std::complex<double> -> ComplexArnoldi
template <typename OpType, typename Scalar = typename OpType::Scalar>
struct factorisation_of
{
    using type = Arnoldi<OpType>;
};
template <typename OpType>
struct factorisation_of<OpType, std::complex<double>>
{
    using type = ComplexArnoldi<OpType>;
};
}  // namespace internal

// The reference's spelling, Arnoldi<ArnoldiOp<OpType, IdentityBOp>> (Arnoldi.h:32-343 takes the wrapper, test/Arnoldi.cpp:94-138):
// the factorisation of the wrapped operator; init() also takes the start vector as a vector object (Arnoldi.h:136).
template <typename OpType>
class Arnoldi<ArnoldiOp<OpType, IdentityBOp>> : public internal::factorisation_of<OpType>::type
{
    using Base = typename internal::factorisation_of<OpType>::type;

protected:
    Arnoldi(const ArnoldiOp<OpType, IdentityBOp>& op, Index m, bool symmetric) : Base(op.op(), m, symmetric) {}

public:
    using Scalar = typename OpType::Scalar;

    Arnoldi(const ArnoldiOp<OpType, IdentityBOp>& op, Index m) : Base(op.op(), m, false) {}

    using Base::init;
    template <typename VectorType>
    auto init(const VectorType& v0, Index& op_counter) -> decltype(v0.data(), void())
    {
        if (static_cast<Index>(v0.size()) != this->m_n)
            throw std::invalid_argument("Arnoldi: the initial vector must have as many entries as the operator has rows");
        Base::init(v0.data(), op_counter);
    }
};

}  // namespace Spectra

#endif
