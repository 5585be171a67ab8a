// Truncated SVD of a sparse matrix through the symmetric eigen solver — counterpart of the reference's
// contrib/PartialSVDSolver.h:16-213 (same classes, members and defaults): the k largest singular
// triplets of an m x n matrix M from the k largest eigenpairs of M'M (tall, m > n) or MM' (wide).
//
// MI355X form: M and M' are both copied to HBM as CSR once (the CSC arrays of M are the CSR arrays of
// M'), and the operator y = M'(M x) / M(M' x) is two chained CSR-stream SpMVs inside the device Lanczos
// loop (mispec_fac_create_product); the Krylov basis never leaves the GPU.  perform_op() keeps the
// reference's host-pointer contract for callers that use the operator classes directly.
// Dense MatrixType (the reference's default template argument) is outside this library's scope: pass a
// Spectra::SparseView<double> or, with Eigen on the include path, an Eigen::SparseMatrix<double>.
#ifndef MISPEC_SPECTRA_PARTIAL_SVD_SOLVER_H
#define MISPEC_SPECTRA_PARTIAL_SVD_SOLVER_H

#include <algorithm>
#include <cmath>
#include <memory>
#include <vector>

#include "../SymEigsSolver.h"

namespace Spectra {

namespace internal {

// M (rows x cols) and M' as device CSR matrices built from one compressed host matrix.
struct DeviceCsrPair
{
    CtxPtr ctx;
    std::shared_ptr<mispec_csr> mat, mat_t;

    DeviceCsrPair(const SparseView<double, int>& v, CtxPtr c) : ctx(c ? c : default_context())
    {
        mispec_csr *a = nullptr, *at = nullptr;
        if (v.row_major)
        {
            check(mispec_csr_upload(ctx.get(), v.rows, v.cols, v.outer, v.inner, v.values, &a));
            mat.reset(a, [](mispec_csr* p) { (void) mispec_csr_destroy(p); });
            check(mispec_csr_from_csc(ctx.get(), v.cols, v.rows, v.outer, v.inner, v.values, &at));
            mat_t.reset(at, [](mispec_csr* p) { (void) mispec_csr_destroy(p); });
        }
        else
        {
            check(mispec_csr_from_csc(ctx.get(), v.rows, v.cols, v.outer, v.inner, v.values, &a));
            mat.reset(a, [](mispec_csr* p) { (void) mispec_csr_destroy(p); });
            check(mispec_csr_upload(ctx.get(), v.cols, v.rows, v.outer, v.inner, v.values, &at));
            mat_t.reset(at, [](mispec_csr* p) { (void) mispec_csr_destroy(p); });
        }
    }
};

inline SparseView<double, int> svd_view(const SparseView<double, int>& v) { return v; }
#ifdef MISPEC_HAVE_EIGEN
template <int Flags>
SparseView<double, int> svd_view(const Eigen::SparseMatrix<double, Flags, int>& m)
{
    if (!m.isCompressed())
        throw std::invalid_argument("PartialSVDSolver: the sparse matrix must be compressed (call makeCompressed())");
    SparseView<double, int> v;
    v.rows = m.rows();
    v.cols = m.cols();
    v.outer = m.outerIndexPtr();
    v.inner = m.innerIndexPtr();
    v.values = m.valuePtr();
    v.row_major = (Flags & Eigen::RowMajorBit) != 0;
    return v;
}
#endif

// A dense matrix (the reference's PartialSVDSolver<Eigen::MatrixXd>, test/SVD.cpp:72-103): the device operators are sparse
// ones, so every entry is stored — a compressed column-major image built here and uploaded like any other sparse matrix.
struct DenseAsCsc
{
    std::vector<int> outer, inner;
    std::vector<double> values;
    Index rows = 0, cols = 0;
    DenseAsCsc(const DenseView<double>& m) : rows(m.rows), cols(m.cols)
    {
        outer.resize(static_cast<std::size_t>(cols) + 1);
        inner.reserve(static_cast<std::size_t>(rows * cols));
        values.reserve(static_cast<std::size_t>(rows * cols));
        for (Index j = 0; j < cols; j++)
        {
            outer[static_cast<std::size_t>(j)] = static_cast<int>(values.size());
            for (Index i = 0; i < rows; i++)
            {
                inner.push_back(static_cast<int>(i));
                values.push_back(m.row_major ? m.data[i * m.ld + j] : m.data[j * m.ld + i]);
            }
        }
        outer[static_cast<std::size_t>(cols)] = static_cast<int>(values.size());
    }
    SparseView<double, int> view() const
    {
        SparseView<double, int> v;
        v.rows = rows;
        v.cols = cols;
        v.outer = outer.data();
        v.inner = inner.data();
        v.values = values.data();
        v.row_major = false;
        return v;
    }
};

// host matrix of any supported kind -> the device pair
template <typename MatrixType>
std::shared_ptr<DeviceCsrPair> make_device_pair(const MatrixType& mat, CtxPtr ctx)
{
    return std::make_shared<DeviceCsrPair>(svd_view(mat), ctx);
}
inline std::shared_ptr<DeviceCsrPair> make_device_pair(const DenseView<double>& mat, CtxPtr ctx)
{
    const DenseAsCsc image(mat);
    return std::make_shared<DeviceCsrPair>(image.view(), ctx);
}
inline std::shared_ptr<DeviceCsrPair> make_device_pair(const PlainMatrix<double>& mat, CtxPtr ctx)
{
    return make_device_pair(DenseView<double>(mat), ctx);
}
#ifdef MISPEC_HAVE_EIGEN
template <int Options>
std::shared_ptr<DeviceCsrPair> make_device_pair(const Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Options>& mat, CtxPtr ctx)
{
    return make_device_pair(DenseView<double>(mat), ctx);
}
#endif

}  // namespace internal

// Common interface of the two operators (reference: contrib/PartialSVDSolver.h:16-34) plus the hooks through which the
// device Lanczos finds the two matrices.
template <typename Scalar_>
class SVDMatOp
{
public:
    using Scalar = Scalar_;
    static_assert(std::is_same<Scalar_, double>::value, "the MI355X path computes in fp64: Scalar must be double");

    virtual Index rows() const = 0;
    virtual Index cols() const = 0;
    // y_out = A' * A * x_in or y_out = A * A' * x_in   (host pointers)
    virtual void perform_op(const Scalar* x_in, Scalar* y_out) const = 0;

    virtual mispec_ctx* mispec_context() const = 0;
    virtual const mispec_csr* mispec_product_first() const = 0;   // applied to x
    virtual const mispec_csr* mispec_product_second() const = 0;  // applied to the result

    virtual ~SVDMatOp() {}
};

// m > n: the Gram operator x -> A'(A x) of size n   (reference: contrib/PartialSVDSolver.h:36-72)
template <typename Scalar, typename MatrixType>
class SVDTallMatOp : public SVDMatOp<Scalar>
{
    std::shared_ptr<internal::DeviceCsrPair> m_dev;
    const Index m_dim;
    mutable std::vector<Scalar> m_cache;

public:
    SVDTallMatOp(const MatrixType& mat, internal::CtxPtr ctx = internal::CtxPtr()) :
        SVDTallMatOp(internal::make_device_pair(mat, ctx))
    {}
    explicit SVDTallMatOp(std::shared_ptr<internal::DeviceCsrPair> dev) :
        m_dev(dev),
        m_dim((std::min)(mispec_csr_rows(dev->mat.get()), mispec_csr_cols(dev->mat.get()))),
        m_cache(static_cast<std::size_t>(mispec_csr_rows(dev->mat.get())))
    {}

    // dimension of the Gram operator
    Index rows() const override { return m_dim; }
    Index cols() const override { return m_dim; }

    // host-pointer form: first A x into the cache, then A' applied to it
    void perform_op(const Scalar* x_in, Scalar* y_out) const override
    {
        internal::check(mispec_spmv_host(m_dev->mat.get(), x_in, m_cache.data()));
        internal::check(mispec_spmv_host(m_dev->mat_t.get(), m_cache.data(), y_out));
    }

    mispec_ctx* mispec_context() const override { return m_dev->ctx.get(); }
    const mispec_csr* mispec_product_first() const override { return m_dev->mat.get(); }
    const mispec_csr* mispec_product_second() const override { return m_dev->mat_t.get(); }
};

// m <= n: the operator x -> A(A' x) of size m   (reference: contrib/PartialSVDSolver.h:74-110)
template <typename Scalar, typename MatrixType>
class SVDWideMatOp : public SVDMatOp<Scalar>
{
    std::shared_ptr<internal::DeviceCsrPair> m_dev;
    const Index m_dim;
    mutable std::vector<Scalar> m_cache;

public:
    SVDWideMatOp(const MatrixType& mat, internal::CtxPtr ctx = internal::CtxPtr()) :
        SVDWideMatOp(internal::make_device_pair(mat, ctx))
    {}
    explicit SVDWideMatOp(std::shared_ptr<internal::DeviceCsrPair> dev) :
        m_dev(dev),
        m_dim((std::min)(mispec_csr_rows(dev->mat.get()), mispec_csr_cols(dev->mat.get()))),
        m_cache(static_cast<std::size_t>(mispec_csr_cols(dev->mat.get())))
    {}

    // dimension of that operator
    Index rows() const override { return m_dim; }
    Index cols() const override { return m_dim; }

    // host-pointer form: A' x into the cache, then A applied to it
    void perform_op(const Scalar* x_in, Scalar* y_out) const override
    {
        internal::check(mispec_spmv_host(m_dev->mat_t.get(), x_in, m_cache.data()));
        internal::check(mispec_spmv_host(m_dev->mat.get(), m_cache.data(), y_out));
    }

    mispec_ctx* mispec_context() const override { return m_dev->ctx.get(); }
    const mispec_csr* mispec_product_first() const override { return m_dev->mat_t.get(); }
    const mispec_csr* mispec_product_second() const override { return m_dev->mat.get(); }
};

// Partial SVD solver (contrib/PartialSVDSolver.h:112-209)
template <typename MatrixType = SparseView<double, int>>
class PartialSVDSolver
{
private:
    using Scalar = double;
    using Matrix = DenseMatrix<Scalar>;
    using Vector = DenseVector<Scalar>;

    std::shared_ptr<internal::DeviceCsrPair> m_dev;
    const Index m_m;
    const Index m_n;
    std::unique_ptr<SVDMatOp<Scalar>> m_op;
    std::unique_ptr<SymEigsSolver<SVDMatOp<Scalar>>> m_eigs;
    Index m_nconv = 0;
    Matrix m_evecs;

    // mat * (evecs[:, :k] ./ sqrt(evals[:k]))  on the device matrix `A` (rows x inner)
    Matrix scaled_product(const mispec_csr* A, Index k) const
    {
        const Vector evals = m_eigs->eigenvalues();
        Matrix scaled(m_evecs.rows(), k);
        for (Index j = 0; j < k; j++)
        {
            const Scalar s = std::sqrt(evals[j]);
            for (Index i = 0; i < m_evecs.rows(); i++)
                scaled(i, j) = m_evecs(i, j) / s;
        }
        Matrix res(static_cast<Index>(mispec_csr_rows(A)), k);
        if (k > 0)
            internal::check(
                mispec_spmm_host(A, scaled.data(), scaled.rows(), static_cast<int>(k), res.data(), res.rows()));
        return res;
    }

public:
    // Constructor
    PartialSVDSolver(const MatrixType& mat, Index ncomp, Index ncv, internal::CtxPtr ctx = internal::CtxPtr()) :
        m_dev(internal::make_device_pair(mat, ctx)),
        m_m(static_cast<Index>(mispec_csr_rows(m_dev->mat.get()))),
        m_n(static_cast<Index>(mispec_csr_cols(m_dev->mat.get())))
    {
        // pick the smaller of the two Gram operators
        if (m_m > m_n)
            m_op.reset(new SVDTallMatOp<Scalar, MatrixType>(m_dev));
        else
            m_op.reset(new SVDWideMatOp<Scalar, MatrixType>(m_dev));
        // the symmetric eigen solver that does the work
        m_eigs.reset(new SymEigsSolver<SVDMatOp<Scalar>>(*m_op, ncomp, ncv));
    }

    // runs the eigen solver for the largest eigenvalues of the Gram operator; returns how many converged
    Index compute(Index maxit = 1000, Scalar tol = 1e-10)
    {
        m_eigs->init();
        m_nconv = m_eigs->compute(SortRule::LargestAlge, maxit, tol);
        m_evecs = Matrix();
        return m_nconv;
    }

    CompInfo info() const { return m_eigs->info(); }
    Index num_iterations() const { return m_eigs->num_iterations(); }
    Index num_operations() const { return m_eigs->num_operations(); }

    // sigma_i = sqrt(lambda_i) of the converged pairs, largest first
    Vector singular_values() const
    {
        Vector svals = m_eigs->eigenvalues();
        for (Index i = 0; i < svals.size(); i++)
            svals[i] = std::sqrt(svals[i]);
        return svals;
    }

    // U: the eigenvectors themselves (wide case) or A V / sigma (tall case)
    Matrix matrix_U(Index nu)
    {
        if (m_evecs.cols() < 1)
            m_evecs = m_eigs->eigenvectors();
        nu = (std::min)(nu, m_nconv);
        if (m_m <= m_n)
            return left_cols(nu);
        return scaled_product(m_dev->mat.get(), nu);
    }

    // V: the eigenvectors themselves (tall case) or A' U / sigma (wide case)
    Matrix matrix_V(Index nv)
    {
        if (m_evecs.cols() < 1)
            m_evecs = m_eigs->eigenvectors();
        nv = (std::min)(nv, m_nconv);
        if (m_m > m_n)
            return left_cols(nv);
        return scaled_product(m_dev->mat_t.get(), nv);
    }

private:
    Matrix left_cols(Index k) const
    {
        Matrix res(m_evecs.rows(), k);
        for (Index j = 0; j < k; j++)
            for (Index i = 0; i < m_evecs.rows(); i++)
                res(i, j) = m_evecs(i, j);
        return res;
    }
};

}  // namespace Spectra

#endif  // MISPEC_SPECTRA_PARTIAL_SVD_SOLVER_H
