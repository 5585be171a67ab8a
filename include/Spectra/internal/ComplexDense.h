// Complex dense operators and the complex factorisation handle — OUTSIDE the hot path of SURVEY.md section 8 (the configs are
// real fp64).  The reference's MatOp/DenseGenMatProd.h:27-102 and MatOp/DenseHermMatProd.h are templates over the scalar and its
// test/Arnoldi.cpp:122-158 instantiates Arnoldi / Lanczos over them with std::complex<double>; here the matrix is copied to HBM
// once (mispec_zdense, include/mispec_extras.h) and the factorisation keeps its basis there (mispec_zfac).
#ifndef MISPEC_SPECTRA_COMPLEX_DENSE_H
#define MISPEC_SPECTRA_COMPLEX_DENSE_H

#include <complex>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>

#include "../../mispec_extras.h"
#include "Dense.h"
#include "Device.h"

namespace Spectra {
namespace internal {

// What DenseGenMatProd<std::complex<double>> and DenseHermMatProd<std::complex<double>> share: the device matrix and the
// reference's members rows(), cols(), perform_op(), operator*, operator().
class ComplexDenseOp
{
public:
    using Scalar = std::complex<double>;

protected:
    using Matrix = DenseMatrix<Scalar>;
    CtxPtr m_ctx;
    std::shared_ptr<mispec_zdense> m_mat;

    void ingest(const DenseView<Scalar>& A, bool want_row_major, char uplo, const char* who)
    {
        if (A.row_major != want_row_major)
            throw std::invalid_argument(std::string(who) +
                                        ": the \"Flags\" template parameter does not match the input matrix (ColMajor/RowMajor)");
        if (uplo && A.rows != A.cols)
            throw std::invalid_argument(std::string(who) + ": matrix must be square");
        mispec_zdense* raw = nullptr;
        check(mispec_zdense_upload(m_ctx.get(), A.rows, A.cols, reinterpret_cast<const double*>(A.data), A.ld, A.row_major ? 1 : 0, uplo,
                                   &raw));
        m_mat = std::shared_ptr<mispec_zdense>(raw, [](mispec_zdense* p) { (void) mispec_zdense_destroy(p); });
    }
    explicit ComplexDenseOp(CtxPtr ctx) : m_ctx(ctx ? ctx : default_context()) {}

public:
    Index rows() const { return static_cast<Index>(mispec_zdense_rows(m_mat.get())); }
    Index cols() const { return static_cast<Index>(mispec_zdense_cols(m_mat.get())); }

    // y_out = A * x_in, host pointers
    void perform_op(const Scalar* x_in, Scalar* y_out) const
    {
        check(mispec_zdense_gemv_host(m_mat.get(), reinterpret_cast<const double*>(x_in), reinterpret_cast<double*>(y_out)));
    }
    // Y = A * X, column by column
    Matrix operator*(const Matrix& mat_in) const
    {
        Matrix res(rows(), mat_in.cols());
        for (Index j = 0; j < mat_in.cols(); j++)
            perform_op(mat_in.data() + j * mat_in.rows(), res.data() + j * res.rows());
        return res;
    }
    Scalar operator()(Index i, Index j) const
    {
        double v[2] = {0.0, 0.0};
        check(mispec_zdense_coeff(m_mat.get(), i, j, v));
        return Scalar(v[0], v[1]);
    }

    mispec_ctx* mispec_context() const { return m_ctx.get(); }
    const mispec_zdense* mispec_zdense_matrix() const { return m_mat.get(); }
};

}  // namespace internal
}  // namespace Spectra

#endif
