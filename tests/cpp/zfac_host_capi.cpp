// TEST INFRASTRUCTURE — not part of libmispec.so.  The mispec_zdense / mispec_zfac entry points of include/mispec_extras.h built
// over a plain HOST backend of spectra_amd/csrc/zfac_flow.hpp (the control flow and the Hermitian mirroring are the library's own
// source; only the vector primitives differ: loops here, HIP kernels in csrc/zfac.hip).  tests/test_host_zfac.py loads it to run
// the checks of tests/zfac_checks.py where there is no GPU; the product has no such path.
//   g++ -std=c++17 -O2 -shared -fPIC -I spectra_amd/csrc tests/cpp/zfac_host_capi.cpp -o <tmp>/libzfac_host.so
#include "zfac_host_backend.hpp"

#include <memory>

struct zfac
{
    HostBackend be;
    std::unique_ptr<mispec::ZFacFlow<HostBackend>> flow;
};

template <typename F>
static int guarded(F&& f)
{
    try
    {
        f();
        return 0;
    }
    catch (const std::invalid_argument&)
    {
        return -1;
    }
    catch (const std::logic_error&)
    {
        return -2;
    }
    catch (...)
    {
        return -3;
    }
}

extern "C" {
int mispec_zdense_upload(void*, int64_t rows, int64_t cols, const double* data, int64_t ld, int row_major, char uplo, zdense** out)
{
    return guarded([&] {
        if (uplo && rows != cols)
            throw std::invalid_argument("square");
        std::unique_ptr<zdense> D(new zdense);
        D->rows = rows;
        D->cols = cols;
        D->a.resize(size_t(rows) * size_t(cols));
        mispec::zdense_expand(rows, cols, reinterpret_cast<const cd*>(data), ld, row_major != 0, uplo, D->a.data());
        *out = D.release();
    });
}
int mispec_zdense_destroy(zdense* D)
{
    delete D;
    return 0;
}
int64_t mispec_zdense_rows(const zdense* D) { return D->rows; }
int64_t mispec_zdense_cols(const zdense* D) { return D->cols; }
int mispec_zdense_gemv_host(const zdense* D, const double* x, double* y)
{
    const cd* xv = reinterpret_cast<const cd*>(x);
    cd* yv = reinterpret_cast<cd*>(y);
    for (int64_t i = 0; i < D->rows; i++)
    {
        cd acc(0.0);
        for (int64_t j = 0; j < D->cols; j++)
            acc += D->a[size_t(j * D->rows + i)] * xv[j];
        yv[i] = acc;
    }
    return 0;
}
int mispec_zdense_coeff(const zdense* D, int64_t i, int64_t j, double* out)
{
    return guarded([&] {
        if (i < 0 || i >= D->rows || j < 0 || j >= D->cols)
            throw std::invalid_argument("range");
        out[0] = D->a[size_t(j * D->rows + i)].real();
        out[1] = D->a[size_t(j * D->rows + i)].imag();
    });
}
int mispec_zfac_create_dense(void*, const zdense* D, int ncv, int hermitian, zfac** out)
{
    return guarded([&] {
        std::unique_ptr<zfac> F(new zfac);
        F->be.n = D->rows;
        F->be.dense = D;
        F->flow.reset(new mispec::ZFacFlow<HostBackend>(F->be, D->rows, ncv, hermitian != 0));
        *out = F.release();
    });
}
int mispec_zfac_create_op(void*, zop_fn op, void* user, int64_t n, int ncv, int hermitian, zfac** out)
{
    return guarded([&] {
        std::unique_ptr<zfac> F(new zfac);
        F->be.n = n;
        F->be.op = op;
        F->be.user = user;
        F->flow.reset(new mispec::ZFacFlow<HostBackend>(F->be, n, ncv, hermitian != 0));
        *out = F.release();
    });
}
int mispec_zfac_destroy(zfac* F)
{
    delete F;
    return 0;
}
int mispec_zfac_init(zfac* F, const double* v0, int64_t* cnt)
{
    return guarded([&] { F->flow->init(reinterpret_cast<const cd*>(v0), *cnt); });
}
int mispec_zfac_factorize(zfac* F, int from_k, int to_m, int64_t* cnt)
{
    return guarded([&] { F->flow->factorize_from(from_k, to_m, *cnt); });
}
int mispec_zfac_subspace_dim(const zfac* F) { return F->flow->subspace_dim(); }
int mispec_zfac_f_norm(const zfac* F, double* out)
{
    *out = F->flow->f_norm();
    return 0;
}
int mispec_zfac_get_H(const zfac* F, double* H)
{
    const std::vector<cd>& h = F->flow->matrix_H();
    std::copy(h.begin(), h.end(), reinterpret_cast<cd*>(H));
    return 0;
}
int mispec_zfac_get_V(const zfac* F, int ncols, double* V)
{
    F->flow->get_V(reinterpret_cast<cd*>(V), ncols);
    return 0;
}
int mispec_zfac_get_f(const zfac* F, double* f)
{
    F->flow->get_f(reinterpret_cast<cd*>(f));
    return 0;
}
}
