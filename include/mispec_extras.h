/* =============================================================================
 * mispec_extras.h — C ABI of the components OUTSIDE the hot path of SURVEY.md section 8
 * (BASELINE.json north_star): dense operators, the block Davidson solver, the complex-shift
 * operator / solver and the Buckling / Cayley modes of the generalized shift solver.  They were
 * built in rounds 1-2 and are kept working, but they are not part of the thin shim the hot path
 * needs: include/mispec.h alone is that shim.
 * Where they live (round 6): the Davidson solver and the complex factorisation (mispec_davidson_*,
 * mispec_zdense_*, mispec_zfac_*: their own kernels, nothing in the hot path calls them) are in a
 * library of their own, spectra_amd/libmispec_extras.so, built on libmispec.so — link -lmispec_extras
 * -lmispec.  The dense operators and the shift variants share objects with the hot path (the dense
 * GEMV applies the last level of the banded shift solve) and stay in libmispec.so.
 * Conventions (handles, error codes, ownership) as in mispec.h.
 * ============================================================================= */
#ifndef MISPEC_EXTRAS_H
#define MISPEC_EXTRAS_H

#include "mispec.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------
 * Dense operators — replace MatOp/DenseSymMatProd.h:28-105 and MatOp/DenseGenMatProd.h:27-102
 * (y = mat.selfadjointView<Uplo>() * x and y = mat * x).  The matrix is copied to HBM once, row-major.
 * uplo = 'L' / 'U': symmetric, only that triangle of the input is read (and mirrored); 0: general.
 * data_host is rows x cols with leading dimension ld_host, column-major unless row_major != 0.
 * ------------------------------------------------------------------------- */
typedef struct mispec_dense mispec_dense;
int mispec_dense_upload(mispec_ctx* ctx, int64_t rows, int64_t cols, const double* data_host, int64_t ld_host, int row_major,
                        char uplo, mispec_dense** out);
int mispec_dense_destroy(mispec_dense* D);
int64_t mispec_dense_rows(const mispec_dense* D);
int64_t mispec_dense_cols(const mispec_dense* D);
int mispec_dense_gemv(const mispec_dense* D, const double* x_dev, double* y_dev);        /* device pointers */
int mispec_dense_gemv_host(const mispec_dense* D, const double* x_host, double* y_host); /* literal perform_op */
/* Y = D X for a host block of k columns (operator*, DenseSymMatProd.h:93-96) and D(i,j) (operator(), :101-104) */
int mispec_dense_gemm_host(const mispec_dense* D, const double* X_host, int64_t ldx, int k, double* Y_host, int64_t ldy);
int mispec_dense_coeff(const mispec_dense* D, int64_t i, int64_t j, double* out);
/* average ms of `reps` back-to-back GEMV launches (benchmark helper, like mispec_spmv_time) */
int mispec_dense_gemv_time(const mispec_dense* D, const double* x_dev, double* y_dev, int reps, float* ms_per_launch);
/* Factorisation whose operator is the dense matrix (must be square). */
int mispec_fac_create_dense(mispec_ctx* ctx, const mispec_dense* D, int ncv, int symmetric, mispec_fac** out);

/* The same with the operator (A - sigma I)^{-1} of a device-resident shift solver (SymEigsShiftSolver path). */

/* ---------------------------------------------------------------------------
 * DavidsonSymEigsSolver — replaces DavidsonSymEigsSolver.h:18-90 + JDSymEigsBase.h:28-187 (block Davidson with the
 * diagonal-preconditioned-residual correction).  The search space, its image under A and the Ritz vectors live in
 * HBM; the projected eigenproblem (<= 256 x 256) is solved on the host.  The search space holds 256 vectors: a larger
 * nvec_max is lowered to 256 - correction size (earlier restarts, same results).
 * Operators: a device CSR matrix, a dense device matrix, or a device-pointer callback plus diag(A).
 * ------------------------------------------------------------------------- */
typedef struct mispec_davidson mispec_davidson;
int mispec_davidson_create(mispec_ctx* ctx, const mispec_csr* A, int64_t nev, int64_t nvec_init, int64_t nvec_max,
                           mispec_davidson** out);
int mispec_davidson_create_dense(mispec_ctx* ctx, const mispec_dense* D, int64_t nev, int64_t nvec_init, int64_t nvec_max,
                                 mispec_davidson** out);
int mispec_davidson_create_device_op(mispec_ctx* ctx, mispec_device_op_fn op, void* op_user, int64_t n, const double* diag_host,
                                     int64_t nev, int64_t nvec_init, int64_t nvec_max, mispec_davidson** out);
int mispec_davidson_destroy(mispec_davidson* S);
/* set_initial_search_space_size / set_max_search_space_size / set_correction_size (JDSymEigsBase.h:86-105); negative = keep */
int mispec_davidson_set_sizes(mispec_davidson* S, int64_t initial_search_space, int64_t max_search_space, int64_t correction);
int mispec_davidson_get_sizes(const mispec_davidson* S, int64_t* initial_search_space, int64_t* max_search_space, int64_t* correction);
/* compute(selection, maxit = 100, tol = 1e-10) (JDSymEigsBase.h:121-128); with guess_host != NULL compute_with_guess
 * (:130-184) on an n x guess_cols column-major block.  *nconv = converged pairs among the first nev. */
int mispec_davidson_compute(mispec_davidson* S, int selection, int64_t maxit, double tol, const double* guess_host,
                            int64_t guess_cols, int64_t ldg, int64_t* nconv);
int mispec_davidson_info(const mispec_davidson* S);               /* CompInfo as int */
int64_t mispec_davidson_num_iterations(const mispec_davidson* S);
int64_t mispec_davidson_num_operations(const mispec_davidson* S); /* matrix-vector products of the last compute() */
int mispec_davidson_eigenvalues(const mispec_davidson* S, double* out_host);                 /* nev values */
int mispec_davidson_eigenvectors(const mispec_davidson* S, double* out_host, int64_t ld);    /* n x nev, column-major */

/* Complex shift for the general operator (mispec_symshift_create_general): afterwards solve = Re((A - sigma I)^{-1} x), the
 * operator of GenEigsComplexShiftSolver (MatOp/SparseGenComplexShiftSolve.h:74-113, DenseGenComplexShiftSolve.h).  n <= 4096. */
int mispec_symshift_set_shift_complex(mispec_symshift* S, double sigmar, double sigmai);

/* The shift modes of the generalized solver (SymGEigsShiftSolver.h:36-207): operator y = (A - sigma B)^{-1} M x in the
 * B-inner product, with S the pencil solver of mispec_symshift_create_pencil (shift already set), B the matrix of the
 * inner product, and M = B (shift-invert and buckling modes) or, with cayley != 0, M = A + sigma B evaluated as
 * x + 2 sigma (A - sigma B)^{-1} B x (SymGEigsCayleyOp.h:88-99). */
int mispec_fac_create_geigs_shift(mispec_ctx* ctx, const mispec_symshift* S, const mispec_csr* B, int cayley, double sigma, int ncv,
                                  mispec_fac** out);

int mispec_symeigs_create_dense(mispec_ctx* ctx, const mispec_dense* D, int64_t nev, int64_t ncv, mispec_symeigs** out);

/* SymGEigsShiftSolver<SymShiftInvert, SparseSymMatProd, mode>: mode 0 = ShiftInvert (lambda = 1/nu + sigma),
 * 1 = Buckling (lambda = sigma nu / (nu - 1); S built from (K, KG), B = K), 2 = Cayley (lambda = sigma (nu+1)/(nu-1)).
 * Calls set_shift(sigma) on S. */
int mispec_symeigs_create_geigs_shift(mispec_ctx* ctx, mispec_symshift* S, const mispec_csr* B, int mode, int64_t nev, int64_t ncv,
                                      double sigma, mispec_symeigs** out);

/* GenEigsComplexShiftSolver (GenEigsComplexShiftSolver.h:20-150): Arnoldi on x -> Re((A - sigma I)^{-1} x); S from
 * mispec_symshift_create_general.  The solver leaves S at a real probe shift afterwards, like the reference. */
int mispec_geneigs_create_complex_shift(mispec_ctx* ctx, mispec_symshift* S, int64_t nev, int64_t ncv, double sigmar, double sigmai,
                                        mispec_geneigs** out);

int mispec_geneigs_create_dense(mispec_ctx* ctx, const mispec_dense* D, int64_t nev, int64_t ncv, mispec_geneigs** out);

/* ---------------------------------------------------------------------------
 * Complex scalars — the reference's factorisation templates instantiated with std::complex<double>
 * (LinAlg/Arnoldi.h:136-295, LinAlg/Lanczos.h:62-187 over MatOp/DenseGenMatProd.h / DenseHermMatProd.h; test/Arnoldi.cpp:122-158).
 * Complex numbers cross the boundary interleaved: a `double*` of 2 * count entries (re, im), the layout of std::complex<double>.
 * The basis, the residual and a dense operator live in HBM; steps are host-driven in the reference's order.
 * uplo = 'L' / 'U': Hermitian, only that triangle of the input is read (mirrored conjugated, diagonal taken real); 0: general.
 * ------------------------------------------------------------------------- */
typedef struct mispec_zdense mispec_zdense;
int mispec_zdense_upload(mispec_ctx* ctx, int64_t rows, int64_t cols, const double* data_host, int64_t ld_host, int row_major,
                         char uplo, mispec_zdense** out);
int mispec_zdense_destroy(mispec_zdense* D);
int64_t mispec_zdense_rows(const mispec_zdense* D);
int64_t mispec_zdense_cols(const mispec_zdense* D);
int mispec_zdense_gemv_host(const mispec_zdense* D, const double* x_host, double* y_host); /* literal perform_op */
int mispec_zdense_coeff(const mispec_zdense* D, int64_t i, int64_t j, double* out_re_im);
/* user operator on host pointers: y = Op(x), n complex entries each; non-zero return = error */
typedef int (*mispec_zop_fn)(void* user, const double* x_host, double* y_host);
typedef struct mispec_zfac mispec_zfac;
/* hermitian != 0: the three-term flow of Lanczos.h; 0: the full projection of Arnoldi.h */
int mispec_zfac_create_dense(mispec_ctx* ctx, const mispec_zdense* D, int ncv, int hermitian, mispec_zfac** out);
int mispec_zfac_create_op(mispec_ctx* ctx, mispec_zop_fn op, void* op_user, int64_t n, int ncv, int hermitian, mispec_zfac** out);
int mispec_zfac_destroy(mispec_zfac* F);
int mispec_zfac_init(mispec_zfac* F, const double* v0_host, int64_t* op_counter);
int mispec_zfac_factorize(mispec_zfac* F, int from_k, int to_m, int64_t* op_counter);
int mispec_zfac_subspace_dim(const mispec_zfac* F);
int mispec_zfac_f_norm(const mispec_zfac* F, double* out);
int mispec_zfac_get_H(const mispec_zfac* F, double* H_host);            /* ncv x ncv, column-major */
int mispec_zfac_get_V(const mispec_zfac* F, int ncols, double* V_host); /* n x ncols, column-major */
int mispec_zfac_get_f(const mispec_zfac* F, double* f_host);

#ifdef __cplusplus
}
#endif

#endif /* MISPEC_EXTRAS_H */
