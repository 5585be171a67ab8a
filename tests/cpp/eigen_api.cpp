// Compile-and-link check of the Eigen-facing constructors of include/Spectra (the MISPEC_HAVE_EIGEN blocks), built against
// tests/cpp/eigen_lite — a stand-in for Eigen's API, NOT Eigen (this image has no Eigen).  The program
// is never run by the CPU tests: constructing the operators needs a GPU.  It is the source a Spectra user would write.
#include <Eigen/Core>
#include <Eigen/SparseCore>

#include <Spectra/DavidsonSymEigsSolver.h>
#include <Spectra/GenEigsComplexShiftSolver.h>
#include <Spectra/GenEigsRealShiftSolver.h>
#include <Spectra/GenEigsSolver.h>
#include <Spectra/MatOp/DenseCholesky.h>
#include <Spectra/MatOp/DenseGenComplexShiftSolve.h>
#include <Spectra/MatOp/DenseGenMatProd.h>
#include <Spectra/MatOp/DenseGenRealShiftSolve.h>
#include <Spectra/MatOp/DenseSymMatProd.h>
#include <Spectra/MatOp/DenseSymShiftSolve.h>
#include <Spectra/MatOp/SparseCholesky.h>
#include <Spectra/MatOp/SparseGenComplexShiftSolve.h>
#include <Spectra/MatOp/SparseGenMatProd.h>
#include <Spectra/MatOp/SparseGenRealShiftSolve.h>
#include <Spectra/MatOp/SparseRegularInverse.h>
#include <Spectra/MatOp/SparseSymMatProd.h>
#include <Spectra/MatOp/SparseSymShiftSolve.h>
#include <Spectra/MatOp/SymShiftInvert.h>
#include <Spectra/SymEigsShiftSolver.h>
#include <Spectra/SymEigsSolver.h>
#include <Spectra/SymGEigsShiftSolver.h>
#include <Spectra/SymGEigsSolver.h>
#include <Spectra/contrib/PartialSVDSolver.h>

#include <cstdio>

#ifndef MISPEC_HAVE_EIGEN
#error "this file must be compiled with an <Eigen/Core> on the include path"
#endif

using namespace Spectra;

static_assert(std::is_same<DenseMatrix<double>, Eigen::MatrixXd>::value, "results are Eigen matrices when Eigen is present");
static_assert(std::is_same<DenseVector<double>, Eigen::VectorXd>::value, "results are Eigen vectors when Eigen is present");

int main(int argc, char**)
{
    if (argc < 100)  // never true at run time in the tests: compile-and-link check only
    {
        std::printf("eigen api check: compiled\n");
        return 0;
    }
    // README.md:150-180 of the reference, verbatim apart from the matrix contents
    Eigen::MatrixXd M(10, 10);
    for (int j = 0; j < 10; j++)
        for (int i = 0; i < 10; i++)
            M(i, j) = (i == j) ? i + 1.0 : 0.0;
    DenseSymMatProd<double> dop(M);
    SymEigsSolver<DenseSymMatProd<double>> deigs(dop, 3, 6);
    deigs.init();
    deigs.compute(SortRule::LargestAlge);
    Eigen::VectorXd evalues = deigs.eigenvalues();
    Eigen::MatrixXd evecs = deigs.eigenvectors();
    std::printf("%g %g\n", evalues[0], evecs(0, 0));
    DenseGenMatProd<double> gop(M);
    GenEigsSolver<DenseGenMatProd<double>> geigs(gop, 3, 6);
    DenseSymShiftSolve<double> dsop(M);
    SymEigsShiftSolver<DenseSymShiftSolve<double>> dshift(dsop, 3, 6, 0.0);
    DenseGenRealShiftSolve<double> drs(M);
    GenEigsRealShiftSolver<DenseGenRealShiftSolve<double>> drshift(drs, 3, 6, 0.5);
    DenseGenComplexShiftSolve<double> dcs(M);
    GenEigsComplexShiftSolver<DenseGenComplexShiftSolve<double>> dcshift(dcs, 3, 6, 0.5, 0.5);
    DenseCholesky<double> dchol(M);
    SymGEigsSolver<DenseSymMatProd<double>, DenseCholesky<double>, GEigsMode::Cholesky> dg(dop, dchol, 3, 6);

    // sparse operators from Eigen::SparseMatrix
    std::vector<int> outer(11), inner(10);
    std::vector<double> vals(10);
    for (int j = 0; j < 10; j++)
    {
        outer[j] = j;
        inner[j] = j;
        vals[j] = j + 1.0;
    }
    outer[10] = 10;
    Eigen::SparseMatrix<double> S(10, 10, outer, inner, vals);
    SparseSymMatProd<double> sop(S);
    SymEigsSolver<SparseSymMatProd<double>> seigs(sop, 3, 6);
    SparseGenMatProd<double> sgop(S);
    GenEigsSolver<SparseGenMatProd<double>> sgeigs(sgop, 3, 6);
    SparseSymShiftSolve<double> ssop(S);
    SymEigsShiftSolver<SparseSymShiftSolve<double>> sshift(ssop, 3, 6, 0.0);
    SparseGenRealShiftSolve<double> srs(S);
    GenEigsRealShiftSolver<SparseGenRealShiftSolve<double>> srshift(srs, 3, 6, 0.5);
    SparseGenComplexShiftSolve<double> scs(S);
    GenEigsComplexShiftSolver<SparseGenComplexShiftSolve<double>> scshift(scs, 3, 6, 0.5, 0.5);
    SparseCholesky<double> schol(S);
    SparseRegularInverse<double> sreg(S);
    SymGEigsSolver<SparseSymMatProd<double>, SparseCholesky<double>, GEigsMode::Cholesky> g1(sop, schol, 3, 6);
    SymGEigsSolver<SparseSymMatProd<double>, SparseRegularInverse<double>, GEigsMode::RegularInverse> g2(sop, sreg, 3, 6);
    SymShiftInvert<double> pencil(S, S);
    SymGEigsShiftSolver<SymShiftInvert<double>, SparseSymMatProd<double>, GEigsMode::ShiftInvert> g3(pencil, sop, 3, 6, 0.5);
    DavidsonSymEigsSolver<SparseSymMatProd<double>> dav(sop, 2);
    PartialSVDSolver<Eigen::SparseMatrix<double>> svds(S, 2, 5);
    svds.compute();
    Eigen::VectorXd sv = svds.singular_values();
    Eigen::MatrixXd U = svds.matrix_U(2), V = svds.matrix_V(2);
    std::printf("%g %g %g\n", sv[0], U(0, 0), V(0, 0));
    Eigen::MatrixXd prod = sop * evecs;
    std::printf("%g %g\n", prod(0, 0), sop(1, 1));
    return 0;
}
