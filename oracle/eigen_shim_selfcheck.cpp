// TEST INFRASTRUCTURE.  Self-check of the dense decompositions oracle/eigen_shim adds for the reference's test/QR.cpp, Eigen.cpp
// (HouseholderQR, EigenSolver, ComplexEigenSolver / ComplexSchur): those programs only TIME Eigen's solvers or compare |Q|, so
// the stand-ins' own residuals are checked here (tests/test_ref_pin.py compiles and runs this; exit code 0 = all below 1e-12).
#include <Eigen/Core>
#include <Eigen/Eigenvalues>
#include <Eigen/QR>
#include <iostream>
int main(){
  std::srand(5);
  const int n=40;
  Eigen::MatrixXd A = Eigen::MatrixXd::Random(n,n);
  Eigen::EigenSolver<Eigen::MatrixXd> es(A);
  Eigen::MatrixXcd V = es.eigenvectors(); Eigen::VectorXcd d = es.eigenvalues();
  Eigen::MatrixXcd Ac = A.cast<std::complex<double>>();
  Eigen::MatrixXcd err = Ac*V - V*d.asDiagonal();
  std::cout << "real gen resid " << err.cwiseAbs().maxCoeff() << " info " << es.info() << "\n";
  Eigen::MatrixXcd B = Eigen::MatrixXcd::Random(n,n);
  Eigen::ComplexEigenSolver<Eigen::MatrixXcd> ces(B);
  Eigen::MatrixXcd err2 = B*ces.eigenvectors() - ces.eigenvectors()*ces.eigenvalues().asDiagonal();
  std::cout << "complex resid " << err2.cwiseAbs().maxCoeff() << "\n";
  Eigen::HouseholderQR<Eigen::MatrixXd> qr(A);
  Eigen::MatrixXd Q = qr.householderQ(); Eigen::MatrixXd R = qr.matrixQR();
  Eigen::MatrixXd e3 = Q*R - A; Eigen::MatrixXd e4 = Q.transpose()*Q - Eigen::MatrixXd::Identity(n,n);
  std::cout << "qr " << e3.cwiseAbs().maxCoeff() << " " << e4.cwiseAbs().maxCoeff() << "\n";
  const double worst = std::max(std::max(err.cwiseAbs().maxCoeff(), err2.cwiseAbs().maxCoeff()),
                                std::max(e3.cwiseAbs().maxCoeff(), e4.cwiseAbs().maxCoeff()));
  return (es.info() == Eigen::Success && ces.info() == Eigen::Success && worst < 1e-12) ? 0 : 1;
}
