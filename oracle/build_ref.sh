#!/bin/bash
# TEST INFRASTRUCTURE.  Builds oracle/_ref/libspectra_ref.so: the REFERENCE'S OWN solver code (yixuan/spectra headers, compiled
# from where they lie under /root/reference/include — nothing of them is copied into this repository) behind the C entry points
# of oracle/ref_driver.cpp, with oracle/eigen_shim standing in for Eigen (absent from this image; the reference's own build
# fetches it from the network, CMakeLists.txt:25-38).  Same compiler flags as the restatement it pins (oracle/Makefile:
# g++ -O2, no FMA contraction).  The output is git-ignored and travels to the GPU box with the snapshot like every built .so;
# /root/reference itself does not exist there and is never read at run time.
#   usage: oracle/build_ref.sh [--force] [--tests]     --tests: also the reference's own test programs on its own headers (oracle/_ref/tests/*.bin)
set -u
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${MISPEC_REFERENCE_DIR:-/root/reference}"
OUT="$HERE/_ref"
if [ ! -d "$REF/include/Spectra" ]; then
    echo "build_ref: $REF/include/Spectra not found: keeping whatever is in $OUT"
    exit 0
fi
mkdir -p "$OUT"
LIB="$OUT/libspectra_ref.so"
if [ "${1:-}" = "" ] && [ -f "$LIB" ] &&
   [ -z "$(find "$HERE/ref_driver.cpp" "$HERE/eigen_shim" "$REF/include/Spectra" -newer "$LIB" -print -quit)" ]; then
    echo "up to date $LIB"
    exit 0
fi
build_tests() {
    # The reference's OWN Catch2 programs on the reference's OWN headers, with the stand-in algebra: a self-check of oracle/eigen_shim
    # (if the stand-in mis-evaluated an expression the reference writes, the reference's own acceptance tests would notice).
    # All 31 programs of the reference's test/CMakeLists.txt.  QR, Eigen, Arnoldi, HermEigs, ComplexEigs, BKLDLT and GenEigsComplexShift
    # instantiate complex scalars (the stand-in's ComplexSchur / EigenSolver / HouseholderQR / complex LU are textbook methods,
    # Eigen/Eigenvalues, Eigen/QR, Eigen/LU here); SymGEigsShift mixes dense and sparse operands of one pencil (dense (+|-)= triangular
    # views of dense and sparse matrices); the Davidson family uses Array expressions of the stand-in.
    local T="$OUT/tests"
    mkdir -p "$T"
    if [ ! -f "$T/tests-main.o" ]; then
        ${CXX:-g++} -std=c++17 -O1 -w -I"$REF/test" -c "$REF/test/tests-main.cpp" -o "$T/tests-main.o" || return 1
    fi
    local status=0
    # SymEigsShift, GenEigsRealShift, SymGEigsRegInv, SymGEigsCholesky, Example3: with the stand-in's plain SparseLU / ConjugateGradient /
    # SimplicialLLT behind the reference's operators (dense factors: fine at the n <= 1000 these programs use)
    for name in SymEigs GenEigs Schur Example1 Example2 Example4 SparseSymMatProd SparseGenMatProd DenseSymMatProd DenseGenMatProd \
                SymEigsShift GenEigsRealShift SymGEigsRegInv SymGEigsCholesky Example3 SVD Givens QR Eigen Arnoldi \
                HermEigs ComplexEigs BKLDLT Orthogonalization JDSymEigsBase JDSymEigsDPRConstructor RitzPairs SearchSpace DavidsonSymEigs \
                GenEigsComplexShift SymGEigsShift; do
        if [ -f "$T/$name.bin" ] && [ -z "$(find "$HERE/eigen_shim" "$REF/include/Spectra" "$REF/test/$name.cpp" -newer "$T/$name.bin" -print -quit)" ]; then
            continue
        fi
        if ${CXX:-g++} -std=c++17 -O2 -ffp-contract=off -w -I"$HERE/eigen_shim" -I"$REF/include" -I"$REF/test" -c "$REF/test/$name.cpp" -o "$T/$name.o" 2> "$T/$name.log" &&
           ${CXX:-g++} "$T/$name.o" "$T/tests-main.o" -o "$T/$name.bin" 2>> "$T/$name.log"; then
            echo "built $T/$name.bin"
            rm -f "$T/$name.log"
        else
            echo "FAILED $name (see $T/$name.log)"
            status=1
        fi
        rm -f "$T/$name.o"
    done
    return $status
}
if [ "${1:-}" = "--tests" ] || [ "${2:-}" = "--tests" ]; then
    build_tests || exit 1
fi
[ "${1:-}" = "--tests" ] && [ -f "$LIB" ] && [ -z "$(find "$HERE/ref_driver.cpp" "$HERE/eigen_shim" "$REF/include/Spectra" -newer "$LIB" -print -quit)" ] && exit 0
${CXX:-g++} -std=c++17 -O2 -fPIC -ffp-contract=off -fvisibility=hidden -fvisibility-inlines-hidden -Wall -I"$HERE/eigen_shim" -I"$REF/include" -shared -Wl,-Bsymbolic "$HERE/ref_driver.cpp" -o "$LIB" || exit 1
echo "built $LIB"
