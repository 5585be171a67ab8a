#!/bin/bash
# Compiles the reference's OWN test programs (/root/reference/test/*.cpp, Catch2 single header included), unmodified and where
# they lie, against this repository's headers (include/Spectra) — with tests/cpp/eigen_lite standing in for Eigen, which this
# image does not have — and links them with spectra_amd/libmispec.so.  Outputs: tests/cpp/_ref/<Name>.bin (git-ignored; they
# travel to the GPU box with the snapshot, the reference's sources do not and are not needed there).
#   usage: tests/cpp/build_reference_tests.sh [Name ...]      default: every program of the list below
# SURVEY.md 8f row 2.  Nothing from /root/reference is copied into the repository.
set -u
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
REF="${MISPEC_REFERENCE_DIR:-/root/reference}"
OUT="$ROOT/tests/cpp/_ref"
if [ ! -d "$REF/test" ]; then
    echo "build_reference_tests: $REF/test not found: nothing to do"
    exit 0
fi
mkdir -p "$OUT"
CXX="${CXX:-g++}"
FLAGS="-std=c++11 -O1 -w -I$ROOT/tests/cpp/eigen_lite -I$ROOT/include -I$REF/test"
# Programs that instantiate complex scalars and Eigen's dense decompositions next to the double cases (Givens, QR, Eigen, Arnoldi):
# the fuller stand-in oracle/eigen_shim (C++17) takes Eigen's place for these — test infrastructure on both sides of the
# comparison; what is under test is include/Spectra/LinAlg.  Givens, QR and Eigen test host-side classes only and run without a
# GPU; Arnoldi runs the device factorisations (real and complex, dense operators); the four *MatProd programs are
# TEMPLATE_TEST_CASEs over float and double (their Random / isApprox / sparse expressions are the stand-in's).
SHIM_PROGRAMS=" Givens QR Eigen Arnoldi SparseSymMatProd SparseGenMatProd DenseSymMatProd DenseGenMatProd "
SHIM_FLAGS="-std=c++17 -O2 -w -I$ROOT/oracle/eigen_shim -I$ROOT/include -I$REF/test"
LINK="-L$ROOT/spectra_amd -lmispec_extras -lmispec -Wl,-rpath,\$ORIGIN/../../../spectra_amd"
# Schur, Orthogonalization, Givens, QR and Eigen test host-side classes only and run without a GPU
LIST="${*:-SymEigs SymEigsShift GenEigs GenEigsRealShift GenEigsComplexShift SymGEigsCholesky SymGEigsRegInv SVD DavidsonSymEigs Example1 Example2 Example3 Example4 Schur Orthogonalization Givens QR Eigen Arnoldi SparseSymMatProd SparseGenMatProd DenseSymMatProd DenseGenMatProd}"
# Catch2's main(): compiled once
if [ ! -f "$OUT/tests-main.o" ] || [ "$REF/test/tests-main.cpp" -nt "$OUT/tests-main.o" ]; then
    $CXX $FLAGS -c "$REF/test/tests-main.cpp" -o "$OUT/tests-main.o" || exit 1
fi
status=0
for name in $LIST; do
    # up to date: nothing it is made from is newer than the binary
    F="$FLAGS"
    STANDIN="$ROOT/tests/cpp/eigen_lite"
    case "$SHIM_PROGRAMS" in *" $name "*) F="$SHIM_FLAGS"; STANDIN="$ROOT/oracle/eigen_shim" ;; esac
    if [ -f "$OUT/$name.bin" ] && [ -z "$(find "$ROOT/include" "$STANDIN" "$REF/test/$name.cpp" "$ROOT/spectra_amd/libmispec.so" "$ROOT/spectra_amd/libmispec_extras.so" -newer "$OUT/$name.bin" -print -quit)" ]; then
        echo "up to date $name.bin"
        continue
    fi
    if $CXX $F -c "$REF/test/$name.cpp" -o "$OUT/$name.o" 2> "$OUT/$name.log" &&
       $CXX "$OUT/$name.o" "$OUT/tests-main.o" $LINK -o "$OUT/$name.bin" 2>> "$OUT/$name.log"; then
        echo "built $name.bin"
        rm -f "$OUT/$name.log"
    else
        echo "FAILED $name (see tests/cpp/_ref/$name.log)"
        status=1
    fi
    rm -f "$OUT/$name.o"
done
exit $status
