#!/bin/bash
# TEST INFRASTRUCTURE.  Builds oracle/_ref/libspectra_ref.so: the REFERENCE'S OWN solver code (yixuan/spectra headers, compiled
# from where they lie under /root/reference/include — nothing of them is copied into this repository) behind the C entry points
# of oracle/ref_driver.cpp, with oracle/eigen_shim standing in for Eigen (absent from this image; the reference's own build
# fetches it from the network, CMakeLists.txt:25-38).  Same compiler flags as the restatement it pins (oracle/Makefile:
# g++ -O2, no FMA contraction).  The output is git-ignored and travels to the GPU box with the snapshot like every built .so;
# /root/reference itself does not exist there and is never read at run time.
#   usage: oracle/build_ref.sh [--force]
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
if [ "${1:-}" != "--force" ] && [ -f "$LIB" ] &&
   [ -z "$(find "$HERE/ref_driver.cpp" "$HERE/eigen_shim" "$REF/include/Spectra" -newer "$LIB" -print -quit)" ]; then
    echo "up to date $LIB"
    exit 0
fi
${CXX:-g++} -std=c++17 -O2 -fPIC -ffp-contract=off -fvisibility=hidden -fvisibility-inlines-hidden -Wall -I"$HERE/eigen_shim" -I"$REF/include" -shared -Wl,-Bsymbolic "$HERE/ref_driver.cpp" -o "$LIB" || exit 1
echo "built $LIB"
