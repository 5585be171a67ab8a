#!/bin/bash
# Runs the reference's own test programs built by tests/cpp/build_reference_tests.sh (needs a GPU); one line per program.
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/refprogs}; mkdir -p $OUT
for b in ${PROGS:-SymEigs SymEigsShift GenEigs Example1 Example2 Example3 Example4 SVD SymGEigsCholesky SymGEigsRegInv GenEigsRealShift GenEigsComplexShift DavidsonSymEigs}; do
  t0=$(date +%s%N)
  timeout ${PER:-40} tests/cpp/_ref/$b.bin > $OUT/$b.log 2>&1; rc=$?
  t1=$(date +%s%N)
  echo "$b rc=$rc $(( (t1 - t0) / 1000000 ))ms $(grep -E 'All tests passed|test cases:' $OUT/$b.log | tail -1)"
done | tee $OUT/summary.txt
