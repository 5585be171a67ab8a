#!/bin/bash
# round 4, GPU call 2: the whole GPU suite with one-sweep as the library default (tests parametrised over both modes where the
# verdict asked), then the int32 CSR kernel in the loop with the non-temporal hint of the basis loads compiled out
OUT=gpurun_out/r07b; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > $OUT/pytest_gpu.log 2>&1
tail -60 $OUT/pytest_gpu.log
timeout 200 python tools/probe_csr_variants.py 0 > $OUT/csr_nt_on.jsonl 2> $OUT/csr_nt_on.err
( cd spectra_amd/csrc && touch krylov.hip && make -s CXXFLAGS='--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wall -Wno-unused-function -DMISPEC_NO_NT_BASIS' ) > $OUT/rebuild.log 2>&1
timeout 200 python tools/probe_csr_variants.py 0 > $OUT/csr_nt_off.jsonl 2> $OUT/csr_nt_off.err
echo "--- NT on"; cat $OUT/csr_nt_on.jsonl; echo "--- NT off"; cat $OUT/csr_nt_off.jsonl; tail -3 $OUT/rebuild.log
