#!/bin/bash
# round 4, GPU call: phase 2 of the staged format — chunks per wavefront per batch (fewer rank rounds per entry) (library rebuilt on the box per variant)
OUT=gpurun_out/r07r; mkdir -p $OUT
BASE='--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wall -Wno-unused-function'
run() {  # run <tag> <defs>
  ( cd spectra_amd/csrc && touch staged.hip && make -s CXXFLAGS="$BASE $2" ) > $OUT/build_$1.log 2>&1
  (cd /tmp && export TMPDIR=/tmp && BENCH_FORMATS=4 MISPEC_SPMV_STAGED=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$1 -o t -- python $GRAFT_REPO_ROOT/tools/bench_staged.py > $GRAFT_REPO_ROOT/$OUT/stdout_$1.txt 2> $GRAFT_REPO_ROOT/$OUT/err_$1.txt)
  find $OUT/prof_$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$1.csv
  rm -rf $OUT/prof_$1
  echo "== $1 ($2)"; grep -E "staged" $OUT/kernel_stats_$1.csv | awk -F'",' '{print substr($1,1,60), $2}' | cut -c1-130; tail -1 $OUT/stdout_$1.txt | cut -c1-220
}
run base ""
run pw2 "-DMISPEC_ST_PERWAVE=2"
run pw2a8 "-DMISPEC_ST_PERWAVE=2 -DMISPEC_ST_AHEAD=8"
run pw4 "-DMISPEC_ST_PERWAVE=4"
run pw4a8 "-DMISPEC_ST_PERWAVE=4 -DMISPEC_ST_AHEAD=8"
run pw2r12t512 "-DMISPEC_ST_PERWAVE=2 -DMISPEC_ST_ROWBITS=12 -DMISPEC_ST_ROWTHREADS=512"
