#!/bin/bash
# round 6, GPU call 13: per-kernel times of the wide-band shift solve (wave-per-chunk), b = 32 and b = 64
OUT=gpurun_out/r11m; mkdir -p $OUT
export TMPDIR=/tmp
for B in 32 64; do
(cd /tmp && W5_B=$B timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$B -o t -- python $GRAFT_REPO_ROOT/tools/bench_configs.py w5 > $GRAFT_REPO_ROOT/$OUT/w5_b$B.json 2> $GRAFT_REPO_ROOT/$OUT/w5_b$B.err)
find $OUT/prof_$B -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/w5_b${B}_kernel_stats.csv
rm -rf $OUT/prof_$B; head -12 $OUT/w5_b${B}_kernel_stats.csv | cut -c1-230
done
