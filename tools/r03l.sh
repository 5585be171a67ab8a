OUT=gpurun_out/r03l; mkdir -p $OUT
for sk in 0 1 2 4 6 8 15; do
(cd /tmp && export TMPDIR=/tmp && MISPEC_SHIFT_DEBUG_SKIP=$sk timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof$sk -o p -- python $GRAFT_REPO_ROOT/tools/shift_kernel_probe.py > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/rocprof$sk.err)
f=$(find $OUT/prof$sk -name '*kernel_stats.csv' | head -1); echo "skip=$sk $(grep chunk_solve_lds $f | sed 's/(long.*)",/",/; s/(anonymous namespace):://g' | cut -c1-120)" >> $OUT/skip.txt; rm -rf $OUT/prof$sk
done
cat $OUT/skip.txt
