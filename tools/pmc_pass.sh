#!/bin/bash
# one rocprofv3 counter pass: tools/pmc_pass.sh OUTDIR NAME "COUNTER [COUNTER..]" script.py   -> OUTDIR/NAME_counters.csv
OUT=$1; NAME=$2; CNT=$3; SCRIPT=$4
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$NAME -o $NAME -- python $GRAFT_REPO_ROOT/$SCRIPT > $GRAFT_REPO_ROOT/$OUT/pmc_$NAME.log 2>&1)
F=$(find $OUT/pmc_$NAME -name "*counter_collection.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in per.items():
    if "spmv" in k or "k_scale" in k or "vq" in k or "orth" in k:
        print(k, {c: (len(v), sum(v) / len(v)) for c, v in cs.items()})
PY
cp $F $OUT/${NAME}_counter_collection.csv; rm -rf $OUT/pmc_$NAME
