#!/bin/bash
# usage (GPU box): tools/pmc_gemm_bf16.sh CASE TILE "COUNTER COUNTER ..." ["COUNTERS of a second pass" ...]
# one rocprofv3 --pmc pass per quoted group; prints the mean over the last three gemm dispatches per counter
CASE=$1; TILE=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "$@"; do
  i=$((i+1))
  D=gpurun_out/pmcb_${CASE}_${TILE}_$i
  rm -rf $D
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $D -- python3 tools/probe_gemm_bf16.py $CASE $TILE > $D.log 2>&1 || { echo "pass $i failed"; tail -3 $D.log; continue; }
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$D/*/*_counter_collection.csv")[0]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "gemmb" in r["Kernel_Name"]:
        by[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in by.items():
    v = v[-3:]
    print("$CASE tile $TILE", k, sum(v) / len(v))
PY
  rm -rf $D
done
