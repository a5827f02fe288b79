#!/bin/bash
# usage: tools/pmc_probe.sh TAG "KIND M N K TILE" COUNTER [COUNTER...]   (one rocprofv3 pass per counter)
TAG=$1; ARGS=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in "$@"; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/probe_${TAG}_$c -- python3 tools/probe_gemm.py $ARGS > gpurun_out/probe_${TAG}_$c.log 2>&1 || exit 1
  python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/probe_${TAG}_$c/*/*_counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"]]
rows = rows[-3:]
vals = [float(r["Counter_Value"]) for r in rows]
print("${TAG} $c", sum(vals) / len(vals))
PY
done
