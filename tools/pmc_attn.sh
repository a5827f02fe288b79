#!/bin/bash
# usage: tools/pmc_attn.sh COUNTER [COUNTER...]   (one rocprofv3 pass per counter over tools/bench_attn.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in "$@"; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pattn_$c -- python3 tools/bench_attn.py > gpurun_out/pattn_$c.log 2>&1 || exit 1
  python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/pattn_$c/*/*_counter_collection.csv")[0]
acc = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1]
    if "attn" in k:
        acc.setdefault(k, []).append(float(r["Counter_Value"]))
print("$c", {k: round(sum(v[-3:]) / len(v[-3:])) for k, v in acc.items()})
PY
done
