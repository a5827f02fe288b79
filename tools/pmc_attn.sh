#!/bin/bash
# usage: tools/pmc_attn.sh "<bench_attn args>" "COUNTER COUNTER ..." ["COUNTER ..." ...]
# one rocprofv3 pass per quoted counter group (<= 8 SQ counters per pass) over tools/bench_attn.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="$1"; shift
i=0
for grp in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/pattn_$i
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pattn_$i -- python3 tools/bench_attn.py $ARGS > gpurun_out/pattn_$i.log 2>&1 || exit 1
  python3 - <<PY
import csv, glob, re
f = glob.glob("gpurun_out/pattn_$i/*/*_counter_collection.csv")[0]
acc = {}
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "").split("(")[0].split("<")[0]
    if "attn" in k:
        acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:24s} {c:28s} {sum(v[-3:]) / len(v[-3:]):16.0f}")
PY
done
