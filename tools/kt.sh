#!/bin/bash
# one-stream kernel trace of a bench configuration -> gpurun_out/TAG_one_stream_{kernel_stats.csv,step_breakdown.txt}
# usage (GPU box): tools/kt.sh TAG [bench.py flags]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/${TAG}_kt
FS2_SIDE_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra-legs --no-refine "$@" > $O/${TAG}_kt.log 2>&1 || { tail -5 $O/${TAG}_kt.log; exit 1; }
cp $O/${TAG}_kt/*/*_kernel_stats.csv $O/${TAG}_one_stream_kernel_stats.csv
python3 tools/step_breakdown.py $O/${TAG}_kt/*/*_kernel_trace.csv 5 > $O/${TAG}_one_stream_step_breakdown.txt
rm -rf $O/${TAG}_kt
head -40 $O/${TAG}_one_stream_step_breakdown.txt | cut -c1-140
