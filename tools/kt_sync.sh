cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; rm -rf $O/sync_kt
FS2_BENCH_FORCE_SYNC=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sync_kt -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra-legs > $O/sync_kt.log 2>&1 || { tail -5 $O/sync_kt.log; exit 1; }
python3 tools/step_breakdown.py $O/sync_kt/*/*_kernel_trace.csv 5 > $O/r5_sync_two_stream_step_breakdown.txt
grep -i "nccl\|rccl\|AllReduce\|queue\|wall" $O/r5_sync_two_stream_step_breakdown.txt | head -20
rm -rf $O/sync_kt
