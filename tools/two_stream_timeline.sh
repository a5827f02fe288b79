cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
C="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra-legs"
rm -rf $O/tl_b $O/tl_f
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl_b -- python3 bench.py $C --precision bf16-mixed --batch 64 > $O/tl_b.log 2>&1 || exit 1
python3 tools/timeline.py $O/tl_b/*/*_kernel_trace.csv 250 > $O/r4_tl_bf16.txt
python3 tools/step_breakdown.py $O/tl_b/*/*_kernel_trace.csv 5 > $O/r4_tl_bf16_breakdown.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl_f -- python3 bench.py $C > $O/tl_f.log 2>&1 || exit 1
python3 tools/timeline.py $O/tl_f/*/*_kernel_trace.csv 250 > $O/r4_tl_fp32.txt
python3 tools/step_breakdown.py $O/tl_f/*/*_kernel_trace.csv 5 > $O/r4_tl_fp32_breakdown.txt
rm -rf $O/tl_b $O/tl_f
head -3 $O/r4_tl_bf16.txt $O/r4_tl_fp32.txt
