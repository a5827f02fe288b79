#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh TAG [extra bench.py flags]
# Produces under gpurun_out/: TAG_bench.json (default line), TAG_one_stream_kernel_stats.csv + _step_breakdown.txt
# (per-kernel durations not inflated by the second stream), TAG_gemm_traffic.json (two --pmc passes, FETCH doubled),
# TAG_mfma_util.json (two --pmc passes).  Copy what should be judged into profiles/.
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
# the benchmarked step runs on the tile table the tuner's in-step stage leaves (bench.py refines by default): that table is
# saved here and REPLAYED by the traced / counter runs below (--no-refine + FS2_GEMM_TILE_CACHE), so the profiles are of
# the configuration the bench line measured -- and no refinement runs under the profiler
rm -f $O/${TAG}_tiles.json
FS2_BENCH_SAVE_TILES=$O/${TAG}_tiles.json FS2_BENCH_GEMM_BREAKDOWN=1 python3 bench.py "$@" > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || { tail -5 $O/${TAG}_bench.err; exit 1; }
[ -f $O/${TAG}_tiles.json ] && export FS2_GEMM_TILE_CACHE=$O/${TAG}_tiles.json
# per-shape GEMM table of the roofline pass (HIP-event intervals, one stream): the first table is the headline configuration's,
# a second one (default command only) the bf16_mixed_b64 leg's
grep -E "gemm Mc|roofline pass|ms/step" $O/${TAG}_bench.err | sed -E 's/^\[bench[^]]*\] //' > $O/${TAG}_gemm_shapes.txt
echo "bench done: $(python3 -c "import json;d=json.load(open('$O/${TAG}_bench.json'));print(d['ms_per_step'], d['value'], d['roofline']['frac'], (d.get('split_fp32') or {}).get('ms_per_step'))")"
COMMON="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra-legs --no-refine"
rm -rf $O/${TAG}_kt
FS2_SIDE_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -- python3 bench.py $COMMON "$@" > $O/${TAG}_kt.log 2>&1 || exit 1
cp $O/${TAG}_kt/*/*_kernel_stats.csv $O/${TAG}_one_stream_kernel_stats.csv
python3 tools/step_breakdown.py $O/${TAG}_kt/*/*_kernel_trace.csv 5 > $O/${TAG}_one_stream_step_breakdown.txt
python3 tools/gemm_trace_sum.py $O/${TAG}_kt/*/*_kernel_trace.csv 5 "$@" > $O/${TAG}_gemm_trace.json
head -12 $O/${TAG}_one_stream_step_breakdown.txt
NGEMM=$(python3 -c "import json;print(json.load(open('$O/${TAG}_bench.json'))['roofline']['launches_per_step'])")
PM="--steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-extra-legs --no-refine"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/${TAG}_pmc_$c
  FS2_SIDE_STREAM=0 timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${TAG}_pmc_$c -- python3 bench.py $PM "$@" > $O/${TAG}_pmc_$c.log 2>&1 || exit 1
done
python3 tools/pmc_traffic.py $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE $NGEMM "$@" > $O/${TAG}_gemm_traffic.json
cat $O/${TAG}_gemm_traffic.json
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rm -rf $O/${TAG}_pmc_$c
  FS2_SIDE_STREAM=0 timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${TAG}_pmc_$c -- python3 bench.py $PM "$@" > $O/${TAG}_pmc_$c.log 2>&1 || exit 1
done
python3 tools/pmc_mfma.py $O/${TAG}_pmc_SQ_VALU_MFMA_BUSY_CYCLES $O/${TAG}_pmc_GRBM_GUI_ACTIVE > $O/${TAG}_mfma_util.json
cat $O/${TAG}_mfma_util.json
rm -rf $O/${TAG}_pmc_* $O/${TAG}_kt
