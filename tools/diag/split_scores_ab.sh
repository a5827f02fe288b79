timeout -k 10 600 python -m pytest tests/test_attention_gpu.py tests/test_gemm_split_gpu.py -x -q > gpurun_out/r4_split_tests.log 2>&1; tail -4 gpurun_out/r4_split_tests.log
S="python3 bench.py --precision 32-split --no-cpu-baseline --no-extra-legs --no-roofline --steps 200"
for r in 1 2 3; do
  FS2_ATTN_SCORES=0 $S 2>&1 | grep -E "timed region" | sed 's/^/32-split, scores recomputed: /'
  $S 2>&1 | grep -E "timed region" | sed 's/^/32-split, scores kept:       /'
done
