B="python3 bench.py --precision bf16-mixed --batch 64 --no-cpu-baseline --no-extra-legs --no-roofline --steps 150"
for r in 1 2; do
  FS2_GEMM_EXCLUDE_TILES=33 $B 2>&1 | grep -E "timed region" | sed 's/^/without 33: /'
  $B 2>&1 | grep -E "timed region" | sed 's/^/with 33:    /'
done
