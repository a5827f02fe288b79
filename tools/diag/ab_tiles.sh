# same-box A/B inside the bf16-mixed step: FS2_GEMM_EXCLUDE_TILES sets, alternating; usage: ab_tiles.sh "33,34" "34" "33" ""
B="python3 bench.py --precision bf16-mixed --batch 64 --no-cpu-baseline --no-extra-legs --no-roofline --steps 300"
for r in 1 2 3; do
  for ex in "$@"; do
    FS2_GEMM_EXCLUDE_TILES="$ex" $B 2>&1 | grep -E "timed region" | sed "s/^/excluded [$ex]: /"
  done
done
