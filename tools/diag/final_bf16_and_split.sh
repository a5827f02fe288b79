bash tools/profile_round.sh r04_b_bf16_b64 --precision bf16-mixed --batch 64 || exit 1
S="python3 bench.py --precision 32-split --no-cpu-baseline --no-extra-legs --no-roofline --steps 200"
for r in 1 2; do
  FS2_ATTN_SPILL=0 $S 2>&1 | grep -E "timed region" | sed 's/^/32-split, recomputing attention backward: /'
  $S 2>&1 | grep -E "timed region" | sed 's/^/32-split, spilled dS:                    /'
done
