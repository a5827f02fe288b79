# five alternations of an environment switch inside the bf16-mixed step; usage: ab_env5.sh VAR valueA valueB
V=$1; shift
B="python3 bench.py --precision bf16-mixed --batch 64 --no-cpu-baseline --no-extra-legs --no-roofline --steps 400"
for r in 1 2 3 4 5; do
  for x in "$@"; do
    env $V=$x $B 2>&1 | grep -E "timed region" | sed "s/^/$V=$x: /"
  done
done
