# same-box A/B of an environment switch inside the fp32 headline step; usage: ab_env32.sh VAR valueA valueB [...]
V=$1; shift
B="python3 bench.py --no-cpu-baseline --no-extra-legs --no-roofline --steps 200"
for r in 1 2 3; do
  for x in "$@"; do
    env $V=$x $B 2>&1 | grep -E "timed region" | sed "s/^/fp32 $V=$x: /"
  done
done
