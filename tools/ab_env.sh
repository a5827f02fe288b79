#!/bin/bash
# Same-box A/B of environment switches: alternates the given settings ROUNDS times, one short bench.py run each, and prints
# ms/step per run (DESIGN.md "How a change is judged": single runs on different boxes cannot resolve 0.2 ms).
# usage: tools/ab_env.sh "<bench.py flags>" ROUNDS "ENV1=.. ENV2=.." "ENVA=.." ...   ("-" = no extra environment)
flags="$1"; rounds="$2"; shift 2
for r in $(seq 1 "$rounds"); do
  for setting in "$@"; do
    if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
    out=$(env $envs python bench.py $flags --no-extra-legs --no-cpu-baseline --no-roofline 2>/tmp/ab_env.err | tail -1)
    echo "round $r [$setting] $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step  host", d.get("host_enqueue_ms_per_step"), " exchange_wait", d.get("exchange_wait_ms"))' 2>/dev/null || { echo FAILED; tail -3 /tmp/ab_env.err; })"
  done
done
