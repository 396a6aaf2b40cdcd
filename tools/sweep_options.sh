#!/bin/bash
# Bench runs over a list of context options, one line per run; stops at the first failure.
#   tools/sweep_options.sh "extend.leaf_lanes=8" "extend.leaf_lanes=12 extend.refill=32" ...
for opts in "$@"; do
  args=""
  for o in $opts; do args="$args --option $o"; done
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 8 $args > /tmp/sweep.json 2>/tmp/sweep.err || { echo "failed: $opts"; tail -n 3 /tmp/sweep.err; exit 1; }
  python - "$opts" <<'PY'
import json, sys
for l in open("/tmp/sweep.json"):
    if l.startswith("{"):
        d = json.loads(l)
        print(sys.argv[1], round(d["value"]), round(d["ms_per_step"], 3), round(d["roofline"]["avg_launch_ms"], 4))
PY
done
