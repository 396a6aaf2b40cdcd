#!/bin/bash
# The bench (with its HBM-regime record) under different splits of the persistent workgroups' LDS
# between stack entries and cached nodes.   bash tools/sweep_lds_split.sh <outdir> S:C ...
set -u
out=$1; shift
mkdir -p $out
for sc in "$@"; do
  S=${sc%%:*}; C=${sc#*:}
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --option extend.stack_entries=$S --option extend.cache_nodes=$C ${BENCH_ARGS:-} > $out/s${S}c${C}.json 2> $out/s${S}c${C}.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "S=$S C=$C rc=$rc"; tail -5 $out/s${S}c${C}.err; exit 1; fi
  python - "$out/s${S}c${C}.json" "S=$S C=$C" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
line = "%-12s %7.1f Mrays/s  extend %.4f ms/launch" % (sys.argv[2], d["value"], d["roofline"]["avg_launch_ms"])
h = d.get("roofline_hbm")
if h:
    line += " | hbm scene: sorted %.2f ms (%.0f Mrays/s)  queue order %.2f ms (%.0f Mrays/s)" % (
        h["avg_launch_ms"], h["mrays_per_s_kernel"], h["queue_order"]["avg_launch_ms"], h["queue_order"]["mrays_per_s"])
print(line)
PY
done
