#!/bin/bash
# As run_variants.sh, but the bench keeps its HBM-regime record (28 M-triangle scene) and runs no tests.
set -u
out=$1; shift
mkdir -p $out
for name in "$@"; do
  if [ "$name" = base ]; then unset MIPT_LIB; else export MIPT_LIB=$PWD/vlib/$name.so; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > $out/$name.bench.json 2> $out/$name.bench.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "$name bench rc=$rc"; tail -5 $out/$name.bench.err; exit 1; fi
  python - "$out/$name.bench.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h = d["roofline_hbm"]
print("%-10s %7.1f Mrays/s  extend %.4f ms/launch | hbm scene: sorted %.2f ms (%.0f Mrays/s, frac %.3f)  queue order %.2f ms (%.0f Mrays/s)  call incl sort %.0f Mrays/s" % (
    sys.argv[2], d["value"], d["roofline"]["avg_launch_ms"], h["avg_launch_ms"], h["mrays_per_s_kernel"], h["frac"],
    h["queue_order"]["avg_launch_ms"], h["queue_order"]["mrays_per_s"], h["mrays_per_s_call_incl_sort"]))
PY
done
