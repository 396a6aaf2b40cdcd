#!/bin/bash
# The bench under different vote thresholds of the persistent loop.  bash tools/sweep_vote.sh <outdir> leaf:refill ...
set -u
out=$1; shift
mkdir -p $out
for lr in "$@"; do
  L=${lr%%:*}; R=${lr#*:}
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-hbm-record --option extend.leaf_lanes=$L --option extend.refill=$R > $out/l${L}r${R}.json 2> $out/l${L}r${R}.err || { echo "L=$L R=$R failed"; tail -3 $out/l${L}r${R}.err; exit 1; }
  echo -n "leaf_lanes=$L refill=$R: "; python tools/show_bench.py $out/l${L}r${R}.json | head -1 | cut -d: -f2- | cut -c1-90
done
