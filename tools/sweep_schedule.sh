#!/bin/bash
# GPU builder: insertion passes / treelet pass after every so many of them, against build time,
# slab tests per ray and the 2^24-ray launch on a 28 M-triangle surface (SCENE=cube|uv).
#   bash tools/sweep_schedule.sh <out.log> <passes>/<every>...
set -u
out=$1; shift
: > $out
for pe in "$@"; do
  echo "== passes/every $pe" >> $out
  PASSES=${pe%/*} TREELET=${pe#*/} LGS="24" timeout -k 10 200 python -u tools/exp_launch_size.py >> $out 2>&1 || exit 1
done
