#!/bin/bash
# HBM-side traffic of the bench's kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3
# --pmc passes (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2).  Run on the GPU box from
# the repository root:  bash profiles/collect_traffic.sh <tag>
set -u
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/traffic_$tag
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $out/g_$c -o g_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/$c.log 2>&1
  rc=$?
  echo "$c rc=$rc"
  if [ $rc -ne 0 ]; then echo "stopping: nothing more is started on the GPU after a failed pass"; exit $rc; fi
done
python3 profiles/summarize_pmc.py $out | tee $out/summary.txt
