#!/bin/bash
# Per-kernel durations of the default bench run (rocprofv3 --kernel-trace --stats) and the
# bench's own JSON line from the same run.  Run on the GPU box from the repository root:
#   bash profiles/collect_kernel_trace.sh <tag>
set -u
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bench -- python3 bench.py > $out/bench.json 2> $out/bench.err
rc=$?
echo "kernel trace rc=$rc"
find $out -name "*kernel_stats.csv" | head -n 1 | xargs -r head -n 12
exit $rc
