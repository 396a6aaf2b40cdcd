#!/bin/bash
# Per-kernel durations (rocprofv3 --kernel-trace --stats) of the rows beyond the headline path:
# the bench with next-event estimation (shadow rays through the any-hit kernel) and the
# two-level (instanced) closest-hit benchmark.  Run on the GPU box from the repository root:
#   bash profiles/collect_next_rows.sh <tag>
set -u
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/next_$tag
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/nee -o nee -- python3 bench.py --nee --no-cpu-baseline --no-hbm-record --no-live-counters > $out/nee.json 2> $out/nee.err
rc=$?; echo "nee rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/inst -o inst -- python3 tools/bench_instanced.py 64 65536 > $out/inst.json 2> $out/inst.err
rc=$?; echo "instanced rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/bench_instanced.py 16 65536 > $out/inst16.json 2> $out/inst16.err
rc=$?; echo "instanced16 rc=$rc"; [ $rc -ne 0 ] && exit $rc
MESH=cube timeout -k 10 300 python3 tools/bench_instanced.py 64 65536 > $out/inst_cube.json 2> $out/inst_cube.err
rc=$?; echo "instanced, uniformly tessellated mesh rc=$rc"; exit $rc
