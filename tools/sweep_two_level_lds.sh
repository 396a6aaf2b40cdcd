# Two-level benchmark (4096 placements, uniformly tessellated 65 536-triangle mesh) against the
# split of a workgroup's LDS between the lanes' stacks and the cached top of the top-level tree.
set -e
for cfg in ${CFGS:-"30 0" "20 160" "16 224" "12 288" "24 96"}; do
  set -- $cfg
  echo "== stack $1 cache $2" >> gpurun_out/r3_two_level_lds.log
  MESH=${MESH:-cube} BUILD=gpu timeout -k 10 300 python -u tools/bench_instanced.py 64 65536 build.optimize_passes=8 extend.stack_entries=$1 extend.cache_nodes=$2 2>/dev/null >> gpurun_out/r3_two_level_lds.log
done
