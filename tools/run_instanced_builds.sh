# Two-level benchmark (4096 placements of a 65 536-triangle mesh) with the mesh's tree from the
# host builder, and built on the GPU under several leaf-splitting settings.
set -e
BUILD=host timeout -k 10 300 python -u tools/bench_instanced.py 64 65536 2>/dev/null >> gpurun_out/r3_instanced.log
for cfg in ${CFGS:-"0 64 50" "6 64 50" "4 64 50" "3 64 100" "2 64 100" "2 128 200"}; do
  set -- $cfg
  echo "== ratio $1 max $2 budget $3" >> gpurun_out/r3_instanced.log
  BUILD=gpu timeout -k 10 300 python -u tools/bench_instanced.py 64 65536 build.optimize_passes=8 build.split_ratio=$1 build.split_max=$2 build.split_budget=$3 2>/dev/null >> gpurun_out/r3_instanced.log
done
BUILD=gpu MESH=cube timeout -k 10 300 python -u tools/bench_instanced.py 64 65536 build.optimize_passes=8 2>/dev/null >> gpurun_out/r3_instanced.log
