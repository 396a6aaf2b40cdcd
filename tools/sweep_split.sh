# Leaf-splitting parameters of the GPU builder against the closest-hit kernel time, 28 M-triangle
# latitude / longitude surface: "ratio max passes" per line of the list.
set -e
for cfg in ${CFGS:-"4 32 8" "4 128 8" "3 64 8" "6 64 8"}; do
  set -- $cfg
  echo "== ratio $1 max $2 passes $3" >> gpurun_out/r3_split_sweep.log
  PASSES=$3 LGS="${LGS:-14 24}" timeout -k 10 300 python -u tools/exp_launch_size.py build.split_ratio=$1 build.split_max=$2 2>/dev/null >> gpurun_out/r3_split_sweep.log
done
