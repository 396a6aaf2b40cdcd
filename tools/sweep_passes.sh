# Optimisation passes of the GPU builder against tree depth, tests per ray, build and trace time.
for scene in ${SCENES:-uv cube}; do
  for p in ${PASSES_LIST:-0 4 8 16 24}; do
    echo "== $scene passes $p ${OPTS:-}" >> gpurun_out/r3_passes.log
    SCENE=$scene PASSES=$p LGS="24" timeout -k 10 300 python -u tools/exp_launch_size.py ${OPTS:-} 2>/dev/null >> gpurun_out/r3_passes.log
  done
done
