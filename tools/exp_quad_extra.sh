# Sensitivity of the HBM-regime kernel to the vector instructions of its four-child step: the
# same launches with the in-tree library and with a variant that issues 33 more per step.
for lib in base qx; do
  if [ $lib = base ]; then unset MIPT_LIB; else export MIPT_LIB=$PWD/vlib/$lib.so; fi
  echo "== $lib" >> gpurun_out/r3_quad_extra.log
  SCENE=cube LGS="24" timeout -k 10 300 python -u tools/exp_launch_size.py 2>/dev/null >> gpurun_out/r3_quad_extra.log
done
