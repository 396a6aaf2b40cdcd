# The four-child step's coarse slab test in the reference's form (variant library vlib/qr.so,
# built with -DPT_QUAD_REFERENCE_FORM) against the one-instruction-per-plane form of the tree.
for lib in qr base; do
  if [ $lib = base ]; then unset MIPT_LIB; else export MIPT_LIB=$PWD/vlib/$lib.so; fi
  for scene in cube uv; do
    echo "== $lib $scene" >> gpurun_out/r3_quad_form.log
    SCENE=$scene LGS="24" timeout -k 10 300 python -u tools/exp_launch_size.py 2>/dev/null | tail -1 >> gpurun_out/r3_quad_form.log
  done
done
