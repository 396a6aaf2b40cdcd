#!/bin/bash
# A/B runs of variant builds of the library (vlib/<name>.so, made with
#   make -C gpu-pathtracer_amd EXTRA=-D... BUILD=build_<name> LIB=../vlib/<name>.so):
# a parity subset of the GPU tests, then the bench without its CPU and HBM legs.
#   bash tools/run_variants.sh <outdir> <name>...      ("base" = the in-tree libmipt.so)
set -u
out=$1; shift
mkdir -p $out
for name in "$@"; do
  if [ "$name" = base ]; then unset MIPT_LIB; else export MIPT_LIB=$PWD/vlib/$name.so; fi
  timeout -k 10 300 python -u -m pytest tests/test_gpu_trace.py tests/test_gpu_render.py -q --timeout 120 --timeout-method thread > $out/$name.tests.log 2>&1
  rc=$?
  echo "$name tests rc=$rc: $(tail -1 $out/$name.tests.log)"
  # tests that assert how many rays leave the voted loop depend on the stack size: listed, not fatal
  if [ $rc -ne 0 ]; then grep -E "^FAILED|differs" $out/$name.tests.log | head -8; fi
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit 1; fi
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-hbm-record ${BENCH_ARGS:-} > $out/$name.bench.json 2> $out/$name.bench.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "$name bench rc=$rc"; tail -5 $out/$name.bench.err; exit 1; fi
  python - "$out/$name.bench.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-10s %7.1f Mrays/s  step %.2f ms  extend %.4f ms/launch  shade %.2f ms/step  ties %d" % (
    sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["kernel_ms_per_step"]["shade"], d["tie_rays"]))
PY
done
