#!/bin/bash
# Workgroup size of the persistent traversal kernels against the share of LDS that goes to the
# copy of the top of the tree (one copy per workgroup: fewer, larger workgroups = more nodes).
# Variants: vlib/b1024.so (-DPT_TRACE_BLOCK=1024 -DPT_LDS_STACK_ENTRIES=15 -DPT_LDS_NODE_CACHE_MAX=1408),
#           vlib/b640.so  (-DPT_TRACE_BLOCK=640 -DPT_LDS_STACK_ENTRIES=24 -DPT_LDS_NODE_CACHE_MAX=1408 -DPT_EXTEND_MIN_WAVES=5)
set -u
out=$1
mkdir -p $out
run() {  # name lib options...
  name=$1; lib=$2; shift 2
  if [ "$lib" = base ]; then unset MIPT_LIB; else export MIPT_LIB=$PWD/vlib/$lib.so; fi
  opts=""; for o in "$@"; do opts="$opts --option $o"; done
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-hbm-record --no-live-counters $opts > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; return 1; }
  python tools/show_bench.py $out/$name.json | tail -1
}
run base base || exit 1
run b1024_1408 b1024 extend.stack_entries=15 extend.cache_nodes=1408 extend.blocks_per_cu=1 || exit 1
run b1024_1024 b1024 extend.stack_entries=15 extend.cache_nodes=1024 extend.blocks_per_cu=1 || exit 1
run b640_528 b640 extend.stack_entries=16 extend.cache_nodes=528 extend.blocks_per_cu=2 || exit 1
run b640_704 b640 extend.stack_entries=12 extend.cache_nodes=704 extend.blocks_per_cu=2 || exit 1
