#!/bin/bash
# Optimiser's search cap (nodes one search may look at) against build time and tree quality on the
# 28 M-triangle surfaces; variants vlib/cap<N>.so built with EXTRA=-DPT_U_SEARCH_CAP=<N>.
#   bash tools/exp_search_cap.sh <out.log> <name>...     ("base" = the in-tree library)
set -u
out=$1; shift
: > $out
for name in "$@"; do
  if [ "$name" = base ]; then unset MIPT_LIB; else export MIPT_LIB=$PWD/vlib/$name.so; fi
  for scene in cube uv; do
    echo "== $name $scene" >> $out
    SCENE=$scene LGS="24" timeout -k 10 200 python -u tools/exp_launch_size.py >> $out 2>&1 || exit 1
  done
done
