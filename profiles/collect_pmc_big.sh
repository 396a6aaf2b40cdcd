#!/bin/bash
# Counters of the closest-hit kernel on the 28 M-triangle scene (beyond the caches), one
# rocprofv3 --pmc pass per run.  Run on the GPU box from the repository root:
#   bash profiles/collect_pmc_big.sh <tag> [tool options, e.g. trace.sort_rays=0]
set -u
tag=${1:-r01}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcbig_$tag
mkdir -p $out
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $grp --output-format csv -d $out/g$i -o g$i -- python3 tools/big_scene_trace.py "$@" > $out/g$i.log 2>&1
  rc=$?
  echo "group $i ($grp): rc=$rc"
  if [ $rc -ne 0 ]; then tail -3 $out/g$i.log; exit 1; fi
done
python3 profiles/summarize_pmc.py $out > $out/summary.txt 2>&1
grep -A30 "k_trace_ordered_persistent" $out/summary.txt
