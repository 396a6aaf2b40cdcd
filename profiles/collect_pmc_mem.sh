#!/bin/bash
# Vector-memory pipeline counters (address unit, data unit, L1) for the bench's kernels, one
# rocprofv3 --pmc pass per run, each group within the per-block limits of gfx950.  Run on the
# GPU box from the repository root:  bash profiles/collect_pmc_mem.sh <tag> [bench args...]
set -u
tag=${1:-r01}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcmem_$tag
mkdir -p $out
i=0
for grp in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE" \
           "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum" \
           "TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_RFIFO_STALL_CYCLES_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TD_LOAD_WAVEFRONT_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $out/g$i -o g$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/g$i.log 2>&1
  rc=$?
  echo "group $i ($grp): rc=$rc"
  if [ $rc -ne 0 ]; then tail -3 $out/g$i.log; exit 1; fi
done
python3 profiles/summarize_pmc.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
