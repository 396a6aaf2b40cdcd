#!/bin/bash
# Profiles of the default bench run (headline workload + its HBM-regime record), on the GPU
# box from the repository root:   bash profiles/collect_r03.sh
#  1. rocprofv3 --kernel-trace --stats over `python3 bench.py` (per-kernel durations, and the
#     bench's own JSON line from the same run);
#  2. rocprofv3 --pmc, one counter group per run (group sizes within gfx950's per-block limits),
#     over `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-counters --hbm-launches 3`.
# Nothing more is started on the GPU after a pass that fails.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_r03
mkdir -p $out
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 bench.py --no-live-counters > $out/bench.json 2> $out/bench.err
rc=$?
echo "kernel trace rc=$rc"
if [ $rc -ne 0 ]; then tail -5 $out/bench.err; exit $rc; fi
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_GATE_EN1_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TD_TD_BUSY_sum"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $grp --output-format csv -d $out/g$i -o g$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-counters --hbm-launches 3 > $out/g$i.log 2>&1
  rc=$?
  echo "group $i ($grp): rc=$rc"
  if [ $rc -ne 0 ]; then tail -3 $out/g$i.log; exit 1; fi
done
python3 profiles/summarize_r03.py $out
