#!/bin/bash
# Collects hardware counters for the bench's kernels, one rocprofv3 --pmc pass per run
# (counter groups sized to the per-block limits of gfx950).  Run on the GPU box from the
# repository root:  bash profiles/collect_pmc.sh <tag>
set -u
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
mkdir -p $out
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $out/g$i -o g$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/g$i.log 2>&1
  rc=$?
  echo "group $i ($grp): rc=$rc"
  if [ $rc -ne 0 ]; then tail -3 $out/g$i.log; echo "stopping: nothing more is started on the GPU after a failed pass"; exit $rc; fi
done
python3 profiles/summarize_pmc.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
