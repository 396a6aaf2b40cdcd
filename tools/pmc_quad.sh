#!/bin/bash
# Counters of the closest-hit trace kernel over two-child and four-child nodes on the 28 M-triangle
# scene (tools/exp_quad.py sphere28m), one counter group per run.  On the GPU box from the
# repository root:   bash tools/pmc_quad.sh
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_quad
mkdir -p $out
export SPLITS="-1:-1" HOST=${HOST:-1}
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "TA_TA_BUSY_sum TD_TD_BUSY_sum"; do
  i=$((i+1))
  timeout -s KILL 400 rocprofv3 --pmc $grp --output-format csv -d $out/g$i -o g$i -- python3 tools/exp_quad.py sphere28m > $out/g$i.log 2>&1
  rc=$?
  echo "group $i ($grp): rc=$rc"
  if [ $rc -ne 0 ]; then tail -3 $out/g$i.log; exit 1; fi
done
python3 - $out <<'PY'
import collections, csv, glob, re, sys
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(root + "/g*/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    for r in rows:
        m = re.search(r"k_trace_ordered_persistent<[^>]*>", r["Kernel_Name"])
        if m:
            acc[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        # per kernel variant: launches in order (counted variant: 1; plain: warm-up + 3 sorted, warm-up + 3 queue order)
        print("   %-40s n=%d  sorted-order mean %.4g   queue-order mean %.4g" % (
            c, len(v), sum(v[:len(v)//2]) / max(1, len(v)//2), sum(v[len(v)//2:]) / max(1, len(v) - len(v)//2)))
PY
