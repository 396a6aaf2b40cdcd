"""Turns a profiles/collect_r03.sh collection (gpurun_out/prof_r03) into the committed summaries:

  profiles/r03_bench_kernel_stats.csv     rocprofv3 --stats per-kernel durations of `python3 bench.py`
  profiles/r03_bench_under_rocprof.json   the bench's own JSON line from that same run
  profiles/r03_trace_kernel_launches.txt  every launch of the closest-hit trace kernel of the HBM-regime
                                          record, in order (counted, 4 in origin-cell order, 4 in queue order)
  profiles/r03_pmc_summary.txt            counters per kernel and launch (the trace kernel split by ray order)
  profiles/r03_traffic.json               HBM-side bytes per launch of the extend kernel (+ derived ratios)
  profiles/r03_traffic_hbm.json           the same for the trace kernel on the 28 M-triangle scene

    python3 profiles/summarize_r03.py gpurun_out/prof_r03
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

root = sys.argv[1]
here = os.path.dirname(os.path.abspath(__file__))


def kernel_name(raw):
    m = re.search(r"\bk_\w+(<[^>]*>)?", raw)
    return m.group(0) if m else raw[:48]


# ---- kernel trace --------------------------------------------------------------------------------
stats = glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(here, "r03_bench_kernel_stats.csv"))
with open(os.path.join(here, "r03_bench_under_rocprof.json"), "w") as f:
    f.writelines(l for l in open(root + "/bench.json") if l.startswith("{"))
trace = glob.glob(root + "/trace/**/*kernel_trace.csv", recursive=True)
if trace:
    rows = [r for r in csv.DictReader(open(trace[0])) if "k_trace_ordered_persistent" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    with open(os.path.join(here, "r03_trace_kernel_launches.txt"), "w") as f:
        f.write("launches of the closest-hit trace kernel in the bench's HBM-regime record, in order\n"
                "(2 with test counters compiled in: two-child and four-child walk; then warm-up + 7 timed in origin-cell order,\n"
                "then warm-up + 7 timed in queue order, on the uniformly tessellated surface; the same eighteen again on the\n"
                "latitude / longitude surface; a last small one belongs to the CPU-baseline section)\n")
        for r in rows:
            f.write("%-60s %8.3f ms\n" % (kernel_name(r["Kernel_Name"]),
                                          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))

# ---- counters --------------------------------------------------------------------------------------
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(root + "/g*/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    for r in rows:
        acc[kernel_name(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))


def split_trace_kernel(name, values):
    """Launches of the uncounted trace kernel in a bench run with --hbm-launches 3, in order: on the
    uniformly tessellated surface 4 in origin-cell order, then 4 in queue order, then the same on
    the latitude / longitude one.  With four-child nodes the two orders are two kernels (without /
    with the LDS copy of the top), each with 4 launches per surface."""
    if name.startswith("k_trace_ordered_persistent<false") and len(values) == 8:
        return {"uniform tessellation": values[:4], "pole fans": values[4:]}
    return {"": values}


lines = []
per_launch = collections.defaultdict(dict)
for name in sorted(acc):
    if not name.startswith("k_"):
        continue
    for counter in sorted(acc[name]):
        for label, vals in split_trace_kernel(name, acc[name][counter]).items():
            key = name + (" [" + label + "]" if label else "")
            per_launch[key][counter] = sum(vals) / len(vals)
for key in sorted(per_launch):
    lines.append(key)
    for counter, v in sorted(per_launch[key].items()):
        lines.append("    %-40s per launch %.6g" % (counter, v))
with open(os.path.join(here, "r03_pmc_summary.txt"), "w") as f:
    f.write("rocprofv3 --pmc, one counter group per run of `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline`;\n"
            "values are means per launch of the kernel (summed over XCDs / channels as rocprofv3 reports them).\n"
            "k_trace_ordered_persistent<counting, LDS copy of the top, 64-bit addresses, four-child nodes>: on the\n"
            "28 M-triangle scenes <false, false, false, true> traces the rays in origin-cell order and\n"
            "<false, true, false, true> the same rays in queue order (four launches each per surface: the\n"
            "uniformly tessellated one, which the bench's second roofline record is quoted on, and the\n"
            "latitude / longitude one with its pole fans cut into several leaves per sliver).\n")
    f.write("\n".join(lines) + "\n")


def derived(c):
    cyc = c["GRBM_GUI_ACTIVE"] / 8                     # the counter sums the 8 XCDs
    out = {"launch_ms_by_counters": cyc / 2.4e6,
           "l2_hit_rate": c["TCC_HIT_sum"] / c["TCC_REQ_sum"],
           "valu_busy": c["SQ_INSTS_VALU"] * 4 / (1024 * cyc),
           "wave_lifetime_waiting_on_memory": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
           "l1_accesses_per_cu_clock": c["TCP_TOTAL_CACHE_ACCESSES_sum"] / (256 * cyc),
           "l1_miss_requests_per_launch": c["TCP_TCC_READ_REQ_sum"],
           "mean_l1_miss_latency_cycles": c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"],
           "address_unit_busy": c["TA_TA_BUSY_sum"] / (256 * cyc),
           "data_return_unit_busy": c["TD_TD_BUSY_sum"] / (256 * cyc),
           "utcl1_miss_fraction": c["TCP_UTCL1_TRANSLATION_MISS_sum"] / c["TCP_UTCL1_REQUEST_sum"],
           "lds_instructions_per_launch": c["SQ_INSTS_LDS"],
           "lds_bank_conflict_cycles_per_launch": c["SQ_LDS_BANK_CONFLICT"]}
    return {k: round(v, 4) if v < 1e4 else round(v) for k, v in out.items()}


bench = json.loads([l for l in open(root + "/bench.json") if l.startswith("{")][-1])

# Calibration of FETCH_SIZE for this access pattern (the guide calibrates it for wide streaming
# reads only, where it shows half the bytes): the bench's own gather microbenchmark on its 4 GB
# table fetches 1280 x 256 x 400 records of 64 bytes per launch, none of them twice.
gather = acc.get("k_gather<4>", {}).get("FETCH_SIZE", [])
known = 1280 * 256 * 400 * 64
calib = {"kernel": "k_gather<4>, 4 GB table (launches 13-16: the fourth pt_measure_gather call of the run)",
         "known_bytes_per_launch": known,
         "fetch_size_bytes_per_launch": int(sum(gather[12:16]) / 4 * 1024) if len(gather) >= 16 else None}
if calib["fetch_size_bytes_per_launch"]:
    calib["fetch_size_over_known"] = round(calib["fetch_size_bytes_per_launch"] / known, 4)
quad_sorted = "k_trace_ordered_persistent<false, false, false, true>"
for fname, prefix, label, rays in (
        ("r03_traffic.json", "k_pt_extend_persistent<false", "",
         bench["config"]["rays_per_step_per_gpu"] * bench["steps"] / bench["roofline"]["launches"]),
        ("r03_traffic_hbm.json", quad_sorted, " [uniform tessellation]", float(1 << 24)),
        ("r03_traffic_hbm_pole_fans.json", quad_sorted, " [pole fans]", float(1 << 24))):
    keys = [k for k in per_launch if k.startswith(prefix) and k.endswith(label or ">")]
    if not keys:
        print("no launches of", prefix)
        continue
    c = per_launch[keys[0]]
    # dependent 64-byte gathers: FETCH_SIZE is the byte count (calibration above; TCC_MISS x 64 B
    # agrees); the ray queue is a wide streaming read of 32 B per ray, of which it shows half
    traffic = c["FETCH_SIZE"] * 1024 + 16.0 * rays + c["WRITE_SIZE"] * 1024
    rec = {"kernel": keys[0],
           "command": "rocprofv3 --pmc <group> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (one group per run)",
           "rays_per_launch": round(rays),
           "fetch_size_kb_per_launch": round(c["FETCH_SIZE"]), "write_size_kb_per_launch": round(c["WRITE_SIZE"]),
           "l2_misses_x_64B_per_launch": int(c["TCC_MISS_sum"] * 64),
           "calibration": calib,
           "traffic_bytes_per_launch": int(traffic),
           "traffic_bytes_per_launch_if_every_fetch_were_a_streaming_read": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
           "counters": derived(c)}
    if fname == "r03_traffic.json":
        rec["workload"] = [262144, 1920, 1080, 8, 64, 0, False, "synthetic"]
    json.dump(rec, open(os.path.join(here, fname), "w"), indent=2)
    print(fname, json.dumps(rec["counters"]), "traffic GB/launch", rec["traffic_bytes_per_launch"] / 1e9)

for r in list(csv.DictReader(open(os.path.join(here, "r03_bench_kernel_stats.csv"))))[:6]:
    print(r["Name"][:80], r["Calls"], round(float(r["AverageNs"]) / 1e6, 4), "ms", r["Percentage"], "%")
print("bench under rocprof:", round(bench["value"]), "Mrays/s; extend", round(bench["roofline"]["avg_launch_ms"], 4),
      "ms by events; hbm record", round(bench["roofline_hbm"]["avg_launch_ms"], 3), "ms by events")
