"""Copies the summaries of a profile collection (gpurun_out/*_<tag>, written by the collect_*.sh
scripts) into profiles/r01_* and refreshes the derived numbers in profiles/r01_traffic.json.
    python profiles/install_profiles.py <tag>"""
import csv, json, os, re, shutil, sys

tag = sys.argv[1]
here = os.path.dirname(os.path.abspath(__file__))
out = os.path.join(os.path.dirname(here), "gpurun_out")


def grab(path, kernel):
    vals, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
            continue
        m = re.match(r"\s+(\S+)\s+total\s+(\S+)\s+per-dispatch\s+(\S+)", line)
        if m and cur == kernel:
            vals[m.group(1)] = float(m.group(3))
    return vals


shutil.copy(f"{out}/prof_{tag}/bench_kernel_stats.csv", f"{here}/r01_bench_kernel_stats.csv")
with open(f"{here}/r01_bench_under_rocprof.json", "w") as f:
    f.writelines(l for l in open(f"{out}/prof_{tag}/bench.json") if l.startswith("{"))
shutil.copy(f"{out}/traffic_{tag}/summary.txt", f"{here}/r01_traffic_pmc_summary.txt")
shutil.copy(f"{out}/pmc_{tag}/summary.txt", f"{here}/r01_pmc_summary.txt")
shutil.copy(f"{out}/pmcmem_{tag}/summary.txt", f"{here}/r01_pmc_mem_summary.txt")
if os.path.isdir(f"{out}/next_{tag}"):
    shutil.copy(f"{out}/next_{tag}/nee/nee_kernel_stats.csv", f"{here}/r01_nee_kernel_stats.csv")
    shutil.copy(f"{out}/next_{tag}/inst/inst_kernel_stats.csv", f"{here}/r01_two_level_kernel_stats.csv")
    with open(f"{here}/r01_nee_bench_under_rocprof.json", "w") as f:
        f.writelines(l for l in open(f"{out}/next_{tag}/nee.json") if l.startswith("{"))
    with open(f"{here}/r01_two_level_bench.json", "w") as f:
        for name in ("inst.json", "inst16.json"):
            f.writelines(l for l in open(f"{out}/next_{tag}/{name}") if l.startswith("{"))

k = "k_pt_extend_persistent<false>"
a = grab(f"{here}/r01_pmc_summary.txt", k)
b = grab(f"{here}/r01_pmc_mem_summary.txt", k)
t = grab(f"{here}/r01_traffic_pmc_summary.txt", k)
cyc = a["GRBM_GUI_ACTIVE"] / 8          # the counter sums the 8 XCDs
d = json.load(open(f"{here}/r01_traffic.json"))
d["fetch_size_kb_per_launch"] = round(t["FETCH_SIZE"])
d["write_size_kb_per_launch"] = round(t["WRITE_SIZE"])
d["traffic_bytes_per_launch"] = int((2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024)
d["counters"].update({
    "l2_hit_rate": round(a["TCC_HIT_sum"] / a["TCC_REQ_sum"], 3),
    "valu_busy": round(a["SQ_INSTS_VALU"] * 4 / (1024 * cyc), 3),
    "wave_lifetime_waiting_on_memory": round(a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], 3),
    "l1_line_lookups_per_cu_clock": round(a["TCP_TOTAL_CACHE_ACCESSES_sum"] / (256 * cyc), 3),
    "address_unit_busy": round(b["TA_TA_BUSY_sum"] / (256 * cyc), 3),
    "data_return_unit_busy": round(b["TD_TD_BUSY_sum"] / (256 * cyc), 3),
    "l1_clock_enabled": round(b["TCP_GATE_EN1_sum"] / (256 * cyc), 3),
})
json.dump(d, open(f"{here}/r01_traffic.json", "w"), indent=2)
rows = list(csv.DictReader(open(f"{here}/r01_bench_kernel_stats.csv")))
for r in rows[:5]:
    print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e6, 4), "ms", r["Percentage"], "%")
for l in open(f"{here}/r01_bench_under_rocprof.json"):
    j = json.loads(l)
    print("bench under rocprof:", round(j["value"]), "Mrays/s", round(j["ms_per_step"], 3), "ms/step, extend",
          round(j["roofline"]["avg_launch_ms"], 4), "ms by events; boxes/ray", round(j["nodes_per_ray"], 2))
print("launch by counters:", round(cyc / 2.4e6, 4), "ms;", json.dumps(d["counters"]))
print("traffic GB/launch:", d["traffic_bytes_per_launch"] / 1e9)
