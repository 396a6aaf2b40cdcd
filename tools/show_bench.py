"""Prints the main numbers of bench.py JSON lines.   python tools/show_bench.py file.json ..."""
import json
import sys

for path in sys.argv[1:]:
    for line in open(path):
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        r = d["roofline"]
        print("%s: %.1f %s  %.2f ms/step  kernels %s  extend %.4f ms x %d  frac %.3f of %.0f %s" % (
            path, d["value"], d["unit"], d["ms_per_step"],
            {k: round(v, 2) for k, v in d.get("kernel_ms_per_step", {}).items()},
            r["avg_launch_ms"], r["launches"], r["frac"], r["peak"], r["unit"]))
        h = d.get("roofline_hbm")
        if h:
            print("   hbm record: %.2f ms/launch  %.0f Mrays/s kernel, %.0f incl. sort  frac %.3f  (queue order %.2f ms, frac %.3f)" % (
                h["avg_launch_ms"], h["mrays_per_s_kernel"], h["mrays_per_s_call_incl_sort"], h["frac"],
                h["queue_order"]["avg_launch_ms"], h["queue_order"]["frac"]))
        c = d.get("cpu_baseline")
        if c:
            print("   cpu baseline: %.2f %s on %d cores" % (c["value"], c["unit"], c["cores"]))
