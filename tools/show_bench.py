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
            print("   hbm record: %.2f ms/launch  %.0f Mrays/s kernel, %.0f incl. sort  frac %s of %s %s  (queue order %.2f ms)" % (
                h["avg_launch_ms"], h["mrays_per_s_kernel"], h["mrays_per_s_call_incl_sort"],
                "%.3f" % h["frac"] if h.get("frac") is not None else "-",
                "%.1f" % h["peak"] if h.get("peak") is not None else "-", h.get("unit", ""),
                h["queue_order"]["avg_launch_ms"]))
            o = h.get("same_surface_other_tessellation")
            if o:
                print("   other tessellation: %.2f ms/launch (queue order %.2f ms), %.1f slab tests per ray" % (
                    o["avg_launch_ms"], o["queue_order"]["avg_launch_ms"], o["nodes_per_ray"]))
        c = d.get("cpu_baseline")
        if c:
            print("   cpu baseline: %.2f %s on %d cores" % (c["value"], c["unit"], c["cores"]))
