"""Sums rocprofv3 --pmc counter CSVs per kernel name (one directory per counter group)."""
import csv
import glob
import re
import collections
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for path in sorted(glob.glob(root + "/g*/**/*counter_collection.csv", recursive=True)):
    with open(path) as f:
        for row in csv.DictReader(f):
            m = re.search(r"\bk_\w+(<\w+>)?", row["Kernel_Name"])
            name = m.group(0) if m else row["Kernel_Name"][:40]
            acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[name][row["Counter_Name"]] += 1
for name in sorted(acc):
    if not name.startswith("k_"):
        continue
    print(name)
    for c in sorted(acc[name]):
        n = calls[name][c]
        print(f"    {c:34s} total {acc[name][c]:.6g}   per-dispatch {acc[name][c] / n:.6g}   ({n} dispatches)")
