"""Mean per-launch value of every counter in a rocprofv3 --pmc counter_collection.csv, grouped by kernel.
usage: python tools/pmc_summary.py <dir-or-csv> [name filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if flt and flt not in name:
            continue
        acc[name[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, ctrs in sorted(acc.items()):
    print(name)
    for c, vals in sorted(ctrs.items()):
        print("    %-32s %16.0f   (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
