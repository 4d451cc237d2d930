"""Clock a kernel really ran at, from a rocprofv3 --pmc GRBM_GUI_ACTIVE pass: GRBM_GUI_ACTIVE / 8 XCDs / wall time of the
dispatch (MI355X_MICROARCH.md, 'DVFS give-back'; reads high on dispatches shorter than ~0.3 ms).  Median per kernel.
usage: python tools/effective_clock.py <dir-or-csv> [name filter]"""
import csv
import glob
import os
import statistics
import sys
from collections import defaultdict

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(list)
for f in files:
    for row in csv.DictReader(open(f)):
        if row.get("Counter_Name") != "GRBM_GUI_ACTIVE":
            continue
        name = row.get("Kernel_Name", "")
        if flt and flt not in name:
            continue
        ns = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        if ns > 0:
            acc[name[:70]].append((float(row["Counter_Value"]) / 8.0 / ns, ns))
for name, v in sorted(acc.items(), key=lambda kv: -sum(t for _, t in kv[1])):
    ghz = statistics.median(c for c, _ in v)
    us = statistics.median(t for _, t in v) / 1e3
    print("%-70s  n=%4d  median %8.1f us  %5.2f GHz" % (name, len(v), us, ghz))
