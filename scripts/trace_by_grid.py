#!/usr/bin/env python3
"""Per-(kernel, grid) average durations from a rocprofv3 --kernel-trace CSV: usage trace_by_grid.py <dir>"""
import csv, glob, os, sys, re
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        name = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").strip()
        key = (name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Workgroup_Size_X"]))
        a = acc[key]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
for (name, g, w), (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if n >= 8: print("%-60s grid %5d x %3d  n %5d  avg %8.2f us" % (name[:60], g, w, n, t / n))
