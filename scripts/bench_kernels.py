#!/usr/bin/env python3
"""Print the per-kernel table of a bench.py JSON line (file argument)."""
import json
import sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["unit"], d["roofline"])
for k in d["kernels"]:
    print("  %-28s n=%-4d avg %8.2f us  %s GB/s" % (k["name"], k["launches"], k["avg_us"], k["gbs"]))
