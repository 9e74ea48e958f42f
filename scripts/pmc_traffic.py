#!/usr/bin/env python3
"""Per-kernel HBM traffic from rocprofv3 --pmc passes (MI355X_MICROARCH.md "HBM": hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950,
FETCH_SIZE / WRITE_SIZE in KiB, the factor 2 because gfx950 tallies 128-B read requests at 64 B).

usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> > profiles/rNN_pmc_traffic.json
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

# bench.py kernel label -> substring of the kernel symbol
LABELS = {"mlp_q4g": "k_mlp_q4g", "attn+o_proj": "k_attn2", "gemv_rows<lm_head+argmax>": "k_gemv_rows", "gemv_q4g<norm>": "k_gemv_q4g_slim",
          "gemv_q4g<plain>": "k_gemv_q4g<0", "gemv_q4g<silu>": "k_gemv_q4g<2"}


def collect(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "").strip()
                a = acc[name]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return acc


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for name in sorted(set(fetch) | set(write)):
        nf, sf = fetch.get(name, [0, 0.0])
        nw, sw = write.get(name, [0, 0.0])
        f = sf / nf if nf else 0.0
        w = sw / nw if nw else 0.0
        out[name] = {"dispatches": max(nf, nw), "FETCH_SIZE_KiB_avg": round(f, 2), "WRITE_SIZE_KiB_avg": round(w, 2),
                     "hbm_bytes_per_launch": round((2 * f + w) * 1024)}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("bz_kernels.hip", "bz_internal.h", "bz_dev.h"):      # the same stamp bench.py computes: a pass taken on other kernel sources is refused there
        h.update(open(os.path.join(root, "blazr_amd", "csrc", f), "rb").read())
    json.dump({"formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950 correction, MI355X_MICROARCH.md HBM section)", "kernels_sha16": h.hexdigest()[:16],
               "labels": LABELS, "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
