#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (profiles/run_pmc.sh): per kernel, mean counter value per
dispatch.  usage: pmc_summary.py <dir> [kernel-substring ...]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
keys = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(d + "/pass*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(d + "/pass1/*/*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc, key=lambda k: -sum(dur.get(k, [0]))):
    if keys and not any(s in k for s in keys):
        continue
    n = len(dur.get(k, []))
    print(f"== {k[:90]}  dispatches={n} avg_us={sum(dur.get(k,[0]))/max(n,1):.1f}")
    for c, v in sorted(acc[k].items()):
        print(f"   {c:32s} {sum(v)/len(v):16.1f}")
