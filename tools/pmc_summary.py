#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSVs (counter_collection): per kernel name, mean counter value per dispatch."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(f"{root}/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        name = r.get("Kernel_Name", "")[:40]
        acc[name][r.get("Counter_Name")].append(float(r.get("Counter_Value", 0)))
for name, cs in sorted(acc.items()):
    if not any(k in name for k in ("lstm", "linear", "stft", "istft", "kmeans")):
        continue
    print(name)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} n={len(v):6d} mean={sum(v) / len(v):14.1f} total={sum(v):16.1f}")
