#!/usr/bin/env python3
"""Average the counters of a rocprofv3 --pmc CSV per kernel.  usage: pmc_summary.py <dir> [kernel substring]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        ours = "_GLOBAL__N_" in k or (k.startswith("void (anonymous namespace)::") and "at::native" not in k)
        if sub in k and (sub or ours):   # default: this library's kernels only
            acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {sum(v) / len(v):16.1f}  (n={len(v)})")
