#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel name: python tools/pmc_summary.py <counter_collection.csv>..."""
import csv, sys, collections, re
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            m = re.search(r"(k_\w+(<[^>]*>)?)", name)
            short = m.group(1) if m else name[:40]
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
            dur[short].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    print("==", path)
    for k in sorted(acc, key=lambda k: -sum(dur[k])):
        if not k.startswith("k_"):
            continue
        d = dur[k]
        print(f"{k:38s} n={len(d)//max(len(acc[k]),1):3d} dur_us={sum(d)/len(d)/1e3:8.2f}  " +
              "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(acc[k].items())))
