#!/usr/bin/env python3
"""Kernel timeline of two graph-replayed iterations from a rocprofv3 --kernel-trace CSV (queue per kernel, gaps)."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_node_residual" in r["Kernel_Name"]]
i0, i1 = idx[22] - 3, idx[24]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1 + 1]:
    m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
    nm = m.group(1) if m else r["Kernel_Name"][:40]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:8.1f} {e/1e3:8.1f} {(e-s)/1e3:7.1f} q={r.get('Queue_Id','?')} {nm}")
