#!/usr/bin/env python3
"""Kernel timeline of two iterations from a rocprofv3 --kernel-trace CSV (queue per kernel, gaps).
usage: trace_iter.py <dir> [k]   (k-th node_residual launch; default -6 = inside the last graph replays)"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_node_residual" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else -6      # which iteration (index of its node_residual launch)
i0, i1 = idx[k] - 3, idx[k + 2]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1 + 1]:
    m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
    nm = m.group(1) if m else r["Kernel_Name"][:40]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:8.1f} {e/1e3:8.1f} {(e-s)/1e3:7.1f} q={r.get('Queue_Id','?')} {nm}")
