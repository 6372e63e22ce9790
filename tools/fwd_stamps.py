"""Diagnostic: per-wave start/end stamps of the MFMA32 forward kernel (PF_N32_DBG=16)."""
import os, sys
os.environ["PF_N32_DBG"] = "16"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model
from pinn_fem_amd.engine import HipEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
model, mv, md, widths = build_model(n, "ex4")
eng = HipEngine(model, mv, md, wg_mode=3)
for rep in range(3):
    eng.eval_properties(0.1)
torch.cuda.synchronize()
raw = eng.partials.cpu().numpy().view(np.uint64)
for which in (0, 1):
    d = raw[which * 65536: which * 65536 + 4096 * 4].reshape(-1, 4)
    d = d[d[:, 1] > 0]
    r0, r1, cyc, meta = d[:, 0].astype(np.float64), d[:, 1].astype(np.float64), d[:, 2].astype(np.float64), d[:, 3]
    t0 = r0.min()
    start, end = (r0 - t0) / 100.0, (r1 - t0) / 100.0   # us (100 MHz)
    ntask = (meta >> np.uint64(48)).astype(int)
    xcc = ((meta >> np.uint64(32)) & np.uint64(0xf)).astype(int)
    hw = (meta & np.uint64(0xffffffff)).astype(np.int64)
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3; waveid = hw & 0xf
    print(f"net {which}: waves {len(d)}, start min/med/max {start.min():.1f}/{np.median(start):.1f}/{start.max():.1f} us, "
          f"end min/med/max {end.min():.1f}/{np.median(end):.1f}/{end.max():.1f} us, life med {(np.median(end - start)):.1f} us, "
          f"clock med {np.median(cyc / ((r1 - r0) * 10.0)):.2f} GHz, tasks {ntask.min()}..{ntask.max()}")
    key = xcc * 1000 + se * 100 + cu * 4 + simd
    uniq, cnt = np.unique(key, return_counts=True)
    print(f"   distinct (xcc,se,cu,simd) slots {len(uniq)}, waves per slot min/med/max {cnt.min()}/{int(np.median(cnt))}/{cnt.max()}; "
          f"per-xcc wave counts {np.bincount(xcc, minlength=8)}")
    # how many waves are alive at mid time
    mid = np.median(end) / 2
    print(f"   alive at t={mid:.1f}us: {int(((start <= mid) & (end >= mid)).sum())}")
