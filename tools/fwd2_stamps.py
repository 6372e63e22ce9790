"""Diagnostic (PINNFEM_N32_DBG=1 build, PF_N32_DBG=16): where the fused forward launch of the iteration graph spends its
time — per-wave s_memrealtime stamps (100 MHz) written into pf_problem.u_alt: 0 entry | 1 update loads + Adam done | 2 first
barrier passed | 3 operand images packed | 4 prologue left | 5 task loop left."""
import os, sys
os.environ["PF_N32_DBG"] = "16"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model
from pinn_fem_amd.engine import HipEngine
from pinn_fem_amd.fem.solver import SolverConfig
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
model, mv, md, _ = build_model(n, "ex4")
eng = HipEngine(model, mv, md)
eng.begin(None, 0.1, SolverConfig(max_iterations=10**6, tolerance=0.0, learning_rate_u=0.01, learning_rate_theta=5e-4))
eng.iterate(3 * eng.GRAPH_ITERS)
torch.cuda.synchronize()
raw = eng.u_alt.cpu().numpy().view(np.uint64)[: 256 * 16 * 8].reshape(-1, 8).astype(np.float64)
raw = raw[raw[:, 0] > 0]
t0 = raw[:, 0].min()
us = (raw[:, :6] - t0) / 100.0
names = ["entry", "update done", "barrier 1", "packed", "prologue left", "loop left"]
for k, nm in enumerate(names):
    c = us[:, k]
    print(f"{nm:14s} min {c.min():7.2f}  med {np.median(c):7.2f}  max {c.max():7.2f} us")
blk = us.reshape(-1, 16, 6) if len(us) % 16 == 0 else None
if blk is not None:
    print("block 0 (lead): prologue left at", blk[0, :, 4].max().round(2), "loop left at", blk[0, :, 5].max().round(2),
          "| other blocks: prologue left med", np.median(blk[1:, :, 4].max(1)).round(2), "loop left med",
          np.median(blk[1:, :, 5].max(1)).round(2), "max", blk[1:, :, 5].max().round(2))

# the fused backward launch (same build / knob): stamps behind the forward's, 8 per wave
rawb = eng.u_alt.cpu().numpy().view(np.uint64)[65536: 65536 + 256 * 8 * 8].reshape(-1, 8).astype(np.float64)
rawb = rawb[rawb[:, 0] > 0]
if len(rawb):
    tb = rawb[:, 0].min()
    usb = (rawb[:, :7] - tb) / 100.0
    for k, nm in enumerate(["bwd entry", "E loop in", "E loop out", "E row out", "A loop in", "A loop out", "A row out"]):
        c = usb[:, k]
        print(f"{nm:14s} min {c.min():7.2f}  med {np.median(c):7.2f}  max {c.max():7.2f} us")
    blkb = usb.reshape(-1, 8, 7) if len(usb) % 8 == 0 else None
    if blkb is not None:
        endb = blkb[:, :, 6].max(1)
        print("blocks end: min", endb.min().round(2), "med", np.median(endb).round(2), "max", endb.max().round(2),
              "| first 160 blocks med", np.median(endb[:160]).round(2), "| last 90 blocks med", np.median(endb[-90:]).round(2))
        a_out = blkb[:, :, 5]
        print("A loop left, per block: spread (max - min over its 8 waves) med", np.median(a_out.max(1) - a_out.min(1)).round(2),
              "| elder waves (0-3) med", np.median(a_out[:, :4]).round(2), "| younger (4-7) med", np.median(a_out[:, 4:]).round(2))
        e_out = blkb[:, :, 2]
        print("E loop left: elder med", np.median(e_out[:, :4]).round(2), "| younger med", np.median(e_out[:, 4:]).round(2))
        print("per wave id, A loop left med:", [float(np.median(a_out[:, w]).round(1)) for w in range(8)])
