"""Debug aid: MFMA32 engine (wg 3) against the exact-f32 4x4x1 engine (wg 2) on the same chain problem:
per-element properties, then loss and gradients per parameter tensor."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model
from pinn_fem_amd.engine import HipEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
workload = sys.argv[2] if len(sys.argv) > 2 else "ex4"
res = {}
for wg in (2, 3):
    model, mv, md, widths = build_model(n, workload)
    eng = HipEngine(model, mv, md, wg_mode=wg, mlp_dtype=(os.environ.get("DBG_DTYPE", "f32") if wg == 3 else None))
    eng.eval_properties(0.1)
    torch.cuda.synchronize()
    E = eng.prop_e[:n].cpu().numpy().copy(); A = eng.prop_a[:n].cpu().numpy().copy()
    rng = np.random.default_rng(0)
    u = (rng.normal(size=eng.plan.n_dofs) * 0.01).astype(np.float32)
    u[eng.plan.fixed_dofs] = 0
    losses, gu, gt = eng.loss_and_grads(torch.from_numpy(u), 0.1, 1.0, 100.0)
    torch.cuda.synchronize()
    res[wg] = dict(E=E, A=A, loss=losses["loss_total"], gu=gu.cpu().numpy().copy(), gt=gt.cpu().numpy().copy(),
                   off=eng.theta.tensor_off)
a, b = res[2], res[3]
def rel(x, y):
    return float(np.max(np.abs(x - y)) / max(np.max(np.abs(x)), 1e-30))
print("E rel", rel(a["E"], b["E"]), "A rel", rel(a["A"], b["A"]))
bad = np.argsort(-np.abs(a["E"] - b["E"]))[:8]
print("worst E elems", bad, a["E"][bad], b["E"][bad])
print("E[:6]", a["E"][:6], b["E"][:6])
print("E[60:70]", a["E"][60:70], b["E"][60:70])
print("loss", a["loss"], b["loss"], "grad_u rel", rel(a["gu"], b["gu"]))
off = a["off"]
for t in range(len(off) - 1):
    lo, hi = off[t], off[t + 1]
    if hi <= len(a["gt"]):
        x, y = a["gt"][lo:hi], b["gt"][lo:hi]
        print(f"tensor {t} [{lo}:{hi}] rel {rel(x, y):.3e}  ref[:4] {x[:4]}  got[:4] {y[:4]}")
