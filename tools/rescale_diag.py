"""Diagnostic: per-tensor / per-column error of grad_theta against the oracle (float64 sums), relative to sum|terms|,
for strain profiles that sweep |g_z| over many octaves along the bar."""
import os, sys
os.environ.setdefault("PINNFEM_QUIET", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import orc
from test_hip_parity import _chain_model, _theta_tensors_like
from pinn_fem_amd.engine import HipEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
fe = int(sys.argv[2]) if len(sys.argv) > 2 else 1        # element-force formulation: 0 reference, 1 delta
hh = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0 / n    # element length (1.0: layer-1 units saturate at the far end)
names = ["E.W1", "E.b1", "E.W2", "E.b2", "E.Wo", "E.bo", "A.W1", "A.b1", "A.W2", "A.b2", "A.Wo", "A.bo"]
for span in (0.0, 4.0, 10.0):
    for direction in ("rising", "falling"):
        for wg, blocks in ((3, 2), (3, 1024), (2, 1024)):
            model, pb, mv, md = _chain_model(n, h=hh)
            eng = HipEngine(model, mv, md, n_part_blocks=blocks, wg_mode=wg, fe_mode=fe)
            e = np.arange(n, dtype=np.float64)
            expo = -span + 2 * span * e / (n - 1)
            if direction == "falling":
                expo = expo[::-1]
            strain = np.exp2(expo) * (1.0 + 0.3 * np.sin(0.37 * e))
            u = np.zeros(2 * (n + 1), dtype=np.float32)
            u[2::2] = np.cumsum(strain).astype(np.float32)
            losses, gu, gt = eng.loss_and_grads(torch.from_numpy(u), 0.6, 1.0, 0.0)
            gt = gt.cpu().numpy().copy()
            ref = orc.loss_and_grads(pb, orc.element_geometry(pb), u, 0.6, 1.0, 0.0, acc64=True, fe_mode='delta' if fe else 'reference')
            active = [g for g in ref.grad_theta if g is not None]
            row = []
            for k, (got, want, asum) in enumerate(zip(_theta_tensors_like(ref.grad_theta, gt), active, ref.grad_theta_abs)):
                asum = asum.reshape(want.shape)
                r = np.abs(got - want) / np.maximum(asum, 1e-300)
                if want.ndim == 2 and want.shape[1] == 3:
                    row.append(f"{names[k]} cols " + "/".join(f"{r[:, c].max():.1e}" for c in range(3)))
                else:
                    row.append(f"{names[k]} {r.max():.1e}")
            print(f"span 2^+-{span:g} {direction:8s} wg {wg} blocks {blocks:5d}: " + "  ".join(row), flush=True)
