"""fp32 vs bf16 residual-tolerance study (BASELINE.json configs[4]): the hybrid examples 7 / 7-P (three NNs; with
network materials the "NR" phase of solve_hybrid is GD again, FEM/python/fem/solver.py:594-651) solved with the MLP
matrix products in float32-grade arithmetic (2-way split f16 operands, the default) and in plain bf16 (f32
accumulate), same initial parameters.  Reports per run: iterations per load increment, converged flag, final
residual norm and loss, and the deviation of the bf16 run from the float32 run in displacements and identified E*A
at the element centroids; plus the fixed-iteration throughput of both variants on the 10^6-element ex4-shape chain
and, on that chain, how far bf16 operands move one evaluation of properties, residual and gradients.
    python tools/bf16_study.py [out.json]"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("PINNFEM_QUIET", "1")
from helpers import load_run, product_example
from pinn_fem_amd.fem.solver import solve, SolverConfig
from pinn_fem_amd.cli.generic import extract_nn_properties

def run_example(ex, dtype):
    run = load_run(ex)
    theta0 = [np.array(t, dtype=np.float32) for t in run["theta0"]]
    parsed = product_example(ex, theta0)
    model = parsed["model"]
    model._pf_mlp_dtype = dtype
    md = parsed["measured_data"]
    # per-increment iteration counts: wrap solve_gd through the history lengths of every call
    from pinn_fem_amd.fem import solver as S
    calls = []
    orig = S.solve_gd
    depth = [0]
    def spy(*a, **k):
        depth[0] += 1
        try:
            r = orig(*a, **k)
        finally:
            depth[0] -= 1
        if depth[0] == 0:            # outermost call of a load increment (two-phase runs call themselves)
            calls.append(len(r.history))
        return r
    S.solve_gd = spy
    try:
        t0 = time.perf_counter()
        res = solve(model, parsed["solver_config"], md.get("values"), md.get("dofs"))
        wall = time.perf_counter() - t0
    finally:
        S.solve_gd = orig
    ident = {}
    full = extract_nn_properties(model)
    for name in ("young", "area"):
        p = full.get(name, {})
        if "load_factor_variations" in p:
            ident[name] = p["load_factor_variations"]["load_factor_1.0"]["at_elements"]["values"]
        elif "at_elements" in p:
            ident[name] = p["at_elements"]["values"]
    if "young" in ident and "area" in ident:
        ident["EA"] = (np.asarray(ident["young"]) * np.asarray(ident["area"])).tolist()
    return dict(example=ex, dtype=dtype, converged=bool(res.converged), iterations_total=len(res.history),
                calls=calls, final_residual_norm=res.history[-1]["residual_norm"], final_loss=res.history[-1]["loss_total"],
                u=res.displacements.flatten().tolist(), ident=ident, wall_s=wall,
                ref_iterations=run["result"]["iterations"], ref_u=run["result"]["displacements"])

def rel(a, b):
    a, b = np.asarray(a, dtype=float).reshape(-1), np.asarray(b, dtype=float).reshape(-1)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))

def ident_err(ia, ib):
    return {k: rel(ia[k], ib[k]) for k in ia if k in ib}

def throughput(dtype, n=1_000_000, steps=50):
    from bench import build_model
    from pinn_fem_amd.engine import HipEngine
    model, mv, mdofs, _ = build_model(n, "ex4")
    eng = HipEngine(model, mv, mdofs, mlp_dtype=dtype)
    cfg = SolverConfig(max_iterations=400, tolerance=0.0, learning_rate_u=0.01, learning_rate_theta=5e-4)
    eng.begin(None, 0.1, cfg, want_history=False)
    eng.prepare_graph(); eng.iterate(20); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); eng.iterate(steps); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    return dict(dtype=dtype, n_elems=n, ms_per_iteration=best * 1e3, evals_per_s=n / best)

def at_scale(n=1_000_000):
    """One evaluation of losses and gradients on the 10^6-element ex4-shape chain (the bench's mesh) from the same
    state with both operand types: how far plain bf16 operands move the properties (forward), the residual and the
    gradients (backward) — the per-iteration perturbation behind the run-level differences above."""
    from bench import build_model
    from pinn_fem_amd.engine import HipEngine
    x = np.arange(n + 1, dtype=np.float64)
    u = np.zeros(2 * (n + 1), dtype=np.float32)
    u[0::2] = (1e-3 * x * (1.0 + 0.05 * np.sin(x / 50.0))).astype(np.float32)
    res = {}
    for dtype in ("f32", "bf16"):
        torch.manual_seed(0)
        model, mv, mdofs, _ = build_model(n, "ex4")
        eng = HipEngine(model, mv, mdofs, mlp_dtype=dtype, fe_mode=1)
        losses, gu, gt = eng.loss_and_grads(torch.from_numpy(u), 0.6)
        res[dtype] = dict(losses=losses, gu=gu.cpu().numpy().copy(), gt=gt.cpu().numpy().copy(),
                          e=eng.prop_e[:n].cpu().numpy().copy(), a=eng.prop_a[:n].cpu().numpy().copy())
    f, b = res["f32"], res["bf16"]
    return dict(n_elems=n, state="u_x = 1e-3 x (1 + 0.05 sin(x/50)), load factor 0.6, fe_mode delta",
                bf16_vs_f32=dict(young=rel(b["e"], f["e"]), area=rel(b["a"], f["a"]),
                                 loss_total=abs(b["losses"]["loss_total"] / f["losses"]["loss_total"] - 1.0),
                                 residual_norm=abs(b["losses"]["residual_norm"] / f["losses"]["residual_norm"] - 1.0),
                                 grad_u=rel(b["gu"], f["gu"]), grad_theta=rel(b["gt"], f["gt"])))

def main():
    out = {"runs": [], "compare": [], "throughput": []}
    for ex in ("example7", "example7-P"):
        r32 = run_example(ex, "f32"); rbf = run_example(ex, "bf16")
        for r in (r32, rbf):
            out["runs"].append({k: v for k, v in r.items() if k not in ("u", "ident", "ref_u")})
        out["compare"].append(dict(example=ex,
            f32_vs_reference_u=rel(r32["u"], r32["ref_u"]), bf16_vs_reference_u=rel(rbf["u"], rbf["ref_u"]),
            bf16_vs_f32_u=rel(rbf["u"], r32["u"]), bf16_vs_f32_identified=ident_err(rbf["ident"], r32["ident"]),
            iterations=dict(reference=r32["ref_iterations"], f32=r32["iterations_total"], bf16=rbf["iterations_total"])))
    for dt in ("f32", "bf16"):
        out["throughput"].append(throughput(dt))
    out["at_scale"] = at_scale()
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt)

if __name__ == "__main__":
    main()
