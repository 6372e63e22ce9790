"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors produced by the
reference and against the oracle on the same inputs.  Tolerances are written next to each check;
everything is float32 arithmetic, so "parity" means agreement to float32 round-off of the quantity's
own scale (the reference's own summation order is not specified by torch either).
"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import (example_problem, load_npz, load_run, mesh_problem, orc, product_example,
                     product_model, rel_err, theta_from)

pytestmark = pytest.mark.gpu

WG_MODES = [3, 2, 1, 0]  # PF_WG_MFMA32 (default), PF_WG_MFMA44, PF_WG_MFMA, PF_WG_SHUFFLE


def _engine(model, mv, md, wg, fe=0):
    from pinn_fem_amd.engine import HipEngine
    return HipEngine(model, mv, md, wg_mode=wg, fe_mode=fe)


def _check_step(eng, rec, alpha_p, alpha_d, tol_f=2e-6, tol_g=2e-5, tol_t=2e-5):
    losses, gu, gt = eng.loss_and_grads(torch.from_numpy(rec["u"]), float(rec["lam"]), alpha_p, alpha_d)
    f_int = eng.internal_force(torch.from_numpy(rec["u"]), float(rec["lam"])).cpu().numpy()
    assert rel_err(f_int, rec["f_int"]) < tol_f
    assert abs(losses["loss_physics"] - rec["loss_physics"]) <= tol_g * max(abs(rec["loss_physics"]), 1e-30)
    assert abs(losses["loss_data"] - rec["loss_data"]) <= 2e-6 * max(abs(rec["loss_data"]), 1e-30)
    assert abs(losses["loss_total"] - rec["loss_total"]) <= tol_g * max(abs(rec["loss_total"]), 1e-30)
    assert abs(losses["residual_norm"] - rec["residual_norm"]) <= tol_g * max(abs(rec["residual_norm"]), 1e-30)
    assert rel_err(gu.cpu().numpy(), rec["grad_u"]) < tol_g
    gt = gt.cpu().numpy()
    ref = []
    i = 0
    while f"grad_theta_{i}" in rec:
        if not rec[f"grad_theta_{i}_is_none"]:
            ref.append(rec[f"grad_theta_{i}"].reshape(-1))
        i += 1
    if ref:
        ref = np.concatenate(ref)
        assert gt.shape == ref.shape
        # per-tensor scale differs by orders of magnitude: compare against the global max and
        # tensor-wise
        assert rel_err(gt, ref) < tol_t
    return losses


@pytest.mark.parametrize("wg", WG_MODES)
@pytest.mark.parametrize("ex", ["example3", "example4"])
@pytest.mark.parametrize("state", ["cold", "analytic", "mid"])
def test_single_step_examples(ex, state, wg):
    """f_int, losses, grad_u, grad_theta of one iteration vs the reference (solver.py:262-289)."""
    rec = load_npz(f"step_{ex}_{state}.npz")
    parsed = product_example(ex, theta_from(rec))
    md = parsed["measured_data"]
    cfg = parsed["solver_config"]
    eng = _engine(parsed["model"], md["values"], md["dofs"], wg)
    _check_step(eng, rec, cfg.alpha_physics, cfg.alpha_data)
    k = eng.dense_k(float(rec["lam"])).cpu().numpy()
    assert rel_err(k, rec["K"]) < 1e-6
    assert rel_err(eng.diag_k(float(rec["lam"])).cpu().numpy(), rec["K_diag"]) < 1e-6


@pytest.mark.parametrize("wg", WG_MODES)
@pytest.mark.parametrize("name,widths,scales,tf,tg,tt", [
    # long chains: u up to 700 with O(1) differences -> float32 cancellation in fe = ke@u_e; the
    # oracle itself sits 2e-5 / 2.5e-4 from the reference on these (summation order)
    ("step_chain300_ex4shape.npz", (20, 15, 10), (1.0, 1.0, 1.0), 1e-4, 1e-4, 1e-3),
    ("step_chain1000_ex4shape.npz", (20, 15, 10), (1.0, 1.0, 1.0), 1e-4, 1e-4, 1e-3),
    ("step_warren_EA.npz", (20, 15, None), (2.0, 0.5, 1.0), 2e-6, 2e-5, 2e-5),
    ("step_bar1d_E.npz", (20, None, None), (3.0, 2.0, 1.0), 2e-6, 2e-5, 2e-5),
])
def test_single_step_meshes(name, widths, scales, tf, tg, tt, wg):
    """Chains of 300/1000 elements, an irregular 2-D truss (node degree up to 5) and a 1-D bar."""
    rec = load_npz(name)
    dim = 1 if rec["nodes"].ndim == 1 else 2
    model = product_model(rec["nodes"], rec["elements"], rec["loads"], rec["fixed"], dim, widths, scales,
                          theta_from(rec))
    eng = _engine(model, rec["meas_vals"], rec["meas_dofs"], wg)
    _check_step(eng, rec, 1.0, 100.0, tol_f=tf, tol_g=tg, tol_t=tt)
    assert rel_err(eng.diag_k(float(rec["lam"])).cpu().numpy(), rec["K_diag"]) < 1e-6


def test_single_step_scalar():
    """example2 shape: scalar material, no theta, alpha_data = 0."""
    rec = load_npz("step_example2_scalar.npz")
    nodes = np.stack([np.arange(4, dtype=float), np.zeros(4)], axis=1)
    model = product_model(nodes, np.array([[0, 1], [1, 2], [2, 3]]), np.array([0, 0, 0, 0, 0, 0, 1.0, 0]),
                          np.array([0, 1, 3, 5, 7]), 2, (None, None, None), (1.0, 1.0, 1.0), [])
    eng = _engine(model, np.array([]), np.array([], dtype=int), 1)
    _check_step(eng, rec, 1.0, 0.0)


@pytest.mark.parametrize("ex", ["example3", "example4"])
@pytest.mark.parametrize("n_it", [1, 3, 12])
def test_first_adam_iterations(ex, n_it):
    """u, theta, reactions and history after 1/3/12 iterations of solve_gd at load factor 0.1
    (fresh torch.optim.Adam on u and theta, solver.py:234-298)."""
    from pinn_fem_amd.fem.solver import solve_gd
    rec = load_npz(f"adam_{ex}_it{n_it}.npz")
    parsed = product_example(ex, theta_from(rec, "theta0_"))
    cfg, md, model = parsed["solver_config"], parsed["measured_data"], parsed["model"]
    cfg.max_iterations = n_it
    res = solve_gd(model, cfg, md["values"], md["dofs"], target_load_factor=0.1)
    assert len(res.history) == n_it
    assert rel_err(res.displacements.flatten(), rec["u"]) < 5e-6
    assert rel_err(res.reactions.flatten(), rec["reactions"]) < 5e-6
    ref_theta = theta_from(rec)
    for i, ref in enumerate(ref_theta):
        got = res.nn_parameters[f"param_{i}"]
        assert got.shape == ref.shape
        assert rel_err(got, ref) < 5e-6
    for key, col in (("loss_total", "hist_loss_total"), ("residual_norm", "hist_residual_norm"),
                     ("u_norm", "hist_u_norm"), ("theta_norm", "hist_theta_norm"),
                     ("loss_data", "hist_loss_data"), ("loss_physics", "hist_loss_physics")):
        got = np.array([h[key] for h in res.history])
        assert rel_err(got, rec[col]) < 2e-5, key


def _leaf_counts(run):
    return [c["n_history"] for c in run["calls"]
            if not (c["preconditioning"] and not c["skip_preconditioning"])]


@pytest.mark.parametrize("ex", ["example2", "example2-P", "example3", "example3-P", "example4",
                                "example4-P", "example6", "example6-P", "example7", "example7-P"])
def test_whole_example_runs(ex, monkeypatch):
    """Whole `generic.py exampleN.json` runs (10 load increments) against the reference at seed 0:
    per-call iteration counts, converged flag, nodal displacements within 1e-5 relative
    (BASELINE.json north_star tolerance), reactions, and the identified E*A at the element centroids."""
    from pinn_fem_amd.cli import generic as g
    import pinn_fem_amd.fem.solver as S
    run = load_run(ex)
    theta0 = [np.array(t, dtype=np.float32) for t in run["theta0"]]
    parsed = product_example(ex, theta0 if theta0 else None)
    counts = []
    orig = S.solve_gd

    def wrapper(model, config=None, measured_disp=None, measured_dofs=None, target_load_factor=1.0,
                u_initial=None, skip_preconditioning=False):
        res = orig(model, config, measured_disp, measured_dofs, target_load_factor, u_initial,
                   skip_preconditioning)
        if not (config.preconditioning and not skip_preconditioning):
            counts.append(len(res.history))
        return res

    monkeypatch.setattr(S, "solve_gd", wrapper)
    out = g.solve_problem(parsed)
    ref = run["result"]
    ref_counts = _leaf_counts(run)
    assert out["converged"] == ref["converged"]
    # Every solve_gd call stops at EXACTLY the reference's iteration: measured on all ten examples (1283-4942 iterations
    # per run, 10-20 calls each; profiles/r03_parity_summary.json) and asserted as such — the device arithmetic is
    # deterministic (fixed-order sums, no float atomics), so a one-iteration drift would be a change of the arithmetic.
    assert counts == ref_counts, (counts, ref_counts)
    assert rel_err(out["displacements"], ref["displacements"]) < 1e-5
    assert np.max(np.abs(np.array(out["reactions"]) - np.array(ref["reactions"]))) < 1e-5
    if "identified_properties" in ref:
        # identified E and A (each on its own, and their product) at the element centroids for the three
        # load factors the reference reports: within 1e-5 relative (north-star tolerance); measured 7e-7
        for lf in ("load_factor_0.2", "load_factor_0.5", "load_factor_1.0"):
            ea_ref, ea_out = 1.0, 1.0
            for name in ("young", "area"):
                pr, po = ref["identified_properties"][name], out["identified_properties"][name]
                assert pr["type"] == po["type"]
                if pr["type"] == "scalar":
                    assert pr["value"] == po["value"]
                    ea_ref, ea_out = ea_ref * pr["value"], ea_out * po["value"]
                else:
                    vr = np.array(pr["load_factor_variations"][lf]["at_elements"]["values"])
                    vo = np.array(po["load_factor_variations"][lf]["at_elements"]["values"])
                    assert rel_err(vo, vr) < 1e-5, (name, lf)
                    nr = np.array(pr["load_factor_variations"][lf]["at_nodes"]["values"])
                    no = np.array(po["load_factor_variations"][lf]["at_nodes"]["values"])
                    assert rel_err(no, nr) < 1e-5, (name, lf)
                    ea_ref, ea_out = ea_ref * vr, ea_out * vo
            assert rel_err(ea_out, ea_ref) < 1e-5, lf
        assert set(out["nn_parameters"].keys()) == set(ref["nn_parameters"].keys())


def _chain_model(n, widths=(20, 15, 10), seed=0, h=1.0):
    from pinn_fem_amd.plan import chain_mesh
    from pinn_fem_amd.nets import SimpleNN
    nodes, elements, loads, fixed, mv, md = chain_mesh(n, h)
    torch.manual_seed(seed)
    theta = []
    for w in widths:
        if w is not None:
            theta += [p.detach().numpy().copy() for p in SimpleNN(2, w, 3).parameters()]
    model = product_model(nodes, elements, loads, fixed, 2, widths, (1.0, 1.0, 1.0), theta)
    pb = orc.Problem(nodes=nodes, elements=elements, loads=loads, fixed_dofs=fixed, dimension=2,
                     measured_vals=mv, measured_dofs=md,
                     **{k: (orc.NetParams([t.copy() for t in theta[6 * i:6 * i + 6]], 1.0) if w else 1.0)
                        for i, (k, w) in enumerate(zip(("young", "area", "density"), widths))})
    return model, pb, mv, md


# engines the oracle comparisons run on: the one the product ships (MFMA32, default) and one cross-check engine
ORACLE_WG = [3, 1]


@pytest.mark.parametrize("wg", ORACLE_WG)
@pytest.mark.parametrize("fe", [0, 1])
@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 63, 64, 65, 127, 129, 511, 513, 4097, 100_000])
def test_oracle_parity_chain_sizes(n, fe, wg):
    """HIP vs oracle on the synthetic chain (SURVEY §8d inputs, h = 3/n); sizes straddle tile (32), wave (64)
    and block (128/256/512) edges — the MFMA32 engine's tail handling (lanes past the end redo the last element
    and do not store / count) included.  fe=0 is the reference's operation order: fe = ke@u_e cancels in
    float32 once |ke||u| >> |fe| (long chains), so f_int is compared on the scale |ke||u| that
    sets its round-off and the derived quantities only while that noise is small (n <= 129).
    fe=1 (delta formulation) has no cancellation and is compared tightly at every size."""
    model, pb, mv, md = _chain_model(n, h=3.0 / n)
    x = np.arange(n + 1) * (3.0 / n)
    u = np.zeros(2 * (n + 1), dtype=np.float32)
    lam = 0.7
    u[0::2] = (lam * x * (1.0 + 0.05 * np.sin(7.0 * x))).astype(np.float32)
    u[0] = 0.0
    geo = orc.element_geometry(pb)
    mode = "delta" if fe else "reference"
    ref = orc.loss_and_grads(pb, geo, u, lam, 1.0, 100.0, fe_mode=mode)
    eng = _engine(model, mv, md, wg, fe)
    losses, gu, gt = eng.loss_and_grads(torch.from_numpy(u), lam, 1.0, 100.0)
    gu, gt = gu.cpu().numpy(), gt.cpu().numpy()
    f_int = eng.internal_force(torch.from_numpy(u), lam).cpu().numpy()
    if fe == 0:
        scale = max(np.max(np.abs(ref.stiffness)) * np.max(np.abs(u)), 1e-30)
        assert np.max(np.abs(f_int - ref.f_int)) < 4e-6 * scale
    else:
        assert np.max(np.abs(f_int - ref.f_int)) < 4e-6 * max(np.max(np.abs(ref.f_int)), 1e-30)
    assert abs(losses["loss_data"] - ref.loss_data) < 1e-5 * max(ref.loss_data, 1e-12)
    if fe == 1 or n <= 129:
        assert abs(losses["loss_physics"] - ref.loss_physics) < 1e-4 * max(ref.loss_physics, 1e-12)
        # fe=0: the float32 noise of f_int (|ke||u| * 2^-24) is amplified once more by |ke| in K^T g_f
        g_tol = 5e-5 * max(np.max(np.abs(ref.grad_u)), 1e-30)
        if fe == 0:
            g_tol = max(g_tol, 4e-6 * scale * np.max(np.abs(ref.stiffness)))
        assert np.max(np.abs(gu - ref.grad_u)) < g_tol
        ref_t = np.concatenate([g.reshape(-1) for g in ref.grad_theta if g is not None])
        assert rel_err(gt, ref_t) < 5e-4


@pytest.mark.parametrize("wg", ORACLE_WG)
def test_full_size_properties(wg):
    """BASELINE size (10^6 elements, example4 shape): size-independent properties of the HIP path.
    (1) element forces are self-equilibrated, so sum(f_int) = 0 up to round-off;
    (2) f_int is linear in u;
    (3) f_int(u) equals the oracle's on a bounded sample of nodes' neighbourhoods;
    (4) grad_u = K^T g: <g_f, K v> == <K^T g_f, v> (adjoint identity) for a random v."""
    n = 1_000_000
    model, pb, mv, md = _chain_model(n, h=3.0 / n)
    eng = _engine(model, mv, md, wg)
    x = np.arange(n + 1) * (3.0 / n)
    u = np.zeros(2 * (n + 1), dtype=np.float32)
    u[0::2] = (0.6 * x * (1.0 + 0.05 * np.sin(7.0 * x))).astype(np.float32)
    ut = torch.from_numpy(u)
    f1 = eng.internal_force(ut, 0.6).double()
    scale = float(eng.prop_e.abs().max() * eng.prop_a.abs().max() / (3.0 / n) * np.abs(u).max())
    assert abs(float(f1.sum())) < 1e-6 * scale * np.sqrt(n)
    f2 = eng.internal_force(2.0 * ut, 0.6).double()
    assert float((f2 - 2.0 * f1).abs().max()) < 1e-6 * scale
    # oracle on the first 5000 elements (same theta, same inputs): interior nodes must agree
    m = 5000
    sub = orc.Problem(nodes=pb.nodes[: m + 1], elements=pb.elements[:m], loads=pb.loads[: 2 * (m + 1)],
                      fixed_dofs=pb.fixed_dofs[pb.fixed_dofs < 2 * (m + 1)], dimension=2,
                      young=pb.young, area=pb.area, density=pb.density)
    geo = orc.element_geometry(sub)
    s, *_ = orc.element_stiffness(sub, geo, 0.6)
    f_ref = orc.internal_force(geo, s, u[: 2 * (m + 1)], 2 * (m + 1))
    got = f1[: 2 * m].float().cpu().numpy()
    assert np.max(np.abs(got - f_ref[: 2 * m])) < 4e-6 * scale
    # adjoint identity through the vjp entry point
    g = torch.randn(2 * (n + 1), generator=torch.Generator().manual_seed(1))
    v = torch.randn(2 * (n + 1), generator=torch.Generator().manual_seed(2))
    kv = eng.internal_force(v, 0.6).double().cpu()
    ktg, _ = eng.vjp(ut, g, 0.6)
    lhs = float((g.double() * kv).sum())
    rhs = float((ktg.double().cpu() * v.double()).sum())
    assert abs(lhs - rhs) < 1e-4 * max(abs(lhs), abs(rhs), 1e-30) + 1e-3 * scale


@pytest.mark.parametrize("wg", ORACLE_WG)
def test_autograd_function_matches_fused_gradients(wg, monkeypatch):
    """The torch.autograd.Function (assemble_system_torch seam) reproduces the fused loss gradients
    when the loss of solver.py:267-283 is written with torch ops on top of it."""
    from pinn_fem_amd.fem.nn_assembly import assemble_system_torch
    rec = load_npz("step_example4_mid.npz")
    parsed = product_example("example4", theta_from(rec))
    model, md = parsed["model"], parsed["measured_data"]
    monkeypatch.setenv("PINNFEM_WG_MODE", str(wg))        # the engine assemble_system_torch builds for itself
    eng = _engine(model, md["values"], md["dofs"], wg)
    losses, gu, gt = eng.loss_and_grads(torch.from_numpy(rec["u"]), float(rec["lam"]), 1.0, 100.0)
    gu, gt = gu.clone(), gt.clone()
    dev = eng.device
    u = torch.tensor(rec["u"], dtype=torch.float32, device=dev, requires_grad=True)
    k_global, f_int = assemble_system_torch(model, u, float(rec["lam"]))
    assert k_global.shape == (8, 8)
    free = torch.tensor(rec["free"], device=dev)
    f_ext = torch.tensor(model.loads, dtype=torch.float32, device=dev)
    r = f_int[free] - float(rec["lam"]) * f_ext[free]
    mv = torch.tensor(md["values"], dtype=torch.float32, device=dev)
    mdofs = torch.tensor(md["dofs"], device=dev)
    loss = 1.0 * 0.5 * torch.sum(r ** 2) + 100.0 * torch.mean((mv - u[mdofs]) ** 2)
    loss.backward()
    assert abs(loss.item() - rec["loss_total"]) < 2e-5 * abs(rec["loss_total"])
    assert rel_err(u.grad.cpu().numpy(), rec["grad_u"]) < 2e-5
    params = model.material.get_all_torch_params()
    got = torch.cat([p.grad.reshape(-1) for p in params[:12]]).cpu().numpy()
    assert rel_err(got, gt.cpu().numpy()) < 2e-5
    assert all(p.grad is None for p in params[12:])  # density net never evaluated


def test_sparse_stiffness_matches_dense_and_oracle():
    """k_global in coordinate format (pf_coo_k / LazyStiffness.to_sparse; SURVEY 7.1b `assemble_coo`): on the 300-element
    chain fixture it equals the oracle's dense matrix, the reference golden's diagonal and the engine's own dense view; on a 10^5-element
    bar (no dense matrix possible: 160 GB) its diagonal equals diag K and K u equals the matrix-free internal force."""
    rec = load_npz("step_chain300_ex4shape.npz")
    model = product_model(rec["nodes"], rec["elements"], rec["loads"], rec["fixed"], 2, (20, 15, 10), (1.0, 1.0, 1.0),
                          theta_from(rec))
    eng = _engine(model, rec["meas_vals"], rec["meas_dofs"], 3)
    lam = float(rec["lam"])
    ks = eng.sparse_k(lam)
    assert ks.is_sparse and ks.shape == (602, 602)
    dense = ks.to_dense().cpu().numpy()
    pb = mesh_problem(rec, (20, 15, 10), (1.0, 1.0, 1.0))
    assert rel_err(dense, orc.dense_stiffness(pb, orc.element_geometry(pb), lam)) < 2e-6
    assert rel_err(np.diag(dense), rec["K_diag"]) < 1e-6          # the reference's own diagonal
    assert rel_err(dense, eng.dense_k(lam).cpu().numpy()) < 1e-6
    # at scale
    from pinn_fem_amd.fem.nn_assembly import assemble_system_torch
    model2, pb, mv, md = _chain_model(100_000, h=0.5)
    u = torch.linspace(0, 1, 2 * 100_001, device="cuda")
    k_lazy, f_int = assemble_system_torch(model2, u, 0.4)
    ks2 = k_lazy.to_sparse()
    assert ks2._nnz() <= 16 * 100_000 and ks2.shape == (200_002, 200_002)
    diag = torch.zeros(200_002, device=ks2.device).index_add_(0, ks2.indices()[0][ks2.indices()[0] == ks2.indices()[1]],
                                                              ks2.values()[ks2.indices()[0] == ks2.indices()[1]])
    assert rel_err(diag.cpu().numpy(), k_lazy.diagonal().cpu().numpy()) < 2e-6
    ku = torch.sparse.mm(ks2, u.reshape(-1, 1).to(ks2.device)).reshape(-1)
    scale = float(ks2.values().abs().max() * u.abs().max())
    assert float((ku - f_int.detach().to(ku.device)).abs().max()) < 4e-6 * scale


def test_unsupported_shapes_fail_loudly():
    from pinn_fem_amd.nets import SimpleNN
    from pinn_fem_amd.fem.properties import NNProperty
    from pinn_fem_amd.fem.model import FEMModel, Material
    from pinn_fem_amd.engine import HipEngine
    nodes = np.array([[0.0, 0.0], [1.0, 0.0]])
    for kw in (dict(hidden_layers=2, neurons_per_layer=40, input_dim=3),
               dict(hidden_layers=4, neurons_per_layer=8, input_dim=3)):
        mat = Material(young=NNProperty(SimpleNN(**kw), input_dim=3), area=1.0)
        model = FEMModel(nodes, np.array([[0, 1]]), mat, np.zeros(4), np.array([0, 1, 3]))
        with pytest.raises(NotImplementedError):
            HipEngine(model)


def _run_hip_ranks(kind, world, tmp_path, port):
    import json
    import os
    import subprocess
    import sys
    from helpers import ROOT
    out = str(tmp_path / f"hip_{kind}_{world}.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PINNFEM_QUIET="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_hip_worker.py"), kind, out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-5000:]
    with open(out) as f:
        return json.load(f)


def test_sharded_hip_example_run_two_ranks_one_gpu(tmp_path):
    """The sharded HIP path (2 ranks sharing this GPU, gloo collectives) reproduces the reference's
    example4-P run: 3 elements split 2+1, interface node shared."""
    got = _run_hip_ranks("example4-P", 2, tmp_path, 29631)
    run = load_run("example4-P")
    assert got["converged"] == run["result"]["converged"]
    assert rel_err(got["u"], run["result"]["displacements"]) < 1e-5
    assert np.max(np.abs(np.array(got["reactions"]) - np.array(run["result"]["reactions"]))) < 1e-5
    assert abs(got["n_history"] - run["result"]["iterations"]) <= 3


@pytest.mark.parametrize("world,port", [(3, 29632), (4, 29636)])
def test_sharded_hip_chain_matches_single_engine(tmp_path, world, port):
    """300-element chain, 25 iterations: 3 and 4 ranks on one GPU == one engine (sum of shards == whole); with the test
    process itself that is at most five processes on the card."""
    from pinn_fem_amd.fem.solver import SolverConfig, solve_gd
    got = _run_hip_ranks("chain300", world, tmp_path, port)
    rec = load_npz("step_chain300_ex4shape.npz")
    model = product_model(rec["nodes"], rec["elements"], rec["loads"], rec["fixed"], 2, (20, 15, 10),
                          (1.0, 1.0, 1.0), theta_from(rec))
    cfg = SolverConfig(max_iterations=25, learning_rate_u=0.01, learning_rate_theta=5e-4, tolerance=1e-12)
    ref = solve_gd(model, cfg, rec["meas_vals"], rec["meas_dofs"], target_load_factor=0.7,
                   u_initial=torch.from_numpy(rec["u"]))
    assert got["n_history"] == 25
    assert rel_err(got["u"], ref.displacements.flatten()) < 2e-5
    assert rel_err(got["loss"], [h["loss_total"] for h in ref.history]) < 1e-4
    for k, v in ref.nn_parameters.items():
        assert rel_err(got["theta"][k], v.reshape(-1)) < 2e-5, k


def test_sharded_hip_warren_truss_matches_single_engine(tmp_path):
    """The 19-element Warren truss cut in two (2 ranks on one GPU, gloo): shared nodes carry several elements on
    each side, so phase A evaluates more than one interface element per shared node; 20 iterations == one engine."""
    from pinn_fem_amd.fem.solver import SolverConfig, solve_gd
    got = _run_hip_ranks("warren", 2, tmp_path, 29633)
    rec = load_npz("step_warren_EA.npz")
    model = product_model(rec["nodes"], rec["elements"], rec["loads"], rec["fixed"], 2, (20, 15, None),
                          (2.0, 0.5, 1.0), theta_from(rec))
    cfg = SolverConfig(max_iterations=20, learning_rate_u=1e-3, learning_rate_theta=5e-4, tolerance=1e-12)
    ref = solve_gd(model, cfg, rec["meas_vals"], rec["meas_dofs"], target_load_factor=0.7,
                   u_initial=torch.from_numpy(rec["u"]))
    assert got["n_history"] == 20
    assert rel_err(got["u"], ref.displacements.flatten()) < 2e-5
    assert rel_err(got["loss"], [h["loss_total"] for h in ref.history]) < 1e-4
    for k, v in ref.nn_parameters.items():
        assert rel_err(got["theta"][k], v.reshape(-1)) < 2e-5, k


def test_sharded_hip_random_truss_matches_single_engine(tmp_path):
    """A random connected truss with shuffled element order (40 nodes, 80 elements; rank boundaries cut through nodes of
    any degree, ghost rings several elements wide) on 3 ranks sharing this GPU == one engine, 20 iterations."""
    from dist_hip_worker import random_case
    from pinn_fem_amd.fem.solver import solve_gd
    got = _run_hip_ranks("rand3", 3, tmp_path, 29637)
    model, cfg, mv, md, lam = random_case(3)
    ref = solve_gd(model, cfg, mv, md, lam)
    assert got["n_history"] == 20
    assert rel_err(got["u"], ref.displacements.flatten()) < 2e-5
    assert rel_err(got["loss"], [h["loss_total"] for h in ref.history]) < 1e-4
    for k, v in ref.nn_parameters.items():
        assert rel_err(got["theta"][k], v.reshape(-1)) < 2e-5, k


def test_api_pinn_gd_identifies_stiffness(tmp_path):
    """api_pinn_gradient_descent.py end to end on the GPU: a 3-bar chain whose measured displacements
    correspond to E*A = 2; the identified product must move from the initial guess (1) towards 2 and
    the displacement field must fit the data.  (Arithmetic parity unpinned: the reference's callee
    does not exist; this pins the build's own behaviour.)"""
    import json
    from pinn_fem_amd.cli import api_pinn_gradient_descent as api
    data = {"nodes": [{"x": 0, "y": 0, "fixed": True}] + [{"x": float(k), "y": 0} for k in (1, 2, 3)],
            "elements": [{"nodes": [0, 1]}, {"nodes": [1, 2]}, {"nodes": [2, 3]}],
            "material": {"young": 1.0, "area": 1.0},
            "loads": [0, 0, 0, 0, 0, 0, 1.0, 0], "fixed_dofs_note": "uy via measured zeros",
            "measured_disp": [0.5, 0.0, 1.0, 0.0, 1.5, 0.0], "measured_dofs": [2, 3, 4, 5, 6, 7],
            "solver_config": {"max_iterations": 1500, "learning_rate": 0.01, "alpha": 1.0, "beta": 100.0,
                              "young_bounds": [0.1, 10.0], "area_bounds": [0.1, 10.0]}}
    fin, fout = tmp_path / "in.json", tmp_path / "out.json"
    fin.write_text(json.dumps(data))
    api.main(["api", str(fin), str(fout)])
    out = json.loads(fout.read_text())
    assert set(out) == {"displacements", "stresses", "strains", "identified_params", "convergence_history",
                        "final_loss"}
    ea = out["identified_params"]["young"] * out["identified_params"]["area"]
    assert 1.7 < ea < 2.3
    u = np.array(out["displacements"])
    assert np.max(np.abs(u[[2, 4, 6]] - [0.5, 1.0, 1.5])) < 0.05
    assert out["convergence_history"][0]["loss_total"] > out["final_loss"]
    assert len(out["convergence_history"]) == 150


@pytest.mark.parametrize("wg", [3, 2, 1])
@pytest.mark.parametrize("width,layers,dim", [(1, 1, 2), (4, 1, 2), (7, 2, 2), (8, 3, 2), (12, 2, 1), (16, 2, 2),
                                                (20, 3, 2), (24, 2, 2), (28, 1, 1), (32, 2, 2), (32, 3, 1), (15, 2, 2),
                                                (27, 2, 2), (30, 3, 1)])
def test_net_shape_menu(width, layers, dim, wg):
    """Every padded width (4..32), 1..3 hidden layers, 1-D (input [load_factor, x]) and 2-D meshes:
    HIP vs oracle on a 150-element mesh with a different net shape for E and A."""
    from pinn_fem_amd.fem.model import FEMModel, Material
    from pinn_fem_amd.fem.properties import NNProperty
    from pinn_fem_amd.nets import SimpleNN
    n = 150
    rng = np.random.default_rng(width * 10 + layers)
    in_dim = dim + 1
    torch.manual_seed(width + 100 * layers)
    net_e = SimpleNN(layers, width, in_dim)
    net_a = SimpleNN(max(1, layers - 1) if layers > 1 else 2, max(2, width // 2), in_dim)
    th_e = [p.detach().numpy().copy() for p in net_e.parameters()]
    th_a = [p.detach().numpy().copy() for p in net_a.parameters()]
    if dim == 2:
        ang = np.cumsum(rng.uniform(-0.3, 0.3, n))
        pts = np.concatenate([[[0.0, 0.0]], np.cumsum(np.stack([np.cos(ang), np.sin(ang)], 1) * 0.1, 0)])
        nodes = pts
        fixed = np.array([0, 1])
    else:
        nodes = np.concatenate([[0.0], np.cumsum(rng.uniform(0.05, 0.15, n))])
        fixed = np.array([0])
    elements = np.stack([np.arange(n), np.arange(1, n + 1)], 1)
    ndof = (n + 1) * dim
    loads = rng.normal(size=ndof) * 0.1
    u = (rng.normal(size=ndof) * 0.02).astype(np.float32)
    u[fixed] = 0
    md = rng.choice(np.arange(dim, ndof), size=40, replace=False)
    mv = rng.normal(size=40) * 0.02
    model = FEMModel(nodes, elements, Material(NNProperty(net_e, in_dim, True, 1.5), NNProperty(net_a, in_dim, True, 0.7)),
                     loads, fixed, dimension=dim)
    pb = orc.Problem(nodes=nodes, elements=elements, loads=loads, fixed_dofs=fixed, dimension=dim,
                     young=orc.NetParams(th_e, 1.5), area=orc.NetParams(th_a, 0.7), measured_vals=mv, measured_dofs=md)
    ref = orc.loss_and_grads(pb, orc.element_geometry(pb), u, 0.45, 1.0, 50.0)
    eng = _engine(model, mv, md, wg)
    losses, gu, gt = eng.loss_and_grads(torch.from_numpy(u), 0.45, 1.0, 50.0)
    assert abs(losses["loss_total"] - ref.loss_total) < 2e-5 * abs(ref.loss_total)
    assert rel_err(gu.cpu().numpy(), ref.grad_u) < 3e-5
    ref_t = np.concatenate([g.reshape(-1) for g in ref.grad_theta])
    assert rel_err(gt.cpu().numpy(), ref_t) < 1e-4


def test_cli_synthetic_chain_compact_output(tmp_path):
    """`generic.py` on a generated 200k-element chain (accel.synthetic_chain), compact npz output;
    the first increment's iterations are checked against the oracle on the same theta."""
    import json
    from pinn_fem_amd.cli import generic as g
    data = {"accel": {"synthetic_chain": {"n_elements": 200_000, "h": 1.0}, "fe_mode": "delta"},
            "material": {"young": 1.0, "area": 1.0, "density": 1.0}, "solver_type": "pinn-gd",
            "nn_config": {"young": {"enabled": True, "neurons_per_layer": 20, "input_dim": 3},
                          "area": {"enabled": True, "neurons_per_layer": 15, "input_dim": 3}},
            "pinn_config": {"max_iterations": 12, "learning_rate_u": 1e-2, "learning_rate_theta": 1e-3,
                            "alpha_data": 100.0},
            "solver_config": {"n_increments": 1}}
    f = tmp_path / "chain.json"
    f.write_text(json.dumps(data))
    torch.manual_seed(11)
    parsed = g.parse_problem(str(f))
    model = parsed["model"]
    theta = [p.detach().numpy().copy() for p in model.material.get_all_torch_params()]
    out = g.solve_problem(parsed)
    arrays = out.pop("_arrays")
    assert out["iterations"] == 12 and "displacements" not in out
    assert arrays["displacements"].shape == (400_002,) and arrays["young_at_elements_lf1.0"].shape == (200_000,)
    md = parsed["measured_data"]
    pb = orc.Problem(nodes=model.nodes, elements=model.elements, loads=model.loads, fixed_dofs=model.fixed_dofs,
                     dimension=2, young=orc.NetParams(theta[0:6]), area=orc.NetParams(theta[6:12]),
                     measured_vals=md["values"], measured_dofs=md["dofs"])
    ref = orc.solve_gd(pb, orc.SolverConfig(max_iterations=12, learning_rate_u=1e-2, learning_rate_theta=1e-3),
                       1.0, fe_mode="delta")
    assert rel_err(arrays["displacements"], ref.displacements.flatten()) < 2e-5
    assert rel_err([h["loss_total"] for h in out["history"]], [h["loss_total"] for h in ref.history]) < 1e-4
    # and through main(): files next to the input
    g.main(["generic.py", str(f)])
    res = json.loads((tmp_path / "chain.res.json").read_text())
    assert res["arrays_npz"] == "chain.res.npz" and (tmp_path / "chain.res.npz").exists()
    assert (tmp_path / "chain.log").exists()


def _random_truss(n_nodes, rng, hub_degree=0):
    """Random planar truss: nodes on a jittered grid, elements to nearest neighbours (+ an optional
    hub node connected to `hub_degree` nodes: skewed node degree)."""
    side = int(np.ceil(np.sqrt(n_nodes)))
    ij = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 2)[:n_nodes]
    nodes = ij + rng.uniform(-0.3, 0.3, ij.shape)
    el = set()
    idx = {tuple(p): k for k, p in enumerate(ij)}
    for k, (i, j) in enumerate(ij):
        for di, dj in ((1, 0), (0, 1), (1, 1), (1, -1)):
            q = idx.get((i + di, j + dj))
            if q is not None:
                el.add((k, q))
    el = sorted(el)
    if hub_degree:
        far = rng.choice(np.arange(1, n_nodes), size=hub_degree, replace=False)
        el += [(0, int(q)) for q in far if (0, int(q)) not in el]
    el = np.array(el)
    rng.shuffle(el)                                   # element order != node order
    flip = rng.random(len(el)) < 0.5                  # random element orientation
    el[flip] = el[flip][:, ::-1]
    return nodes, el


@pytest.mark.parametrize("wg", [3, 2])
@pytest.mark.parametrize("n_nodes,hub", [(400, 0), (2500, 300)])
def test_irregular_truss_vs_oracle(n_nodes, hub, wg):
    """General connectivity: shuffled element order, random orientation, node degree up to 300."""
    from pinn_fem_amd.fem.model import FEMModel, Material
    from pinn_fem_amd.fem.properties import NNProperty
    from pinn_fem_amd.nets import SimpleNN
    rng = np.random.default_rng(n_nodes)
    nodes, elements = _random_truss(n_nodes, rng, hub)
    ndof = 2 * n_nodes
    loads = rng.normal(size=ndof) * 0.05
    fixed = np.unique(rng.choice(ndof, size=ndof // 20, replace=False))
    u = (rng.normal(size=ndof) * 0.01).astype(np.float32)
    u[fixed] = 0
    md = rng.choice(ndof, size=ndof // 3, replace=False)
    mv = rng.normal(size=md.size) * 0.01
    torch.manual_seed(5)
    ne, na = SimpleNN(2, 20, 3), SimpleNN(2, 15, 3)
    th = [p.detach().numpy().copy() for p in list(ne.parameters()) + list(na.parameters())]
    model = FEMModel(nodes, elements, Material(NNProperty(ne, 3, True, 2.0), NNProperty(na, 3, True, 0.3)), loads, fixed)
    pb = orc.Problem(nodes=nodes, elements=elements, loads=loads, fixed_dofs=fixed, dimension=2,
                     young=orc.NetParams(th[0:6], 2.0), area=orc.NetParams(th[6:12], 0.3),
                     measured_vals=mv, measured_dofs=md)
    ref = orc.loss_and_grads(pb, orc.element_geometry(pb), u, 0.8, 1.0, 100.0)
    eng = _engine(model, mv, md, wg)
    losses, gu, gt = eng.loss_and_grads(torch.from_numpy(u), 0.8, 1.0, 100.0)
    f_int = eng.internal_force(torch.from_numpy(u), 0.8).cpu().numpy()
    assert rel_err(f_int, ref.f_int) < 5e-6
    assert abs(losses["loss_total"] - ref.loss_total) < 2e-5 * abs(ref.loss_total)
    assert rel_err(gu.cpu().numpy(), ref.grad_u) < 2e-5
    assert rel_err(gt.cpu().numpy(), np.concatenate([g.reshape(-1) for g in ref.grad_theta])) < 1e-4
    assert rel_err(eng.diag_k(0.8).cpu().numpy(), orc.diag_stiffness(pb, orc.element_geometry(pb), 0.8)) < 2e-6
    # 15 GD iterations from this state: same trajectory as the oracle
    from pinn_fem_amd.fem.solver import SolverConfig, solve_gd
    cfg = SolverConfig(max_iterations=15, learning_rate_u=1e-3, learning_rate_theta=1e-3, tolerance=1e-14)
    res = solve_gd(model, cfg, mv, md, target_load_factor=0.8, u_initial=torch.from_numpy(u))
    r2 = orc.solve_gd(pb, orc.SolverConfig(max_iterations=15, learning_rate_u=1e-3, learning_rate_theta=1e-3,
                                           tolerance=1e-14), 0.8, u_initial=u)
    assert rel_err(res.displacements.flatten(), r2.displacements.flatten()) < 2e-5
    assert rel_err([h["loss_total"] for h in res.history], [h["loss_total"] for h in r2.history]) < 1e-4


def test_error_behaviour_matches_reference_semantics():
    from pinn_fem_amd.fem.model import FEMModel, Material
    from pinn_fem_amd.fem.properties import NNProperty
    from pinn_fem_amd.fem.solver import SolverConfig, solve, solve_gd, solve_hybrid
    from pinn_fem_amd.nets import SimpleNN
    nodes = np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 0.0]])
    with pytest.raises(ValueError, match="zero initial length"):       # nn_assembly.py:66-67
        solve_gd(FEMModel(nodes, np.array([[1, 2]]), Material(1.0, 1.0), np.zeros(6), np.array([0])))
    with pytest.raises(ValueError, match="loads size"):                # model.py:78-79
        FEMModel(nodes[:2], np.array([[0, 1]]), Material(1.0, 1.0), np.zeros(3), np.array([0]))
    # NN with input_dim != dimension+1: the reference raises a torch shape error (properties.py:150)
    bad = FEMModel(nodes[:2], np.array([[0, 1]]), Material(NNProperty(SimpleNN(2, 8, 2), 2), 1.0), np.zeros(4),
                   np.array([0, 1, 3]))
    with pytest.raises(RuntimeError, match="cannot be multiplied"):
        solve_gd(bad, SolverConfig(max_iterations=2))
    ok = FEMModel(nodes[:2], np.array([[0, 1]]), Material(1.0, 1.0), np.array([0, 0, 1.0, 0]), np.array([0, 1, 3]))
    rh = solve_hybrid(ok, SolverConfig(max_iterations=20))            # scalar hybrid -> solve_nr (solver.py:653-692)
    assert rh.converged and abs(rh.displacements[1, 0] - 1.0) < 1e-12 and rh.history[-1]["iterations"] == 2.0
    ra = solve(ok, SolverConfig(max_iterations=20))                    # auto + no NN + no data -> nr (:1077-1079)
    assert ra.converged and abs(ra.displacements[1, 0] - 1.0) < 1e-12
    with pytest.raises(ValueError, match="Unknown solver method"):
        solve(ok, SolverConfig(method="bogus"))
    r = solve(ok, SolverConfig(method="gd", max_iterations=400, learning_rate_u=0.01, n_increments=2))
    assert r.converged and abs(r.displacements[1, 0] - 1.0) < 5e-3 and r.nn_parameters is None
    assert set(r.history[0]) == {"iteration", "loss_total", "loss_physics", "loss_data", "u_norm", "residual_norm"}


def test_cli_under_torchrun_two_ranks_one_gpu(tmp_path):
    """`torchrun --nproc-per-node 2 generic.py example3-P.json` (rehearsed on one GPU with gloo): the
    sharded run writes the same result schema; displacements match the single-process run."""
    import json
    import os
    import shutil
    import subprocess
    import sys
    from helpers import ROOT, input_json
    dst = tmp_path / "example3-P.json"
    shutil.copy(input_json("example3-P"), dst)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PINNFEM_QUIET="1", PINNFEM_DIST_BACKEND="gloo",
               PINNFEM_ONE_GPU="1")
    # same seed on both ranks and in the single-process run: identical initial theta
    runner = tmp_path / "run.py"
    runner.write_text("import sys, torch\nsys.path.insert(0, %r)\ntorch.manual_seed(0)\n"
                      "from pinn_fem_amd.cli.generic import main\nmain(['generic.py'] + sys.argv[1:])\n" % ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29671", str(runner), str(dst)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    sharded = json.loads((tmp_path / "example3-P.res.json").read_text())
    run = load_run("example3-P")
    assert sharded["converged"] == run["result"]["converged"]
    assert rel_err(sharded["displacements"], run["result"]["displacements"]) < 1e-5
    assert set(sharded) == set(run["result"])


def test_cli_under_torchrun_unseeded_ranks_share_rank0_theta(tmp_path):
    """under torchrun every rank builds its networks from its own RNG stream; the sharded
    solve must start every replica from rank 0's parameters.  Ranks are seeded DIFFERENTLY on purpose; all ranks
    must end with the same theta, and the result must be the single-process run started from rank 0's seed."""
    import json
    import os
    import shutil
    import subprocess
    import sys
    from helpers import ROOT, input_json
    dst = tmp_path / "example3-P.json"
    shutil.copy(input_json("example3-P"), dst)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PINNFEM_QUIET="1", PINNFEM_DIST_BACKEND="gloo",
               PINNFEM_ONE_GPU="1")
    runner = tmp_path / "run.py"
    runner.write_text(
        "import json, os, sys, torch\nsys.path.insert(0, %r)\n"
        "rank = int(os.environ['RANK'])\ntorch.manual_seed(1234 + 77 * rank)\n"
        "from pinn_fem_amd.cli import generic as g\n"
        "g._init_distributed()\n"
        "parsed = g.parse_problem(sys.argv[1])\n"
        "theta0 = [p.detach().reshape(-1).tolist() for p in parsed['model'].material.get_all_torch_params()]\n"
        "out = g.solve_problem(parsed)\n"
        "json.dump({'theta0': theta0, 'theta': out['nn_parameters'], 'u': out['displacements'],\n"
        "           'iterations': out['iterations']}, open(sys.argv[2] + '.rank%%d' %% rank, 'w'))\n"
        "g._shutdown_distributed(False)\n" % ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29672", str(runner), str(dst), str(tmp_path / "out")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    got = [json.loads((tmp_path / f"out.rank{k}").read_text()) for k in range(2)]
    # the ranks did start from different networks ...
    assert got[0]["theta0"] != got[1]["theta0"]
    # ... and finished with the same one, the same displacements and the same iteration count
    assert got[0]["theta"] == got[1]["theta"]
    assert got[0]["u"] == got[1]["u"] and got[0]["iterations"] == got[1]["iterations"]
    # single-process run from rank 0's seed
    from pinn_fem_amd.cli import generic as g
    torch.manual_seed(1234)
    single = g.solve_problem(g.parse_problem(str(dst)))
    assert rel_err(got[0]["u"], single["displacements"]) < 1e-5
    for k, v in single["nn_parameters"].items():
        # whole 10-increment run (thousands of Adam steps); the shards sum in a different order: 1e-4 on theta
        assert rel_err(np.array(got[0]["theta"][k]).reshape(-1), np.array(v).reshape(-1)) < 1e-4, k


def test_sharded_c_driver_real_rccl_world1():
    """The product's multi-GPU driver (pf_shard_iterations: kernels + ncclAllReduce from one C loop on an own
    RCCL communicator), its opt-in hipGraph form (pf_shard_iterations_graph: the collective captured inside) and the
    torch.distributed driver, all on a real RCCL process group of ONE rank (all this box offers): bit-identical to
    the single-engine path after 40 iterations on a 20000-element chain."""
    import subprocess
    import sys
    from helpers import ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", RANK="0", WORLD_SIZE="1",
               PINNFEM_QUIET="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_world1.py"), "20000", "30"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    got = json.loads(line)
    assert got["iters"] == [40, 40]
    assert got["rel_err_u"] == 0.0 and got["rel_err_theta"] == 0.0, got          # all three drivers, bit for bit
    # the opt-in form with whole iterations (collective included) replayed from a hipGraph really ran
    assert got["driver_used_c"] == "c-rccl" and got["driver_used_c+graph"] == "c-rccl+graph", got


# ---- classical Newton-Raphson on the device ------------------------------------
@pytest.mark.parametrize("name", ["nr_warren_scalar.npz", "nr_chain300_scalar.npz"])
def test_newton_raphson_meshes_vs_reference(name):
    """solve_nr (matrix-free float64 K v + Jacobi-PCG on the device) against the reference's dense float64
    Newton-Raphson on the fixture meshes.  Tolerance 1e-6 relative: the device geometry (c^2, cs, s^2, l0) is
    the float32 rounding of the reference's float64 values (the plan the GD kernels share)."""
    from pinn_fem_amd.fem.model import FEMModel, Material
    from pinn_fem_amd.fem.solver import SolverConfig, solve_nr
    rec = load_npz(name)
    model = FEMModel(nodes=rec["nodes"], elements=rec["elements"],
                     material=Material(float(rec["young"]), float(rec["area"]), 1.0), loads=rec["loads"],
                     fixed_dofs=rec["fixed"], dimension=2)
    res = solve_nr(model, SolverConfig(max_iterations=20, tolerance=float(rec["tolerance"])),
                   target_load_factor=float(rec["lam"]))
    assert res.converged == bool(rec["converged"])
    assert res.history[-1]["iterations"] == float(rec["iterations"])
    assert rel_err(res.displacements.reshape(-1), rec["u"]) < 1e-6
    assert np.max(np.abs(res.reactions.reshape(-1) - rec["reactions"])) < 1e-6 * max(1.0, np.max(np.abs(rec["reactions"])))


@pytest.mark.parametrize("ex", ["example1", "example1-1", "example5", "example5-P"])
def test_nr_and_scalar_hybrid_example_runs(ex):
    """`generic.py example1|1-1|5|5-P.json`: classical FEM (NR) and the scalar hybrid's GD -> NR switch,
    against the reference's results (integer-coordinate chains: exact geometry, so 1e-9)."""
    from pinn_fem_amd.cli import generic as g
    run = load_run(ex)
    out = g.solve_problem(product_example(ex))
    ref = run["result"]
    assert out["converged"] == ref["converged"]
    assert rel_err(out["displacements"], ref["displacements"]) < 1e-9
    assert np.max(np.abs(np.array(out["reactions"]) - np.array(ref["reactions"]))) < 1e-9
    last, rlast = out["history"][-1], ref["history"][-1]
    assert last["iterations"] == rlast["iterations"] and last["converged"] == rlast["converged"]
    assert abs(last["max_strain"] - rlast["max_strain"]) < 1e-9
    if "iteration" in rlast:
        assert abs(last["iteration"] - rlast["iteration"]) <= 3
    assert abs(out["iterations"] - ref["iterations"]) <= 3


def test_newton_raphson_errors_like_the_reference():
    """NN materials are refused with the reference's ValueError (solver.py:436-441); a mechanism (a free
    dof no element holds) raises the reference's RuntimeError instead of returning garbage."""
    from pinn_fem_amd.fem.model import FEMModel, Material
    from pinn_fem_amd.fem.properties import NNProperty
    from pinn_fem_amd.fem.solver import SolverConfig, solve_nr
    from pinn_fem_amd.nets import SimpleNN
    nodes = np.array([[0.0, 0.0], [1.0, 0.0], [2.0, 0.0]])
    el = np.array([[0, 1], [1, 2]])
    loads = np.array([0, 0, 0, 0, 1.0, 0.0])
    nn_model = FEMModel(nodes, el, Material(NNProperty(SimpleNN(2, 8, 3), input_dim=3, scale=1.0), 1.0, 1.0),
                        loads, np.array([0, 1, 3, 5]))
    with pytest.raises(ValueError, match="Newton-Raphson solver with NN materials"):
        solve_nr(nn_model, SolverConfig())
    mech = FEMModel(nodes, el, Material(1.0, 1.0, 1.0), np.array([0, 0, 0, 1.0, 0, 0]), np.array([0, 1]))
    with pytest.raises(RuntimeError, match="singular"):
        solve_nr(mech, SolverConfig(max_iterations=3))


@pytest.mark.parametrize("wg", [3, 2])
@pytest.mark.parametrize("panels", [1, 16, 33, 1000])
def test_warren_girder_vs_oracle(panels, wg, monkeypatch):
    """The synthetic 2-D truss of bench.py's second mesh (Warren girder, inclined members, node degree 4):
    one loss/gradient evaluation and 12 GD iterations against the oracle."""
    from pinn_fem_amd.fem.solver import SolverConfig, solve_gd
    from pinn_fem_amd.nets import SimpleNN
    from pinn_fem_amd.plan import warren_mesh
    nodes, elements, loads, fixed, mv, md = warren_mesh(panels)
    torch.manual_seed(3)
    widths = (20, 15, None)
    theta = []
    for w in widths:
        if w is not None:
            theta += [p.detach().numpy().copy() for p in SimpleNN(2, w, 3).parameters()]
    model = product_model(nodes, elements, loads, fixed, 2, widths, (2.0, 0.5, 1.0), theta)
    pb = orc.Problem(nodes=nodes, elements=elements, loads=loads, fixed_dofs=fixed, dimension=2,
                     measured_vals=mv, measured_dofs=md,
                     young=orc.NetParams([t.copy() for t in theta[0:6]], 2.0),
                     area=orc.NetParams([t.copy() for t in theta[6:12]], 0.5), density=1.0)
    rng = np.random.default_rng(5)
    u = (1e-3 * rng.standard_normal(2 * len(nodes))).astype(np.float32)
    u[fixed] = 0.0
    geo = orc.element_geometry(pb)
    ref = orc.loss_and_grads(pb, geo, u, 0.6, 1.0, 100.0)
    monkeypatch.setenv("PINNFEM_WG_MODE", str(wg))        # solve_gd below builds its own engine
    eng = _engine(model, mv, md, wg, 0)
    losses, gu, gt = eng.loss_and_grads(torch.from_numpy(u), 0.6, 1.0, 100.0)
    assert abs(losses["loss_total"] - ref.loss_total) < 2e-5 * abs(ref.loss_total)
    assert rel_err(gu.cpu().numpy(), ref.grad_u) < 2e-5
    ref_t = np.concatenate([g.reshape(-1) for g in ref.grad_theta if g is not None])
    assert rel_err(gt.cpu().numpy(), ref_t) < 5e-5
    # Adam divides by sqrt(v): on a dof whose physics and data gradients cancel to round-off size the step is
    # decided by the last bits (the first steps are lr*g/(|g|+eps), i.e. +-lr), so single dofs may legitimately
    # take another step direction.  Checked: the gradients above (2e-5), the loss trajectory (2e-5), the median dof within
    # 1e-5 of max|u|, and no dof further apart than the steps taken allow.
    for n_it in (3, 12):
        cfg = SolverConfig(max_iterations=n_it, learning_rate_u=1e-4, learning_rate_theta=5e-4, tolerance=0.0)
        got = solve_gd(model, cfg, mv, md, target_load_factor=0.6)
        oref = orc.solve_gd(pb, orc.SolverConfig(max_iterations=n_it, learning_rate_u=1e-4,
                                                 learning_rate_theta=5e-4, tolerance=0.0), 0.6)
        du = np.abs(np.asarray(got.displacements, dtype=np.float64) - oref.displacements).reshape(-1)
        scale = np.max(np.abs(oref.displacements))
        assert np.median(du) <= 1e-5 * scale
        assert du.max() <= 2.0 * 1e-4 * n_it
        assert rel_err([h["loss_total"] for h in got.history], [h["loss_total"] for h in oref.history]) < 2e-5
        # both paths train the modules / parameter lists in place: restart from the same theta
        model = product_model(nodes, elements, loads, fixed, 2, widths, (2.0, 0.5, 1.0), theta)
        pb.young = orc.NetParams([t.copy() for t in theta[0:6]], 2.0)
        pb.area = orc.NetParams([t.copy() for t in theta[6:12]], 0.5)


def test_newton_raphson_1d_bar_vs_oracle():
    """dimension = 1 (list-format nodes; fem/element.py:15-42): the float64 K v / Jacobi-PCG kernels' 1-D
    instantiation against the oracle's dense restatement of solve_nr on a non-uniform 40-element bar."""
    from pinn_fem_amd.fem.model import FEMModel, Material
    from pinn_fem_amd.fem.solver import SolverConfig, solve_nr
    rng = np.random.default_rng(11)
    x = np.concatenate([[0.0], np.cumsum(0.5 + rng.random(40))])
    el = np.stack([np.arange(40), np.arange(1, 41)], axis=1)
    loads = np.zeros(41)
    loads[40] = 2.0
    loads[17] = -0.7
    fixed = np.array([0])
    model = FEMModel(nodes=x, elements=el, material=Material(3.0, 0.25, 1.0), loads=loads, fixed_dofs=fixed,
                     dimension=1)
    res = solve_nr(model, SolverConfig(max_iterations=10, tolerance=1e-10), target_load_factor=0.8)
    pb = orc.Problem(nodes=x, elements=el, loads=loads, fixed_dofs=fixed, dimension=1, young=3.0, area=0.25,
                     density=1.0)
    ref = orc.solve_nr(pb, orc.SolverConfig(max_iterations=10, tolerance=1e-10), 0.8)
    assert res.converged and ref.converged and res.history[-1]["iterations"] == ref.history[-1]["iterations"]
    # float32 element lengths in the device plan: 1e-6
    assert rel_err(res.displacements.reshape(-1), ref.displacements.reshape(-1)) < 1e-6
    assert np.max(np.abs(res.reactions.reshape(-1) - ref.reactions.reshape(-1))) < 1e-6
    assert abs(res.history[-1]["max_strain"] - ref.history[-1]["max_strain"]) < 1e-6 * ref.history[-1]["max_strain"]


def test_cli_one_dimensional_list_format_json(tmp_path):
    """`generic.py` on a dimension-1 problem in the list format (nodes [[x], ...], elements [[i, j], ...], explicit
    fixed_dofs; examples/json/generic.py:155-192): classical FEM (auto -> Newton-Raphson) and a PINN-GD run with an
    E = NN(load_factor, x) net, against the oracle on the same inputs."""
    from pinn_fem_amd.cli import generic as g
    xs = [0.0, 1.0, 2.5, 3.0, 4.5]
    base = {"nodes": [[x] for x in xs], "elements": [[0, 1], [1, 2], [2, 3], [3, 4]], "fixed_dofs": [0],
            "loads": [0.0, 0.0, 0.5, 0.0, 1.0], "material": {"young": 2.0, "area": 0.5, "density": 1.0}}
    # (1) classical FEM
    p1 = tmp_path / "bar1d_fem.json"
    p1.write_text(json.dumps(dict(base, solver_type="fem", solver_config={"n_increments": 2, "tolerance": 1e-10})))
    out = g.solve_problem(g.parse_problem(str(p1)))
    pb = orc.Problem(nodes=np.array(xs), elements=np.array(base["elements"]), loads=np.array(base["loads"]),
                     fixed_dofs=np.array([0]), dimension=1, young=2.0, area=0.5, density=1.0)
    ref = orc.solve(pb, orc.SolverConfig(method="nr", n_increments=2, tolerance=1e-10))
    assert out["converged"] and rel_err(np.array(out["displacements"]).reshape(-1), ref.displacements.reshape(-1)) < 1e-6
    # (2) PINN-GD with a net on E: 40 iterations per increment from the same initial parameters
    p2 = tmp_path / "bar1d_gd.json"
    meas = {"global_dof": [1, 2, 3, 4], "measured_u": [0.4, 1.0, 1.2, 1.9]}
    p2.write_text(json.dumps(dict(base, solver_type="pinn-gd", measured_displacements=meas,
                                  nn_config={"young": {"enabled": True, "hidden_layers": 2, "neurons_per_layer": 12,
                                                       "input_dim": 2}},
                                  solver_config={"n_increments": 2, "max_iterations": 40, "tolerance": 1e-30,
                                                 "learning_rate_u": 0.01, "learning_rate_theta": 1e-3})))
    torch.manual_seed(5)
    parsed = g.parse_problem(str(p2))
    theta = [p.detach().cpu().numpy().copy() for p in parsed["model"].material.get_all_torch_params()]
    out2 = g.solve_problem(parsed)
    pb2 = orc.Problem(nodes=np.array(xs), elements=np.array(base["elements"]), loads=np.array(base["loads"]),
                      fixed_dofs=np.array([0]), dimension=1, young=orc.NetParams([t.copy() for t in theta], 2.0),
                      area=0.5, density=1.0, measured_vals=np.array(meas["measured_u"]), measured_dofs=np.array(meas["global_dof"]))
    ref2 = orc.solve(pb2, orc.SolverConfig(method="gd", n_increments=2, max_iterations=40, tolerance=1e-30,
                                           learning_rate_u=0.01, learning_rate_theta=1e-3))
    assert rel_err(np.array(out2["displacements"]).reshape(-1), ref2.displacements.reshape(-1)) < 1e-5


# ---- reduced-precision MLP variant (BASELINE.json configs[4]: fp32 vs bf16 study) -------------------
def test_bf16_mlp_single_step_error_bound():
    """mlp_dtype='bf16' (plain bf16 operands of the hidden-layer and gradient products, f32 accumulate, everything else
    float32): one loss+gradient evaluation against the float32 oracle.  bf16 carries 8 significand bits (2^-9 relative
    rounding): documented bound 2e-3 on the properties' effect (loss), 1e-2 on the parameter gradients."""
    rec = load_npz("step_chain300_ex4shape.npz")
    pb = mesh_problem(rec, (20, 15, 10), (1.0, 1.0, 1.0))
    ref = orc.loss_and_grads(pb, orc.element_geometry(pb), rec["u"], 0.7, 1.0, 100.0)
    model = product_model(rec["nodes"], rec["elements"], rec["loads"], rec["fixed"], 2, (20, 15, 10), (1.0, 1.0, 1.0),
                          theta_from(rec))
    from pinn_fem_amd.engine import HipEngine
    eng = HipEngine(model, rec["meas_vals"], rec["meas_dofs"], mlp_dtype="bf16")
    losses, gu, gt = eng.loss_and_grads(torch.from_numpy(rec["u"]), 0.7, 1.0, 100.0)
    assert abs(losses["loss_total"] - ref.loss_total) < 2e-3 * abs(ref.loss_total)
    ref_t = np.concatenate([g.reshape(-1) for g in ref.grad_theta if g is not None])
    err = rel_err(gt.cpu().numpy()[: ref_t.size], ref_t)
    assert 1e-6 < err < 1e-2, err          # really the reduced-precision path, and within its bound
    with pytest.raises(NotImplementedError):
        HipEngine(model, rec["meas_vals"], rec["meas_dofs"], wg_mode=2, mlp_dtype="bf16")


@pytest.mark.parametrize("ex", ["example7", "example7-P"])
def test_bf16_mlp_hybrid_example_tolerance(ex):
    """The hybrid examples (3 NNs) with bf16 MLP products still converge to the solver tolerance, in a comparable
    number of iterations, to displacements within 5e-4 and identified E*A within 5e-3 of the float32 run
    (measured: 2.6e-5 / 1.2e-4 and 7e-4 / 1.5e-3, profiles/r02_bf16_study.json)."""
    from pinn_fem_amd.cli.generic import extract_nn_properties
    from pinn_fem_amd.fem.solver import solve
    run = load_run(ex)
    theta0 = [np.array(t, dtype=np.float32) for t in run["theta0"]]
    out = {}
    for dt in ("f32", "bf16"):
        parsed = product_example(ex, theta0)
        model = parsed["model"]
        model._pf_mlp_dtype = dt
        md = parsed["measured_data"]
        res = solve(model, parsed["solver_config"], md.get("values"), md.get("dofs"))
        ident = extract_nn_properties(model)
        ea = (np.array(ident["young"]["load_factor_variations"]["load_factor_1.0"]["at_elements"]["values"]) *
              np.array(ident["area"]["load_factor_variations"]["load_factor_1.0"]["at_elements"]["values"]))
        out[dt] = (res, ea)
    r32, rbf = out["f32"][0], out["bf16"][0]
    assert r32.converged and rbf.converged
    assert rbf.history[-1]["residual_norm"] < 1e-3 or rbf.history[-1]["loss_total"] < parsed["solver_config"].tolerance
    assert rel_err(rbf.displacements, r32.displacements) < 5e-4
    assert rel_err(out["bf16"][1], out["f32"][1]) < 5e-3
    assert abs(len(rbf.history) - len(r32.history)) <= 0.5 * len(r32.history)


# ---- BASELINE-size checks of the product path ----------------------------------------------------


def _assert_history_equal(a, b):
    """Histories of the iteration graph and of eager launches: every column bit for bit, except the u-norm monitor
    (column 3) — the graph's displacement update runs inside the forward launch and adds the block partials of
    sum u_free^2 in another (fixed) grouping than the stand-alone kernel: same sum to float32 round-off."""
    cols = [c for c in range(a.shape[1]) if c != 3]
    assert np.array_equal(a[:, cols], b[:, cols])
    assert np.allclose(a[:, 3], b[:, 3], rtol=2e-6, atol=0.0)


def test_full_size_graph_equals_eager_bitwise():
    """10^6 elements, ex4 shape, default element-force formulation: 40 iterations replayed as the dependency-DAG
    hipGraph (the product path of bench.py) and the same 40 launched eagerly in stream order end in bit-identical
    u, theta and loss history."""
    from bench import build_model
    from pinn_fem_amd.engine import HipEngine
    from pinn_fem_amd.fem.solver import SolverConfig
    outs = []
    for use_graph in (True, False):
        model, mv, md, _ = build_model(1_000_000, "ex4")
        cfg = SolverConfig(max_iterations=45, tolerance=0.0, learning_rate_u=0.01, learning_rate_theta=5e-4)
        eng = HipEngine(model, mv, md)
        eng.begin(None, 0.1, cfg, want_history=True)
        eng.iterate(40, use_graph=use_graph)
        torch.cuda.synchronize()
        assert eng.state().iter == 40
        outs.append((eng.u.cpu().numpy().copy(), eng.theta.flat.cpu().numpy().copy(), eng.history(40).copy()))
        if use_graph:
            assert eng.graph_creates == 1
        del eng
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])
    _assert_history_equal(outs[0][2], outs[1][2])
    assert np.all(np.isfinite(outs[0][2])) and outs[0][2][-1, 0] < outs[0][2][0, 0]      # and the loss went down


@pytest.mark.parametrize("mesh,n,fe", [("chain", 3000, 0), ("chain", 60_000, 1), ("warren", 60_000, 0), ("warren", 250_000, 1),
                                       ("bar1d", 70_001, 0), ("bar1d", 70_001, 1), ("truss", 2500, 0)])
def test_graph_equals_eager_meshes(mesh, n, fe):
    """The iteration graph runs the displacement update of iteration t-1 as node tasks INSIDE the forward launch of t
    (pf_node.h: 2 x 64 nodes per task, the gather rewritten with all loads of a level in flight); the eager launches use
    k_node_gradu.  Same bits in u, theta, the Adam moments and every history column but the u-norm, on: a chain with 3
    forward blocks, the Warren girder (node degree 4: two gather rounds per node), a 1-D bar (one dof per node, 4-byte
    stiffness records, net input [load_factor, x]), a random truss with shuffled, randomly oriented elements and a hub node
    of degree 300 (150 gather rounds in one lane of a node task), element forces in the reference and in the difference
    form."""
    from bench import build_model
    from pinn_fem_amd.engine import HipEngine
    from pinn_fem_amd.fem.model import FEMModel, Material
    from pinn_fem_amd.fem.properties import NNProperty
    from pinn_fem_amd.fem.solver import SolverConfig
    from pinn_fem_amd.nets import SimpleNN
    outs = []
    for use_graph in (True, False):
        if mesh == "bar1d":
            rng = np.random.default_rng(11)
            torch.manual_seed(5)
            nodes = np.concatenate([[0.0], np.cumsum(rng.uniform(0.5, 1.5, n))]) * (3.0 / n)
            elements = np.stack([np.arange(n), np.arange(1, n + 1)], 1)
            loads = rng.normal(size=n + 1) * 0.1
            md = rng.choice(np.arange(1, n + 1), size=n // 7, replace=False)
            mv = rng.normal(size=len(md)) * 0.02
            model = FEMModel(nodes, elements, Material(NNProperty(SimpleNN(2, 20, 2), 2, True, 1.5),
                                                       NNProperty(SimpleNN(2, 15, 2), 2, True, 0.7)),
                             loads, np.array([0]), dimension=1)
        elif mesh == "truss":
            rng = np.random.default_rng(n)
            torch.manual_seed(5)
            nodes, elements = _random_truss(n, rng, 300)
            loads = rng.normal(size=2 * n) * 0.05
            fixed = np.unique(rng.choice(2 * n, size=n // 10, replace=False))
            md = rng.choice(2 * n, size=2 * n // 3, replace=False)
            mv = rng.normal(size=md.size) * 0.01
            model = FEMModel(nodes, elements, Material(NNProperty(SimpleNN(2, 20, 3), 3, True, 2.0),
                                                       NNProperty(SimpleNN(2, 15, 3), 3, True, 0.3)), loads, fixed)
        else:
            model, mv, md, _ = build_model(n, "ex4", mesh=mesh)
        cfg = SolverConfig(max_iterations=60, tolerance=0.0, learning_rate_u=0.01, learning_rate_theta=5e-4)
        eng = HipEngine(model, mv, md, fe_mode=fe)
        assert eng.fusion_info() & 16                           # the displacement update rides in the forward launch
        eng.begin(None, 0.1, cfg, want_history=True)
        n_it = 2 * eng.GRAPH_ITERS + 3                          # two whole replays and a part of one
        eng.iterate(n_it, use_graph=use_graph)
        torch.cuda.synchronize()
        assert eng.state().iter == n_it
        outs.append((eng.u.cpu().numpy().copy(), eng.theta.flat.cpu().numpy().copy(), eng.m_u.cpu().numpy().copy(),
                     eng.v_u.cpu().numpy().copy(), eng.m_t.cpu().numpy().copy(), eng.history(n_it).copy()))
        del eng
    for a, b in zip(outs[0][:-1], outs[1][:-1]):
        assert np.array_equal(a, b)
    _assert_history_equal(outs[0][-1], outs[1][-1])
    assert np.all(np.isfinite(outs[0][-1]))


@pytest.mark.parametrize("layers,we,wa,dim", [(1, 7, 4, 2), (3, 8, 12, 2), (2, 27, 30, 2), (3, 30, 5, 1), (1, 16, 16, 1)])
def test_graph_equals_eager_net_shapes(layers, we, wa, dim):
    """The update prologue of the graph's forward launches computes the padded-image index of every parameter by arithmetic
    (pf_pad_index_of) where the eager update reads pf_problem.pad_index: same bits in theta, the Adam moments and u for one,
    two and three hidden layers, ragged widths (register buckets 4 ... 15) and both input dimensions."""
    from pinn_fem_amd.engine import HipEngine
    from pinn_fem_amd.fem.model import FEMModel, Material
    from pinn_fem_amd.fem.properties import NNProperty
    from pinn_fem_amd.fem.solver import SolverConfig
    from pinn_fem_amd.nets import SimpleNN
    n = 5000
    outs = []
    for use_graph in (True, False):
        rng = np.random.default_rng(7)
        torch.manual_seed(9)
        if dim == 2:
            nodes = np.stack([np.arange(n + 1) * (3.0 / n), np.zeros(n + 1)], 1)
            fixed = np.array([0, 1])
        else:
            nodes = np.arange(n + 1) * (3.0 / n)
            fixed = np.array([0])
        elements = np.stack([np.arange(n), np.arange(1, n + 1)], 1)
        ndof = (n + 1) * dim
        loads = rng.normal(size=ndof) * 0.1
        md = rng.choice(np.arange(dim, ndof), size=n // 5, replace=False)
        mv = rng.normal(size=len(md)) * 0.02
        model = FEMModel(nodes, elements, Material(NNProperty(SimpleNN(layers, we, dim + 1), dim + 1, True, 1.5),
                                                   NNProperty(SimpleNN(layers, wa, dim + 1), dim + 1, True, 0.7)),
                         loads, fixed, dimension=dim)
        cfg = SolverConfig(max_iterations=40, tolerance=0.0, learning_rate_u=0.01, learning_rate_theta=5e-4)
        eng = HipEngine(model, mv, md)
        eng.begin(None, 0.2, cfg, want_history=True)
        n_it = 2 * eng.GRAPH_ITERS + 1
        eng.iterate(n_it, use_graph=use_graph)
        torch.cuda.synchronize()
        assert eng.state().iter == n_it
        outs.append((eng.u.cpu().numpy().copy(), eng.theta.flat.cpu().numpy().copy(), eng.m_t.cpu().numpy().copy(),
                     eng.v_t.cpu().numpy().copy(), eng.history(n_it).copy()))
        del eng
    for a, b in zip(outs[0][:-1], outs[1][:-1]):
        assert np.array_equal(a, b)
    _assert_history_equal(outs[0][-1], outs[1][-1])


@pytest.mark.parametrize("n,max_it,tol,n_call,expect", [(250_000, 200, 0.0, 3 * 20 + 7, 67), (250_000, 200, 1e30, 60, 12),
                                                       (250_000, 47, 0.0, 60, 47), (60_000, 200, 0.0, 4 * 10, 40)])
def test_chained_replays_equal_plain_replays(n, max_it, tol, n_call, expect):
    """iterate(defer_tail=True): a replay ends behind its last gradient-row reduction and the NEXT replay's first iteration
    carries that iteration's parameter update, displacement update and bookkeeping (pf_graph_create_ex); flush() runs the one
    pending tail.  Same bits as plain replays in u, theta, the Adam moments and the history: whole replays plus an eager
    remainder, a stop raised inside the second replay's head (tolerance met at iteration 12), max_iterations reached inside a
    replay, and the 10-iteration graphs of a smaller mesh."""
    from bench import build_model
    from pinn_fem_amd.engine import HipEngine
    from pinn_fem_amd.fem.solver import SolverConfig
    outs = []
    for defer in (True, False):
        model, mv, md, _ = build_model(n, "ex4")
        cfg = SolverConfig(max_iterations=max_it, tolerance=tol, learning_rate_u=0.01, learning_rate_theta=5e-4)
        eng = HipEngine(model, mv, md)
        eng.begin(None, 0.1, cfg, want_history=True)
        if defer:
            assert eng.prepare_graph(chained=True)          # this problem has the chained form
        eng.iterate(n_call, defer_tail=defer)
        if defer and n_call % eng.GRAPH_ITERS == 0 and tol == 0.0 and max_it > n_call:
            torch.cuda.synchronize()
            assert eng.state().iter == n_call - 1           # the last iteration's bookkeeping is still pending
        eng.flush()
        torch.cuda.synchronize()
        st = eng.state()
        assert st.iter == expect and st.theta_half == 0
        outs.append((eng.u.cpu().numpy().copy(), eng.theta.flat.cpu().numpy().copy(), eng.m_t.cpu().numpy().copy(),
                     eng.v_t.cpu().numpy().copy(), eng.m_u.cpu().numpy().copy(), eng.history(expect).copy(),
                     (st.done, st.converged)))
        del eng
    for a, b in zip(outs[0][:-2], outs[1][:-2]):
        assert np.array_equal(a, b)
    _assert_history_equal(outs[0][-2], outs[1][-2])         # (a plain replay's last displacement update is the stand-alone kernel)
    assert outs[0][-1] == outs[1][-1]


@pytest.mark.parametrize("max_it,tol,expect", [(40, 1e30, 12), (13, 0.0, 13), (15, 0.0, 15)])
def test_graph_stop_in_mid_replay_equals_eager(max_it, tol, expect):
    """The iteration graph ping-pongs the displacement vector and the parameter state between two halves; a stop raised
    INSIDE a replay (stop test at iteration 12; max_iterations 13 and 15: odd counts, so the live halves are the
    alternates) must still leave u, theta, the Adam moments and the history where and what the eager launches leave.
    250000 elements: the dependency-DAG form of the graph (>= 2*10^5 elements)."""
    from bench import build_model
    from pinn_fem_amd.engine import HipEngine
    from pinn_fem_amd.fem.solver import SolverConfig
    outs = []
    for use_graph in (True, False):
        model, mv, md, _ = build_model(250_000, "ex4")
        cfg = SolverConfig(max_iterations=max_it, tolerance=tol, learning_rate_u=0.01, learning_rate_theta=5e-4)
        eng = HipEngine(model, mv, md)
        assert eng.fusion_info() == 1 + 2 + 4 + 16              # every fused form is active on this problem
        eng.begin(None, 0.1, cfg, want_history=True)
        eng.iterate(2 * eng.GRAPH_ITERS, use_graph=use_graph)
        torch.cuda.synchronize()
        st = eng.state()
        assert st.iter == expect and st.done == 1 and st.theta_half == 0
        outs.append((eng.u.cpu().numpy().copy(), eng.theta.flat.cpu().numpy().copy(), eng.m_t.cpu().numpy().copy(),
                     eng.v_t.cpu().numpy().copy(), eng.m_u.cpu().numpy().copy(), eng.history(expect).copy()))
        # the nets' operand images belong to the final theta: a property evaluation from the stored images (no re-pack)
        # equals one after re-packing
        del eng
    for a, b in zip(outs[0][:-1], outs[1][:-1]):
        assert np.array_equal(a, b)
    _assert_history_equal(outs[0][-1], outs[1][-1])


def test_full_size_loss_and_grads_vs_oracle_prefix():
    """pf_loss_and_grads at 10^6 elements (ex4 shape): f_int, grad_u and the residual-driven element adjoint on the
    first 5000 elements equal the oracle's on that prefix (interior nodes; the prefix's last node sees one element
    less in the oracle), and the parameter gradient is finite and reproducible run to run (fixed summation order).
    The bench's mesh (h = 1) with element forces in the difference form on both sides: the reference's 4-term dot
    cancels against |u| up to 10^3 there, which would bury the comparison in round-off both sides share."""
    n, m = 1_000_000, 5000
    model, pb, mv, md = _chain_model(n, h=1.0)
    eng = _engine(model, mv, md, 3, fe=1)
    x = np.arange(n + 1, dtype=np.float64)
    u = np.zeros(2 * (n + 1), dtype=np.float32)
    u[0::2] = (1e-3 * x * (1.0 + 0.05 * np.sin(x / 50.0))).astype(np.float32)
    losses, gu, gt = eng.loss_and_grads(torch.from_numpy(u), 0.6, 1.0, 100.0)
    gu1, gt1 = gu.cpu().numpy().copy(), gt.cpu().numpy().copy()
    losses2, gu, gt = eng.loss_and_grads(torch.from_numpy(u), 0.6, 1.0, 100.0)
    assert np.array_equal(gt1, gt.cpu().numpy()) and np.array_equal(gu1, gu.cpu().numpy())
    assert losses == losses2 and np.all(np.isfinite(gt1)) and np.any(gt1 != 0)
    # oracle on the prefix: same theta, same u; measurements of the prefix nodes; global measurement count n_meas
    mask = np.asarray(md) < 2 * (m + 1)
    sub_md = np.asarray(md)[mask]
    sub = orc.Problem(nodes=pb.nodes[: m + 1], elements=pb.elements[:m], loads=pb.loads[: 2 * (m + 1)],
                      fixed_dofs=pb.fixed_dofs[pb.fixed_dofs < 2 * (m + 1)], dimension=2,
                      young=pb.young, area=pb.area, density=pb.density,
                      measured_vals=np.asarray(mv)[mask], measured_dofs=sub_md)
    ref = orc.loss_and_grads(sub, orc.element_geometry(sub), u[: 2 * (m + 1)], 0.6, 1.0, 100.0 * sub_md.size / len(md),
                             fe_mode="delta")
    # interior dofs of the prefix (the last two nodes feel the cut)
    k = 2 * (m - 1)
    scale = np.max(np.abs(ref.grad_u[:k]))
    assert np.max(np.abs(gu1[:k] - ref.grad_u[:k])) < 2e-5 * scale


def _theta_tensors_like(ref_list, flat):
    """split the engine's flat gradient into the oracle's tensor list (active tensors only)"""
    out, o = [], 0
    for g in ref_list:
        if g is None:
            continue
        out.append(flat[o:o + g.size].reshape(g.shape))
        o += g.size
    assert o == flat.size
    return out


def test_full_size_loss_and_grads_vs_oracle_all_elements():
    """the DEFAULT engine (MFMA32) and the DEFAULT element-force formulation at the bench's state —
    10^6 elements, example4 shape, five GD iterations from u = 0 at load factor 0.1 — against the oracle over ALL
    elements: loss terms, grad_u on every dof, and EVERY parameter-gradient tensor.  The oracle accumulates its sums
    over elements in float64 (acc64; per-element terms stay float32), so the tolerance is the device path's own:
    fixed-order float32 partial sums over 10^6 terms and 2^-22 products.  Bounds: loss 2e-6, grad_u 1e-5 of its
    maximum, each grad_theta tensor 2e-5 of that tensor's maximum (measured ~3e-6)."""
    from bench import build_model
    from pinn_fem_amd.engine import HipEngine
    from pinn_fem_amd.fem.solver import SolverConfig
    n = 1_000_000
    model, mv, md, _ = build_model(n, "ex4")
    eng = HipEngine(model, mv, md)
    assert eng.wg_mode == 3 and eng.fe_mode == 0
    cfg = SolverConfig(max_iterations=10, tolerance=0.0, learning_rate_u=0.01, learning_rate_theta=5e-4)
    eng.begin(None, 0.1, cfg, want_history=False)
    eng.iterate(5, use_graph=False)
    torch.cuda.synchronize()
    u = eng.u.cpu().numpy().copy()
    theta = [p.detach().cpu().numpy().copy() for p in model.material.get_all_torch_params()]
    losses, gu, gt = eng.loss_and_grads(torch.from_numpy(u), 0.1, 1.0, 100.0)
    gu, gt = gu.cpu().numpy().copy(), gt.cpu().numpy().copy()
    pb = orc.Problem(nodes=model.nodes, elements=model.elements, loads=model.loads, fixed_dofs=model.fixed_dofs,
                     dimension=2, young=orc.NetParams(theta[0:6]), area=orc.NetParams(theta[6:12]),
                     density=orc.NetParams(theta[12:18]), measured_vals=mv, measured_dofs=md)
    ref = orc.loss_and_grads(pb, orc.element_geometry(pb), u, 0.1, 1.0, 100.0, acc64=True)
    assert abs(losses["loss_total"] - ref.loss_total) < 2e-6 * abs(ref.loss_total)
    assert abs(losses["loss_physics"] - ref.loss_physics) < 2e-6 * abs(ref.loss_physics)
    assert abs(losses["loss_data"] - ref.loss_data) < 2e-6 * abs(ref.loss_data)
    assert np.max(np.abs(gu - ref.grad_u)) < 1e-5 * np.max(np.abs(ref.grad_u))
    active = [g for g in ref.grad_theta if g is not None]
    assert len(active) == 12
    for k, (got, want) in enumerate(zip(_theta_tensors_like(ref.grad_theta, gt), active)):
        assert np.max(np.abs(got - want)) < 2e-5 * max(np.max(np.abs(want)), 1e-30), k


@pytest.mark.parametrize("direction", ["rising", "falling"])
def test_backward_running_rescale_vs_oracle(direction):
    """the MFMA32 backward's running power-of-two rescale.  A wave walks its 64-element tasks in
    ascending order, scales each task's back-propagated gradients by S = 2^14 / (max |g_z| * weight bound), lets S only
    FALL over its loop and multiplies its accumulated gradient tiles by the (exact) ratio when it does.  Two blocks
    (n_part_blocks = 2 -> 16 waves, ~20 tasks per wave on 20000 elements) and a displacement field whose strain grows
    (falls) by 2^20 along the bar, so that |g_z| ~ strain^2 sweeps 2^40: "rising" rescales at nearly every task,
    "falling" keeps the first S while the later tasks' operands shrink towards the f16 underflow range.  Gradients
    against the oracle (float64 sums).  The per-element terms alternate in sign here (g_z follows the residual, a
    difference of neighbouring element forces), so the sums cancel by orders of magnitude: the error is measured
    against sum_e |term_e| per entry, the scale of a float32 accumulation — bound 5e-6 (measured <= 5e-7 on the
    MFMA32 engine, 3e-7 on the exact-f32 engine: profiles/r03_rescale_diag.txt)."""
    n = 20_000
    # coordinates in [0, 3]: with h = 1 the layer-1 units of the far elements saturate (|z| ~ 10^4), where the reference's
    # tanh backward 1 - y*y (y = tanh in float32, one ulp from 1) is quantised in steps of 1.2e-7 while the kernels'
    # r (1 - r) form keeps its relative precision — a 5e-4 difference on EVERY engine, the exact-f32 one included
    # (tools/rescale_diag.py), that has nothing to do with the rescale under test
    model, pb, mv, md = _chain_model(n, h=3.0 / n)
    from pinn_fem_amd.engine import HipEngine
    # element forces in the difference form on both sides: |u| reaches 10^7 at the stiff end of this bar
    eng = HipEngine(model, mv, md, n_part_blocks=2, fe_mode=1)
    assert eng.wg_mode == 3
    e = np.arange(n, dtype=np.float64)
    expo = -10.0 + 20.0 * e / (n - 1)
    if direction == "falling":
        expo = expo[::-1]
    strain = np.exp2(expo) * (1.0 + 0.3 * np.sin(0.37 * e))
    u = np.zeros(2 * (n + 1), dtype=np.float32)
    u[2::2] = np.cumsum(strain).astype(np.float32)
    losses, gu, gt = eng.loss_and_grads(torch.from_numpy(u), 0.6, 1.0, 0.0)
    gt = gt.cpu().numpy().copy()
    ref = orc.loss_and_grads(pb, orc.element_geometry(pb), u, 0.6, 1.0, 0.0, acc64=True, fe_mode="delta")
    assert np.all(np.isfinite(gt))
    active = [g for g in ref.grad_theta if g is not None]
    worst = 0.0
    for k, (got, want, asum) in enumerate(zip(_theta_tensors_like(ref.grad_theta, gt), active, ref.grad_theta_abs)):
        asum = asum.reshape(want.shape)
        ratio = float(np.max(np.abs(got - want) / np.maximum(asum, 1e-300)))
        worst = max(worst, ratio)
        assert ratio < 5e-6, (direction, k, ratio)
        assert np.max(np.abs(got - want)) < 5e-6 * np.max(asum), (direction, k)
    print(f"running rescale [{direction}]: worst |err| / sum|terms| = {worst:.2e}")
    assert abs(losses["loss_total"] - ref.loss_total) < 2e-5 * abs(ref.loss_total)


@pytest.mark.parametrize("lam", [1.0e3, 3.0e4])
def test_large_load_factor_gradients_vs_oracle(lam):
    """the load factor is an INPUT of the nets and so an operand of the layer-1 gradient tile, where
    it travels as split f16; it carries its own power-of-two scale (from |lam|, like the coordinates') — with the fixed
    factor 256 of round 2 it overflowed f16 from |lam| >= 256 and dW1[:,0] came out wrong without any sign.  A 300-element
    bar with coordinates in [0, 3] and load-factor weights scaled by 1/lam keeps every unit off saturation, so the
    comparison with the oracle (float64 sums) is tight: every column of W1 on its own scale (the load-factor column's
    gradient is lam times the bias gradient) within 2e-4 of the column maximum, on the MFMA32 and the exact-f32 engine."""
    n = 300
    model, pb, mv, md = _chain_model(n, h=3.0 / n)
    with torch.no_grad():
        for prop, net in ((model.material.young, pb.young), (model.material.area, pb.area)):
            w1 = next(prop.net.parameters())
            w1[:, 0] *= 1.0 / lam
            net.tensors[0] = w1.detach().cpu().numpy().copy()
    rng = np.random.default_rng(7)
    u = np.zeros(2 * (n + 1), dtype=np.float32)
    u[2::2] = np.cumsum(0.01 * (1.0 + 0.3 * rng.standard_normal(n))).astype(np.float32)
    ref = orc.loss_and_grads(pb, orc.element_geometry(pb), u, lam, 1.0, 100.0, acc64=True)
    active = [g for g in ref.grad_theta if g is not None]
    for wg in (3, 2):
        eng = _engine(model, mv, md, wg)
        losses, gu, gt = eng.loss_and_grads(torch.from_numpy(u), lam, 1.0, 100.0)
        gt = gt.cpu().numpy().copy()
        assert np.all(np.isfinite(gt))
        assert abs(losses["loss_total"] - ref.loss_total) < 2e-5 * abs(ref.loss_total)
        for k, (got, want) in enumerate(zip(_theta_tensors_like(ref.grad_theta, gt), active)):
            w2 = want.reshape(want.shape[0], -1) if want.ndim > 1 else want.reshape(-1, 1)
            g2 = got.reshape(w2.shape)
            for c in range(w2.shape[1]):
                scale = max(np.max(np.abs(w2[:, c])), 1e-30)
                assert np.max(np.abs(g2[:, c] - w2[:, c])) < 2e-4 * scale, (wg, k, c)


def test_beyond_infinity_cache_1e7_elements():
    """10^7 elements (working set ~1.2 GB, far beyond the 256 MiB Infinity Cache), example4 shape,
    default engine and formulation: (1) sum f_int = 0 (self-equilibrated element forces); (2) one replay of the iteration
    hipGraph (20 iterations) == the same iterations launched eagerly, bit for bit; (3) loss terms / grad_u on the first 5000 elements'
    interior dofs against the oracle on that prefix (delta formulation on both sides: |u| reaches 10^4 on this bar)."""
    from bench import build_model
    from pinn_fem_amd.engine import HipEngine
    from pinn_fem_amd.fem.solver import SolverConfig
    n, m = 10_000_000, 5000
    model, mv, md, _ = build_model(n, "ex4")
    theta0 = [p.detach().cpu().numpy().copy() for p in model.material.get_all_torch_params()]
    cfg = SolverConfig(max_iterations=24, tolerance=0.0, learning_rate_u=0.01, learning_rate_theta=5e-4)
    eng = HipEngine(model, mv, md)
    outs = []
    for use_graph in (True, False):
        with torch.no_grad():
            for p, a in zip(model.material.get_all_torch_params(), theta0):
                p.copy_(torch.from_numpy(a).to(p.device))
        eng.begin(None, 0.1, cfg, want_history=True)
        eng.iterate(eng.GRAPH_ITERS, use_graph=use_graph)       # one whole replay
        torch.cuda.synchronize()
        assert eng.state().iter == eng.GRAPH_ITERS
        outs.append((eng.u.cpu().numpy().copy(), eng.theta.flat.cpu().numpy().copy(), eng.history(eng.GRAPH_ITERS).copy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    _assert_history_equal(outs[0][2], outs[1][2])
    assert np.all(np.isfinite(outs[0][2]))
    # (1) equilibrium of the element forces at a smooth displacement field
    x = np.arange(n + 1, dtype=np.float64)
    u = np.zeros(2 * (n + 1), dtype=np.float32)
    u[0::2] = (1e-3 * x * (1.0 + 0.05 * np.sin(x / 50.0))).astype(np.float32)
    ut = torch.from_numpy(u)
    f = eng.internal_force(ut, 0.6).double()
    scale = float(eng.prop_e[:n].abs().max() * eng.prop_a[:n].abs().max() * np.abs(u).max())
    assert abs(float(f.sum())) < 1e-6 * scale * np.sqrt(n)
    del f
    # (3) prefix oracle, delta formulation
    eng2 = HipEngine(model, mv, md, fe_mode=1)
    losses, gu, gt = eng2.loss_and_grads(ut, 0.6, 1.0, 100.0)
    gu = gu[: 2 * (m + 1)].cpu().numpy().copy()
    assert np.all(np.isfinite(gt.cpu().numpy()))
    theta = [p.detach().cpu().numpy().copy() for p in model.material.get_all_torch_params()]
    mask = np.asarray(md) < 2 * (m + 1)
    sub_md = np.asarray(md)[mask]
    sub = orc.Problem(nodes=model.nodes[: m + 1], elements=model.elements[:m], loads=model.loads[: 2 * (m + 1)],
                      fixed_dofs=model.fixed_dofs[model.fixed_dofs < 2 * (m + 1)], dimension=2,
                      young=orc.NetParams(theta[0:6]), area=orc.NetParams(theta[6:12]), density=1.0,
                      measured_vals=np.asarray(mv)[mask], measured_dofs=sub_md)
    ref = orc.loss_and_grads(sub, orc.element_geometry(sub), u[: 2 * (m + 1)], 0.6, 1.0, 100.0 * sub_md.size / len(md),
                             fe_mode="delta")
    k = 2 * (m - 1)
    assert np.max(np.abs(gu[:k] - ref.grad_u[:k])) < 2e-5 * np.max(np.abs(ref.grad_u[:k]))


def test_config1_ex3_shape_1e5_vs_oracle():
    """BASELINE.json configs[1]: example3 shape (E = NN, A and rho scalar) on a 10^5-element bar: 12 GD iterations of the
    product path (plain chain inside the hipGraph below 2*10^5 elements) against the oracle, default formulation."""
    from pinn_fem_amd.fem.solver import SolverConfig, solve_gd
    n = 100_000
    model, pb, mv, md = _chain_model(n, widths=(20, None, None), h=1.0)
    cfg = SolverConfig(max_iterations=12, learning_rate_u=0.01, learning_rate_theta=1e-3, tolerance=0.0)
    res = solve_gd(model, cfg, mv, md, target_load_factor=0.1)
    ref = orc.solve_gd(pb, orc.SolverConfig(max_iterations=12, learning_rate_u=0.01, learning_rate_theta=1e-3,
                                            tolerance=0.0), 0.1)
    assert len(res.history) == len(ref.history) == 12
    got_l = np.array([h["loss_total"] for h in res.history])
    ref_l = np.array([h["loss_total"] for h in ref.history])
    assert np.max(np.abs(got_l - ref_l) / np.abs(ref_l)) < 2e-5
    assert rel_err(res.displacements, ref.displacements) < 2e-5
    th = np.concatenate([t.reshape(-1) for t in pb.theta_list()])
    got_th = np.concatenate([v.reshape(-1) for v in res.nn_parameters.values()])
    assert rel_err(got_th, th) < 2e-5


def test_api_pinn_gd_matches_oracle_restatement():
    """pinn_inverse_problem_gd on the device (K_1 u and its adjoint by the HIP kernels, no host synchronisation inside
    the loop) against the oracle's restatement of the same definition: 300 iterations, loss trajectory, identified E and
    A, displacements.  PARITY UNPINNED against the reference: its callee does not exist (ImportError at
    FEM/python/api_pinn_gradient_descent.py:19); what is checked is that the device path computes what it documents."""
    from pinn_fem_amd.fem.nn_solver_gd import pinn_inverse_problem_gd
    rng = np.random.default_rng(5)
    n = 40
    ang = np.cumsum(rng.uniform(-0.2, 0.2, n))
    nodes = np.concatenate([[[0.0, 0.0]], np.cumsum(np.stack([np.cos(ang), np.sin(ang)], 1), 0)])
    el = np.stack([np.arange(n), np.arange(1, n + 1)], 1)
    f = np.zeros(2 * (n + 1))
    f[-2:] = [800.0, 300.0]
    fixed = [0, 1]
    e0, a0 = 2.0e5, 4.0e-3
    md = np.arange(2, 2 * (n + 1))
    um = rng.normal(size=md.size) * 1e-2
    kw = dict(n_iterations=300, learning_rate=2e-3, alpha=1.0, beta=50.0)
    got = pinn_inverse_problem_gd(nodes, el, f, fixed, e0, a0, um, md, **kw)
    ref = orc.pinn_inverse_problem_gd(nodes, el, f, fixed, e0, a0, um, md, **kw)
    gl = np.array([h["loss_total"] for h in got["history"]])
    rl = np.array([h["loss_total"] for h in ref["history"]])
    assert len(gl) == len(rl) == 300
    assert np.max(np.abs(gl - rl) / np.abs(rl)) < 1e-4
    assert abs(got["young_final"] / ref["young_final"] - 1) < 1e-5 and abs(got["area_final"] / ref["area_final"] - 1) < 1e-5
    # Adam turns a round-off-sized gradient into a full-size step (m / sqrt(v) ~ +-1), so single dofs of two float32
    # evaluations drift apart by a few steps of lr * u_scale = 2e-5 over 300 iterations (measured: 4e-6 absolute on one
    # dof = 2.3e-4 of max |u|, every other dof to 1e-7); bound: two such steps relative to max |u| ~ 0.018
    assert rel_err(got["u_final"], ref["u_final"]) < 2e-3


def _bench_line(args, env_extra=None, timeout=900):
    import subprocess
    import sys
    from helpers import ROOT
    env = dict(os.environ, PINNFEM_QUIET="1", **(env_extra or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line (the driver's contract)
    return json.loads(lines[0])


def test_bench_contract_single_gpu():
    """`bench.py` at a small size: one JSON line with the driver's keys, the roofline and cpu_baseline objects, no
    graph capture inside the timed region (asserted inside bench.py), K timed steps."""
    d = _bench_line(["--gpus", "1", "--steps", "20", "--warmup", "5", "--elems", "50000", "--cpu-sample", "20000",
                     "--cpu-iters", "2", "--no-also"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["graph_captured_before_timing"] is True and "workload" in d["config"]
    assert abs(d["value"] - 50000 * 20 / (d["ms_per_step"] * 20e-3)) < 1e-6 * d["value"]
    rf, cb = d["roofline"], d["cpu_baseline"]
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and len(cb["rows"]) >= 7
    assert any(r["mode"].startswith("V (vectorised CPU PyTorch") and r["torch_num_threads"] == cb["cores"] for r in cb["rows"])
    assert d["repeat"]["regions"] == 5 and d["repeat"]["steps_each"] == 20


def test_bench_iters_to_tol_entry():
    """bench.py's metric (ii) helper (GD iterations to tolerance) on a 200-element bar: ten increments, per
    increment iteration counts within the cap, hits counted, both element-force formulations."""
    import bench
    out = bench.iters_to_tol(torch.device("cuda", 0), sizes=(200,), max_iterations=400)
    assert set(out) == {"N=200, fe_mode=reference", "N=200, fe_mode=delta"}
    for v in out.values():
        assert len(v["iterations_per_increment"]) == 10 and all(12 <= k <= 400 for k in v["iterations_per_increment"])
        assert v["max_iterations_hits"] == sum(1 for k in v["iterations_per_increment"] if k == 400) or v["max_iterations_hits"] <= 10
        assert np.isfinite(v["final_loss_total"])


def test_bench_two_rank_rehearsal():
    """`bench.py --gpus 2` without a launcher starts its own two ranks; here both on this one GPU over gloo
    (PINNFEM_BENCH_ONE_GPU=1: a rehearsal of the N>1 code path, not a multi-GPU number): the sharded driver runs, the
    line reports the whole job (2 x elems per step) and which shard driver ran."""
    d = _bench_line(["--gpus", "2", "--steps", "10", "--warmup", "2", "--elems", "30000", "--no-cpu-baseline", "--repeat", "2"],
                    {"PINNFEM_BENCH_ONE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "REHEARSAL" in d["data"]
    # the fixed-total configuration beside the weak-scaling headline (configs[3]; 1e5 elements in the rehearsal)
    (c3,) = d["also"].values()
    assert c3["elements_total"] == 100_000 and c3["elements_per_gpu"] == 50_000 and c3["value"] > 0
    assert d["repeat"]["regions"] == 2 and d["repeat"]["ms_per_step_min"] <= d["repeat"]["ms_per_step_median"]
    assert d["config"]["elements_total"] == 60000 and "shard_driver" in d["config"]
    assert abs(d["value"] - 60000 * 10 / (d["ms_per_step"] * 10e-3)) < 1e-6 * d["value"]
