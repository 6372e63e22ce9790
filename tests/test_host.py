"""CPU-side tests of the host logic: C-ABI library loads and exports every declared symbol, struct
layouts, padded-parameter indexing, mesh plan, JSON parsing precedence, loud failure without a GPU."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from helpers import ROOT, input_json


def test_library_exports_every_declared_symbol():
    from pinn_fem_amd import _capi
    lib = _capi.load()
    header = open(os.path.join(ROOT, "include", "pinnfem_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|long long|const char\*)\s+(pf_\w+)\s*\(", header, re.M))
    assert declared, "no declarations parsed"
    assert declared == set(_capi.SYMBOLS.keys())
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.pf_abi_version() == _capi.PF_ABI_VERSION
    for idx, st in enumerate((_capi.PfMesh, _capi.PfNet, _capi.PfState, _capi.PfProblem)):
        assert lib.pf_sizeof(idx) == C.sizeof(st)


def test_param_count_and_pad_index_are_a_bijection():
    from pinn_fem_amd import _capi
    lib = _capi.load()
    assert lib.pf_net_param_count(3, 20, 2) == 521     # example3 young net
    assert lib.pf_net_param_count(3, 15, 2) == 316
    assert lib.pf_net_param_count(3, 10, 2) == 161
    for in_dim, w, L in [(3, 20, 2), (3, 15, 2), (2, 7, 1), (3, 32, 3), (2, 1, 1)]:
        n = lib.pf_net_param_count(in_dim, w, L)
        pad = lib.pf_net_pad_count(in_dim, w, L)
        idx = [lib.pf_net_pad_index(in_dim, w, L, q) for q in range(n)]
        assert len(set(idx)) == n and min(idx) >= 0 and max(idx) < pad
        assert lib.pf_net_pad_index(in_dim, w, L, n) < 0
    assert lib.pf_padded_width(33) < 0 and lib.pf_padded_width(0) < 0
    assert lib.pf_net_pad_count(3, 40, 2) < 0 and lib.pf_net_pad_count(3, 8, 4) < 0


def test_argument_errors_do_not_touch_the_gpu():
    from pinn_fem_amd import _capi
    lib = _capi.load()
    p = _capi.PfProblem()
    assert lib.pf_gd_iterations(C.byref(p), 1, None) == _capi.PF_ERR_ARG
    assert b"mesh.dim" in lib.pf_last_error()
    with pytest.raises(ValueError):
        _capi.check(lib.pf_net_forward(C.byref(p), 0, None), "pf_net_forward")


def test_plan_chain_and_csr_order():
    from pinn_fem_amd.plan import build_host_plan, chain_mesh
    nodes, elements, loads, fixed, mv, md = chain_mesh(5, 0.5)
    hp = build_host_plan(nodes, elements, loads, fixed, 2, mv, md)
    assert hp.n_dofs == 12 and hp.n_meas == 10
    assert np.array_equal(hp.adj_ptr, [0, 1, 3, 5, 7, 9, 10])
    # node 1: element 0 end 1 (code 1), element 1 end 0 (code 2) — ascending element id
    assert list(hp.adj[1:3]) == [1, 2]
    assert np.allclose(hp.egeo[:, 0], 1) and np.allclose(hp.egeo[:, 3], 0.5)
    assert np.allclose(hp.ecent[:, 0], [0.25, 0.75, 1.25, 1.75, 2.25])
    assert hp.dof_flags[0] == 1 and hp.dof_flags[1] == 1 and hp.dof_flags[2] == 2 and hp.dof_flags[3] == 3
    assert np.array_equal(hp.free_dofs, [2, 4, 6, 8, 10])


def test_plan_irregular_mesh_matches_oracle_geometry():
    from helpers import load_npz, orc
    from pinn_fem_amd.plan import build_host_plan
    rec = load_npz("step_warren_EA.npz")
    hp = build_host_plan(rec["nodes"], rec["elements"], rec["loads"], rec["fixed"], 2,
                         rec["meas_vals"], rec["meas_dofs"])
    pb = orc.Problem(nodes=rec["nodes"], elements=rec["elements"], loads=rec["loads"],
                     fixed_dofs=rec["fixed"], dimension=2)
    geo = orc.element_geometry(pb)
    assert np.array_equal(hp.egeo[:, 0], geo.pattern[:, 0, 0])
    assert np.array_equal(hp.egeo[:, 1], geo.pattern[:, 0, 1])
    assert np.array_equal(hp.egeo[:, 2], geo.pattern[:, 1, 1])
    assert np.array_equal(hp.egeo[:, 3], geo.l0)
    assert np.array_equal(hp.ecent, geo.nn_input)
    # CSR: every (elem,end) appears exactly once, grouped by node, ascending inside a node
    for n in range(hp.n_nodes):
        codes = hp.adj[hp.adj_ptr[n]:hp.adj_ptr[n + 1]]
        assert list(codes) == sorted(codes)
        for c in codes:
            assert rec["elements"][c >> 1, c & 1] == n
    assert sorted(hp.adj) == list(range(2 * hp.n_elems))


def test_plan_rejects_bad_input():
    from pinn_fem_amd.plan import build_host_plan
    nodes = np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 0.0]])
    with pytest.raises(ValueError, match="zero initial length"):
        build_host_plan(nodes, [[1, 2]], np.zeros(6), [0], 2)
    with pytest.raises(ValueError, match="loads size"):
        build_host_plan(nodes, [[0, 1]], np.zeros(5), [0], 2)
    with pytest.raises(ValueError, match="out-of-range"):
        build_host_plan(nodes, [[0, 1]], np.zeros(6), [9], 2)
    with pytest.raises(NotImplementedError):
        build_host_plan(nodes, [[0, 1]], np.zeros(6), [0], 2, [1.0, 2.0], [2, 2])
    hp = build_host_plan(np.array([0.0, 1.0, 3.0]), [[0, 1], [1, 2]], np.zeros(3), [0], 1)
    assert hp.dim == 1 and np.allclose(hp.egeo[:, 3], [1, 2]) and hp.ecent.shape == (2, 1)


def test_parse_problem_mirrors_reference_semantics(tmp_path):
    from pinn_fem_amd.cli import generic as g
    parsed = g.parse_problem(input_json("example4-P"))
    cfg, model, md = parsed["solver_config"], parsed["model"], parsed["measured_data"]
    assert cfg.method == "gd" and cfg.preconditioning and cfg.max_iterations == 5000
    assert cfg.learning_rate_u == 0.01 and cfg.learning_rate_theta == 0.0005 and cfg.n_increments == 10
    assert list(md["dofs"]) == [2, 3, 4, 5, 6, 7] and list(md["values"]) == [1, 0, 2, 0, 3, 0]
    assert list(model.fixed_dofs) == [0, 1, 3, 5, 7] and model.dimension == 2
    assert sum(p.numel() for p in model.material.get_all_torch_params()) == 998
    assert g.parse_problem(input_json("example7"))["solver_config"].method == "hybrid"
    # example2: pinn-gd without measurements -> empty arrays, not None (generic.py:345-362)
    p2 = g.parse_problem(input_json("example2"))
    assert p2["measured_data"]["dofs"].size == 0 and not p2["model"].material.has_trainable_params()
    # precedence: solver_config.method over solver_type; solver_config lr over pinn_config lr;
    # pinn_config max_iterations over solver_config; list-format 1-D nodes; explicit fixed_dofs
    data = {"nodes": [[0.0], [1.0], [2.0]], "elements": [{"nodes": [0, 1]}, {"nodes": [1, 2]}],
            "fixed_dofs": [0], "loads": [0, 0, 1.0], "material": {"young": 2.0, "area": 3.0},
            "solver_type": "pinn-gd",
            "measured_displacements": {"global_dof": [2], "measured_u": [0.5]},
            "solver_config": {"method": "hybrid", "learning_rate_u": 0.5, "max_iterations": 7, "n_increments": 3},
            "pinn_config": {"learning_rate_u": 0.1, "max_iterations": 9, "neuronsPerLayer": 4},
            "nn_config": {"young": {"enabled": True, "hiddenLayers": 1, "neuronsPerLayer": 6, "input_dim": 2}}}
    f = tmp_path / "p.json"
    f.write_text(json.dumps(data))
    p = g.parse_problem(str(f))
    assert p["solver_config"].method == "hybrid" and p["solver_config"].learning_rate_u == 0.5
    assert p["solver_config"].max_iterations == 9 and p["solver_config"].n_increments == 3
    assert p["model"].dimension == 1 and p["model"].nodes.shape == (3,)
    assert list(p["measured_data"]["dofs"]) == [2]
    net = p["model"].material.young.net
    assert sum(q.numel() for q in net.parameters()) == 2 * 6 + 6 + 6 + 1
    assert float(net.net[-1].bias) == 1.0 and float(net.net[-1].weight[0, 0]) == pytest.approx(0.1)


def test_solver_config_defaults_match_reference():
    from pinn_fem_amd.fem.solver import SolverConfig
    c = SolverConfig()
    assert (c.max_iterations, c.tolerance, c.print_every, c.n_increments) == (1000, 1e-6, 10, 10)
    assert (c.learning_rate_u, c.learning_rate_theta, c.alpha_physics, c.alpha_data) == (1e-7, 1e-4, 1.0, 100.0)
    assert c.method == "auto" and c.preconditioning is False and c.min_denominator == 1e-10


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pinn_fem_amd._capi import PinnFemHipError
    from pinn_fem_amd.cli import generic as g
    parsed = g.parse_problem(input_json("example2"))
    with pytest.raises(PinnFemHipError, match="no CPU fallback"):
        g.solve_problem(parsed)


def test_describe_module_rejects_foreign_architectures():
    import torch.nn as nn
    from pinn_fem_amd.nets import SimpleNN, describe_module
    spec = describe_module(SimpleNN(2, 15, 3))
    assert (spec.in_dim, spec.width, spec.n_hidden, spec.n_params) == (3, 15, 2, 316)
    with pytest.raises(NotImplementedError):
        describe_module(nn.Sequential(nn.Linear(3, 8), nn.ReLU(), nn.Linear(8, 1)))
    with pytest.raises(NotImplementedError):
        describe_module(nn.Sequential(nn.Linear(3, 8), nn.Tanh(), nn.Linear(8, 4), nn.Tanh(), nn.Linear(4, 1)))
    with pytest.raises(NotImplementedError):
        describe_module(SimpleNN(2, 40, 3))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pinn_fem_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dirpath, fn)).read()
                # no import/exec of anything under oracle/ (the word may not even appear)
                assert "oracle" not in src, os.path.join(dirpath, fn)


def test_api_pinn_gd_surface(tmp_path):
    """api_pinn_gradient_descent.py surface: parse_input semantics (incl. the reference's elif chain)
    and the error JSON + exit code 1 contract (:206-219)."""
    import subprocess
    import sys
    from pinn_fem_amd.cli import api_pinn_gradient_descent as api
    data = {"nodes": [{"x": 0, "y": 0, "fixed": True}, {"x": 1, "y": 0, "fixed_x": True, "fixed_y": True},
                      {"x": 2, "y": 0, "fixed_y": True}],
            "elements": [{"nodes": [0, 1]}, {"nodes": [1, 2]}], "material": {"young": 2.0, "area": 0.5},
            "loads": [0, 0, 0, 0, 1.0, 0], "measured_disp": [1.0, 2.0], "measured_dofs": [2, 4],
            "solver_config": {"max_iterations": 7, "beta": 10.0}}
    p = api.parse_input(data)
    assert p["fixed_dofs"] == [0, 1, 2, 5]            # node 1: only x (elif chain), node 2: y
    assert p["n_iterations"] == 7 and p["beta"] == 10.0 and p["alpha"] == 1.0 and p["learning_rate"] == 0.001
    assert p["young_bounds"] == [1e9, 500e9] and p["n_dofs"] == 6
    bad = dict(data)
    bad.pop("measured_disp")
    fin, fout = tmp_path / "in.json", tmp_path / "out.json"
    fin.write_text(json.dumps(bad))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "api_pinn_gradient_descent.py"), str(fin), str(fout)],
                       capture_output=True, text=True)
    assert r.returncode == 1
    err = json.loads(fout.read_text())
    assert err["type"] == "ValueError" and "measured_disp" in err["error"]


def test_synthetic_chain_json(tmp_path):
    """build-only "accel" namespace: mesh generator for large runs; reference JSONs are unaffected."""
    from pinn_fem_amd.cli import generic as g
    data = {"accel": {"synthetic_chain": {"n_elements": 1000, "h": 0.5, "tip_load": 2.0}, "fe_mode": "delta"},
            "material": {"young": 1.0, "area": 1.0, "density": 1.0}, "solver_type": "pinn-gd",
            "nn_config": {"young": {"enabled": True, "neurons_per_layer": 20, "input_dim": 3}},
            "pinn_config": {"max_iterations": 30, "learning_rate_u": 0.01, "learning_rate_theta": 1e-3},
            "solver_config": {"n_increments": 2}}
    f = tmp_path / "chain.json"
    f.write_text(json.dumps(data))
    p = g.parse_problem(str(f))
    m = p["model"]
    assert m.nelm == 1000 and m.nnode == 1001 and m.loads[2000] == 2.0 and m.nodes[-1, 0] == 500.0
    assert len(m.fixed_dofs) == 1002 and p["measured_data"]["dofs"].size == 2000
    assert p["solver_config"].n_increments == 2 and p["solver_config"].max_iterations == 30
    assert getattr(m, "_pf_fe_mode", 0) == 1 and p["accel"]["fe_mode"] == "delta"
    assert "accel" in g.parse_problem(input_json("example3"))
