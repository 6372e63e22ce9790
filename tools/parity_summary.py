#!/usr/bin/env python3
"""Parity summary against the reference goldens (tests/golden/run_*.json, seed 0): for every example
run the HIP path (and the numpy oracle) end to end and report iteration counts, max relative errors of
displacements, reactions and of the identified E, A and E*A at the element centroids.
    python tools/parity_summary.py > profiles/r0N_parity_summary.json   (needs a GPU)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("PINNFEM_QUIET", "1")

_OUT = os.fdopen(os.dup(1), "w")      # the JSON goes to the real stdout; the solvers' progress prints go to stderr
os.dup2(2, 1)

from helpers import example_problem, load_run, orc, product_example, rel_err  # noqa: E402


def props_at_centroids(res_props, lf):
    out = {}
    for name in ("young", "area"):
        p = res_props[name]
        out[name] = (np.array(p["load_factor_variations"][lf]["at_elements"]["values"])
                     if p["type"] != "scalar" else np.full(3, p["value"]))
    return out


def oracle_props(pb, lam):
    geo = orc.element_geometry(pb)
    s, e, a, *_ = orc.element_stiffness(pb, geo, lam)
    return {"young": e, "area": a}


def main():
    import pinn_fem_amd.fem.solver as S
    from pinn_fem_amd.cli import generic as g
    rows = []
    for ex in ["example2", "example2-P", "example3", "example3-P", "example4", "example4-P", "example6",
               "example6-P", "example7", "example7-P"]:
        run = load_run(ex)
        ref = run["result"]
        theta0 = [np.array(t, dtype=np.float32) for t in run["theta0"]]
        ref_counts = [c["n_history"] for c in run["calls"]
                      if not (c["preconditioning"] and not c["skip_preconditioning"])]
        # HIP
        parsed = product_example(ex, theta0 if theta0 else None)
        counts = []
        orig = S.solve_gd

        def wrapper(model, config=None, measured_disp=None, measured_dofs=None, target_load_factor=1.0,
                    u_initial=None, skip_preconditioning=False):
            r = orig(model, config, measured_disp, measured_dofs, target_load_factor, u_initial,
                     skip_preconditioning)
            if not (config.preconditioning and not skip_preconditioning):
                counts.append(len(r.history))
            return r

        S.solve_gd = wrapper
        try:
            out = g.solve_problem(parsed)
        finally:
            S.solve_gd = orig
        # oracle
        pb, cfg = example_problem(ex, theta0 if theta0 else None)
        olog = []
        ores = orc.solve(pb, cfg, call_log=olog)
        row = {"example": ex, "reference_iterations_per_call": ref_counts, "hip_iterations_per_call": counts,
               "oracle_iterations_per_call": [c["n_history"] for c in olog],
               "hip_rel_err_displacements": rel_err(out["displacements"], ref["displacements"]),
               "oracle_rel_err_displacements": rel_err(ores.displacements.flatten(), ref["displacements"]),
               "hip_abs_err_reactions": float(np.max(np.abs(np.array(out["reactions"]) - np.array(ref["reactions"]))))}
        if "identified_properties" in ref:
            errs = {"hip": {}, "oracle": {}}
            for lf, lam in (("load_factor_0.2", 0.2), ("load_factor_0.5", 0.5), ("load_factor_1.0", 1.0)):
                pr = props_at_centroids(ref["identified_properties"], lf)
                ph = props_at_centroids(out["identified_properties"], lf)
                po = oracle_props(pb, lam)
                for who, pp in (("hip", ph), ("oracle", po)):
                    for name in ("young", "area"):
                        errs[who].setdefault(name, []).append(rel_err(pp[name], pr[name]))
                    errs[who].setdefault("young*area", []).append(
                        rel_err(pp["young"] * pp["area"], pr["young"] * pr["area"]))
            row["identified_at_centroids_max_rel_err"] = {
                who: {k: float(max(v)) for k, v in d.items()} for who, d in errs.items()}
        rows.append(row)
    _OUT.write(json.dumps(rows, indent=1) + "\n")


if __name__ == "__main__":
    main()
