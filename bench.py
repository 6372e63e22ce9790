#!/usr/bin/env python3
"""bench.py — element-residual-grad evaluations per second of the PINN+GD hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--elems E] [--workload ex4|ex3]

One "step" is one complete GD iteration (FEM/python/fem/solver.py:254-355) over a synthetic
collinear 2-D truss (SURVEY.md §8(d) inputs): MLP forward for E and A at every element, element
stiffness, assembly of f_int, residual + data loss, full backward to u and theta, Adam on both,
BC clamp, monitors and stop test.  Default workload: BASELINE.json configs[2] — example4 shape
(3 NNs configured, E and A evaluated, like the reference) at 10^6 elements on one GPU; with
--gpus N every rank owns its own 10^6-element shard of one N*10^6-element bar (weak scaling).
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("PINNFEM_QUIET", "1")     # the solver's per-iteration table would break the ONE-JSON-line contract

ALGO_BYTES_PER_EVAL = 144.0      # algorithmic HBM bytes per element-eval (fp32)
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
F32_PEAK_TFLOPS = 157.3          # dense f32 MFMA peak = f32 vector peak (MI355X_MICROARCH.md)


def net_flops(width: int, in_dim: int = 3) -> float:
    """Flops of one net_backward launch per element (forward recompute + backward), SURVEY §8(d):
    forward MACs = in*h + h*h + h, backward ~= 1.9x forward."""
    macs = in_dim * width + width * width + width
    return 2.0 * macs * (1.0 + 1.9)


def build_model(n_elems: int, workload: str, seed: int = 0, x0: float = 0.0, mesh: str = "chain"):
    import torch
    from pinn_fem_amd.fem.model import FEMModel, Material
    from pinn_fem_amd.fem.properties import NNProperty
    from pinn_fem_amd.nets import SimpleNN
    from pinn_fem_amd.plan import chain_mesh, warren_mesh
    if mesh == "warren":      # 4n-1 elements for n panels
        nodes, elements, loads, fixed, mv, md = warren_mesh((n_elems + 1) // 4, 1.0, 1.0)
    else:
        nodes, elements, loads, fixed, mv, md = chain_mesh(n_elems, 1.0)
        nodes[:, 0] += x0
        mv = mv.copy()
        mv[0::2] += x0
    torch.manual_seed(seed)  # young -> area -> density, like parse_problem
    widths = {"ex4": (20, 15, 10), "ex3": (20, None, None)}[workload]
    props = []
    for w in widths:
        props.append(1.0 if w is None else NNProperty(SimpleNN(2, w, 3), input_dim=3,
                                                      enforce_positive=True, scale=1.0))
    model = FEMModel(nodes=nodes, elements=elements, material=Material(*props), loads=loads,
                     fixed_dofs=fixed, dimension=2)
    return model, mv, md, widths


def _cpu_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return model, os.cpu_count() or 1


def _oracle_problem(orc, n, workload):
    import torch
    from pinn_fem_amd.nets import SimpleNN
    from pinn_fem_amd.plan import chain_mesh
    nodes, elements, loads, fixed, mv, md = chain_mesh(n, 1.0)
    torch.manual_seed(0)
    widths = {"ex4": (20, 15, 10), "ex3": (20, None, None)}[workload]
    props = []
    for w in widths:
        if w is None:
            props.append(1.0)
        else:
            props.append(orc.NetParams([p.detach().numpy().copy() for p in SimpleNN(2, w, 3).parameters()]))
    return orc.Problem(nodes=nodes, elements=elements, loads=loads, fixed_dofs=fixed, dimension=2,
                       young=props[0], area=props[1], density=props[2], measured_vals=mv, measured_dofs=md)


def _usable_cpus() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(workload: str, n_sample: int, iters: int, threads: int):
    """The reference algorithm's CPU ports (kind "port"; the reference's files never travel to the GPU box) timed on the
    host cores on bounded samples of the same workload:
      (R)  reference-ORDER mode of the numpy oracle, one element at a time like assemble_system_torch's Python loop,
           N = 3 / 300 / 1000 (beside BASELINE.md's import-measured reference: 676 / 474 / 430 element-evals/s on 8
           cores); 1 thread;
      (V)  vectorised CPU PyTorch restatement WITH autograd and torch.optim.Adam (oracle/torch_vectorised.py: batched MLP,
           index_add_ assembly, loss.backward()), torch intra-op threads = `threads`: the reference's own cost class at
           scale, and the headline `value` of this object;
      (Vn) the numpy oracle's vectorised mode (hand-written backward, no autograd): BLAS threads = `threads`, its
           element-wise work (tanh, softplus, gathers) is single-threaded numpy.
    Each row states the threads it really used (`cores`)."""
    from oracle import pinn_oracle as orc
    import torch
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads)
    except Exception:
        limiter = None
    torch_threads_before = torch.get_num_threads()
    torch.set_num_threads(int(threads))
    lr_t = 5e-4 if workload == "ex4" else 1e-3
    model_name, ncpu = _cpu_info()
    rows = []
    ref_rows = {3: 676.0, 300: 474.0, 1000: 430.0}            # BASELINE.md section 2 (reference imported, 8 cores)
    for n, it in ((3, 400), (300, 10), (1000, 4)):
        pb = _oracle_problem(orc, n, workload)
        cfg = orc.SolverConfig(learning_rate_u=0.01, learning_rate_theta=lr_t, tolerance=0.0)
        orc.gd_iterations_loop(pb, cfg, 0.1, 1)
        t0 = time.perf_counter()
        orc.gd_iterations_loop(pb, cfg, 0.1, it)
        dt = time.perf_counter() - t0
        rows.append({"mode": "R (reference order, per-element loop, numpy oracle)", "n_elems": n, "iterations": it,
                     "value": n * it / dt, "unit": "element-evals/s", "wall_s": dt, "cores": 1,
                     "reference_measured_BASELINE_md": ref_rows[n]})
    from oracle.torch_vectorised import TorchVectorisedGD
    head = None
    for n, it in ((100_000, max(iters, 10)), (n_sample, iters)):
        pb = _oracle_problem(orc, n, workload)
        tv = TorchVectorisedGD(pb, 0.1, 0.01, lr_t)
        tv.run(1)                                                                       # touch pages, build the graph once
        t0 = time.perf_counter()
        tv.run(it)
        dt = time.perf_counter() - t0
        head = {"mode": "V (vectorised CPU PyTorch, autograd + torch.optim.Adam)", "n_elems": n, "iterations": it,
                "value": n * it / dt, "unit": "element-evals/s", "wall_s": dt, "cores": int(torch.get_num_threads()),
                "torch_num_threads": int(torch.get_num_threads())}
        rows.append(head)
        del tv
    for n, it in ((100_000, max(iters, 10)), (n_sample, iters)):
        pb = _oracle_problem(orc, n, workload)
        cfg = orc.SolverConfig(max_iterations=it, learning_rate_u=0.01, learning_rate_theta=lr_t, tolerance=0.0)
        geo = orc.element_geometry(pb)
        orc.solve_gd(pb, orc.SolverConfig(max_iterations=1, tolerance=0.0), 0.1, geo=geo)  # touch pages
        t0 = time.perf_counter()
        orc.solve_gd(pb, cfg, 0.1, geo=geo)
        dt = time.perf_counter() - t0
        rows.append({"mode": "Vn (vectorised numpy oracle, no autograd)", "n_elems": n, "iterations": it,
                     "value": n * it / dt, "unit": "element-evals/s", "wall_s": dt,
                     "cores": f"BLAS pool {threads}; element-wise numpy work on 1"})
    if limiter is not None:
        limiter.restore_original_limits()
    torch.set_num_threads(torch_threads_before)
    return {"value": head["value"], "unit": "element-evals/s", "cores": head["cores"], "kind": "port",
            "sample": f"oracle/torch_vectorised.py (batched CPU PyTorch restatement of the reference iteration, autograd + "
                      f"torch.optim.Adam), {head['n_elems']} elements x {head['iterations']} GD iterations, {workload} "
                      f"shape, {head['wall_s']:.1f} s wall; torch.get_num_threads() = {head['cores']} of "
                      f"{_usable_cpus()} usable / {ncpu} logical CPUs ({model_name})",
            "cpu_model": model_name, "host_cpus": ncpu, "usable_cpus": _usable_cpus(), "rows": rows}


def iters_to_tol(dev, sizes=(3, 30, 1000, 10_000), max_iterations=5000, n_increments=10):
    """SURVEY.md section 8(d) metric (ii), "GD iterations to tolerance" (FEM/python/fem/solver.py:341-355): the synthetic
    chain (h = 1, measurements ux_i = x_i at every node, alpha_data = 100), example4 shape, seed 0, 10 load increments
    lam_k = k/10 with warm starts (fem/solver.py:1045-1167), stop test `it > 10 and (||r|| < tol or L < tol)` with
    tol = 1e-6, at most `max_iterations` per increment — in the reference's element-force order and in the delta form.
    Reported per increment: iterations and whether the increment hit max_iterations.  With the SURVEY's inputs the
    measured field is u = x at EVERY load factor and Adam moves a dof by at most lr_u = 0.01 per iteration, so an increment
    needs >= 100 x_max iterations: the tolerance is attainable within the cap for the example-sized bars (N = 3, 30) and
    every increment of the 10^3 / 10^4-element bars ends at max_iterations (conditioning ~ N^2) — which is
    why throughput at 10^5..10^7 elements is quoted at a fixed iteration count."""
    import torch
    from pinn_fem_amd.fem.solver import SolverConfig, solve_gd
    out = {}
    for n in sizes:
        for fe_name, fe in (("reference", 0), ("delta", 1)):
            model, mv, md, _ = build_model(n, "ex4")
            model._pf_fe_mode = fe
            cfg = SolverConfig(max_iterations=max_iterations, tolerance=1e-6, learning_rate_u=0.01,
                               learning_rate_theta=5e-4, alpha_physics=1.0, alpha_data=100.0)
            u, its, hits, last = None, [], 0, None
            t0 = time.perf_counter()
            for k in range(1, n_increments + 1):
                res = solve_gd(model, cfg, mv, md, target_load_factor=k / n_increments, u_initial=u)
                its.append(len(res.history))
                hits += 0 if res.converged else 1
                last = res.history[-1]
                u = torch.tensor(res.displacements.flatten(), dtype=torch.float32)
            torch.cuda.synchronize(dev)
            out[f"N={n}, fe_mode={fe_name}"] = {
                "iterations_per_increment": its, "iterations_total": int(sum(its)), "max_iterations": max_iterations,
                "max_iterations_hits": hits, "final_loss_total": last["loss_total"],
                "final_residual_norm": last["residual_norm"], "wall_s": time.perf_counter() - t0}
    return out


def side_config(workload, n_elems, dev, steps, mesh="chain"):
    """Same measurement as the headline (warm-up, K graph-replayed steps, sync on both sides) on another
    BASELINE.json configuration / another synthetic mesh."""
    import torch
    from pinn_fem_amd.engine import HipEngine
    from pinn_fem_amd.fem.solver import SolverConfig
    model, mv, md, widths = build_model(n_elems, workload, mesh=mesh)
    n_elems = len(model.elements)
    cfg = SolverConfig(max_iterations=steps + 18, tolerance=0.0, learning_rate_u=0.01,
                       learning_rate_theta=5e-4 if workload == "ex4" else 1e-3)
    eng = HipEngine(model, mv, md, device=dev)
    eng.begin(None, 0.1, cfg, want_history=False)
    eng.prepare_graph(chained=True)
    eng.iterate(eng.GRAPH_ITERS, defer_tail=True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    eng.iterate(steps, defer_tail=True)      # chained replays, like solve_gd's loop (the tail is paid once, by flush())
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    eng.flush()
    return {"value": n_elems * steps / dt, "unit": "element-evals/s", "ms_per_step": dt / steps * 1e3,
            "steps": steps}


def launch_workers(n: int) -> int:
    """python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same arguments>"""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--elems", type=int, default=1_000_000, help="elements per GPU")
    ap.add_argument("--workload", default="ex4", choices=["ex4", "ex3"])
    ap.add_argument("--cpu-sample", type=int, default=1_000_000)
    ap.add_argument("--cpu-iters", type=int, default=5)
    ap.add_argument("--cpu-threads", type=int, default=0, help="thread-pool size of the CPU baseline (0: min(64, cpus))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the extra configs[1] measurement")
    ap.add_argument("--repeat", type=int, default=5, help="further timed K-step regions for the repeat statistic")
    ap.add_argument("--total-elems", type=int, default=0,
                    help="elements of the WHOLE bar (overrides --elems with total/gpus); BASELINE.json configs[3] is "
                         "--gpus 8 --total-elems 10000000")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks here, as fresh child processes, BEFORE anything
        # in this process has touched the GPU (no torch import yet); this process only relays the exit code
        return launch_workers(args.gpus)

    import torch
    import torch.distributed as dist
    from pinn_fem_amd import _capi
    from pinn_fem_amd.fem.solver import SolverConfig

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         "python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    # PINNFEM_BENCH_ONE_GPU=1: rehearsal of the N>1 path on a single GPU (all ranks on cuda:0, gloo
    # collectives staged through the host) — for testing only, never for reported numbers
    rehearsal = os.environ.get("PINNFEM_BENCH_ONE_GPU", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    n_local = args.elems if args.total_elems <= 0 else args.total_elems // world
    lr_t = 5e-4 if args.workload == "ex4" else 1e-3
    cfg = SolverConfig(max_iterations=max(args.warmup, 50) + (3 + max(args.repeat, 0)) * args.steps + 64, tolerance=0.0,
                       learning_rate_u=0.01, learning_rate_theta=lr_t, alpha_physics=1.0, alpha_data=100.0)
    if world == 1:
        from pinn_fem_amd.engine import HipEngine
        model, mv, md, widths = build_model(n_local, args.workload)
        eng = HipEngine(model, mv, md, device=dev)
        eng.begin(None, 0.1, cfg, want_history=False)
        eng.prepare_graph(chained=True)               # capture + instantiate: never inside the timed region
        gk = eng.GRAPH_ITERS
        # >= W, whole replays (the graph is uploaded and hot), and at least 50 iterations (~8 ms): the chip's clocks are still
        # settling during the first replays after the event pass (the first 20-step region read 3-5 us per step above the
        # following ones); the number actually run is reported as warmup_iterations_run
        n_warm = ((max(args.warmup, 50) + gk - 1) // gk) * gk
        # chained replays, as in solve_gd's loop: a replay hands the updates and the bookkeeping of its last iteration to
        # the next replay's first one (every iteration of a replay carries its predecessor's anyway); each replay of K
        # iterations still does K forward / residual / backward passes, K parameter and displacement updates and K
        # bookkeeping steps.  flush() pays the one pending tail before the state is read.
        run_warm = lambda n: eng.iterate(n, defer_tail=True)
        run_timed = lambda n: eng.iterate(n, defer_tail=True)     # hipGraph replays (20 iterations per graph at 10^6 elements)
        run_events = lambda n: (eng.flush(), eng.iterate_timed(n))[1]   # eager launches with HIP events around every kernel
        graph_count = lambda: eng.graph_creates
    else:
        from pinn_fem_amd.dist import ShardedChainEngine
        widths = {"ex4": (20, 15, 10), "ex3": (20, None, None)}[args.workload]
        eng = ShardedChainEngine(n_local, args.workload, rank, world, dev)
        eng.begin(None, 0.1, cfg)
        eng.prepare()                                 # the C driver's RCCL communicator (collective)
        n_warm = max(args.warmup, 50)
        run_warm = lambda n: eng.iterate(n)
        run_timed = lambda n: eng.iterate(n)
        run_events = lambda n: eng.iterate_timed(n)
        graph_count = lambda: 0

    # (1) K steps launched eagerly with a HIP event before every kernel on the launch stream: the per-kernel
    # durations of the roofline (not part of `value`).  Run first, so that the timed region below starts on a chip
    # that has already left its idle clocks; the W warm-up steps still come right before the timed K.
    t1 = time.perf_counter()
    slot_ms = run_events(args.steps)
    torch.cuda.synchronize(dev)
    dt_events = time.perf_counter() - t1
    iters_before = args.steps
    if world > 1:
        # the event pass ran this rank's kernels WITHOUT the collective, so the replicated state (theta, interface
        # dofs) has drifted apart between ranks: start the sharded solve again (theta re-broadcast, fresh Adam state)
        eng.begin(None, 0.1, cfg)
        eng.prepare()
        iters_before = 0
    # (2) W untimed warm-up steps — the last K of them issued exactly like the timed region (synchronise, K steps,
    # synchronise): the first region issued that way after a burst of back-to-back replays read 3-6 us per step above all
    # later ones whatever the length of the burst (60, 200, 500 iterations) — (3) EXACTLY K timed steps between barrier +
    # synchronize
    run_warm(n_warm)
    torch.cuda.synchronize(dev)
    run_timed(args.steps)
    n_warm += args.steps
    graphs_before = graph_count()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run_timed(args.steps)                    # the product path (hipGraph replay)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if world == 1:
        eng.flush()                          # (outside the timed region: the one pending tail of the chained replays)
    st = eng.state()
    assert st.iter == iters_before + n_warm + args.steps, (st.iter, iters_before + n_warm + args.steps)
    assert graph_count() == graphs_before, "a hipGraph was captured inside the timed region"
    # repeat statistic (not `value`): R further regions of K steps each, timed the same way
    rep_ms = []
    for _ in range(max(args.repeat, 0)):
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        t2 = time.perf_counter()
        run_timed(args.steps)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        d2 = time.perf_counter() - t2
        if world > 1:
            tm = torch.tensor([d2], dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            d2 = float(tm.item())
        rep_ms.append(d2 / args.steps * 1e3)

    mlp_dtype = (eng if world == 1 else eng.backend.eng).mlp_dtype
    # N > 1: BASELINE.json configs[3] is a 10^7-element bar sharded over the ranks (1.25e6 per rank at N = 8), which the
    # weak-scaling headline (--elems per rank) does not reproduce: measured here as well, the same way (collective: every
    # rank takes part), unless the headline already is that configuration
    also_c3 = None
    if world > 1 and not args.no_also and n_local * world != 10_000_000:
        from pinn_fem_amd.dist import ShardedChainEngine
        n3 = (10_000_000 if not rehearsal else 100_000) // world
        eng3 = ShardedChainEngine(n3, args.workload, rank, world, dev)
        eng3.begin(None, 0.1, cfg)
        eng3.prepare()
        eng3.iterate(max(args.warmup, 1))
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)
        t3 = time.perf_counter()
        eng3.iterate(args.steps)
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)
        d3 = time.perf_counter() - t3
        tm = torch.tensor([d3], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        d3 = float(tm.item())
        also_c3 = {"value": n3 * world * args.steps / d3, "unit": "element-evals/s", "ms_per_step": d3 / args.steps * 1e3,
                   "steps": args.steps, "elements_total": n3 * world, "elements_per_gpu": n3, "scaling": "strong (fixed total)"}
        del eng3
    if rank == 0:
        total_elems = n_local * world
        value = total_elems * args.steps / dt
        names = _capi.KERNEL_SLOT_NAMES
        fused = (eng if world == 1 else eng.backend.eng).fusion_info()
        dom = int(np.argmax(slot_ms))
        # roofline of the dominant kernel.  The net_backward kernels are bound by f32 matrix/vector
        # throughput (dense MLP GEMM work); the node kernels by HBM.
        if names[dom].startswith("net_backward"):
            both = bool(fused & _capi.PF_FUSED_BACKWARD)         # ONE launch carries the backward of both nets
            if both:
                flops, kname = sum(net_flops(w) for w in widths[:2] if w is not None), "net_backward (young + area, one launch)"
            else:
                flops, kname = net_flops(widths[0] if names[dom].endswith("young") else widths[1]), names[dom]
            # the launch's own duration: HIP events around 20 back-to-back launches on the engine's stream (the eager
            # slot time above also contains the gap to the next launch of the event pass)
            launch_ms = float(slot_ms[dom])
            if both:
                try:
                    launch_ms = float((eng if world == 1 else eng.backend.eng).time_step_launch("backward", 20))
                except Exception:
                    pass
            ach = flops * n_local / (launch_ms * 1e-3) / 1e12
            roof = {"kernel": kname, "bound": "mfma", "achieved": ach, "peak": F32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": ach / F32_PEAK_TFLOPS, "traffic": None,
                    "avg_launch_ms": launch_ms, "eager_slot_ms": float(slot_ms[dom]),
                    "note": "algorithmic float32 flops (forward recompute + backward) against the dense f32 peak; the "
                            "engine executes every f32-grade product as three f16 MFMAs (2-way split operands), and the "
                            "kernel is bound by vector-ALU / transcendental issue, not by the matrix pipe (DESIGN.md §4)"}
        else:
            # share of the 144 B/eval the kernel is responsible for is not separable: price the
            # whole per-eval figure against this kernel's time (upper bound on its byte rate)
            ach = ALGO_BYTES_PER_EVAL * n_local / (slot_ms[dom] * 1e-3) / 1e9
            roof = {"kernel": names[dom], "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                    "avg_launch_ms": float(slot_ms[dom])}
        # HBM traffic of the dominant kernel: rocprofv3 PMC (FETCH_SIZE, WRITE_SIZE in separate passes,
        # gfx950 correction applied) of this same command, committed under profiles/
        try:
            with open(os.path.join(ROOT, "profiles", "r03_traffic.json")) as f:
                tr = json.load(f)["kernels"]
            key = {"net_backward_young": "k_net32_backward2<10, 8, 2, 3>" if fused & _capi.PF_FUSED_BACKWARD
                   else "k_net32_backward<10, 2, 3, true>", "net_backward_area": "k_net32_backward<8, 2, 3, false>",
                   "node_residual": "k_node_residual<2>", "node_gradu_adam": "k_node_gradu<2, true>"}.get(names[dom])
            if key in tr and n_local == 1_000_000 and args.workload == "ex4":
                roof["traffic"] = tr[key]["hbm_bytes_corrected"]
                roof["traffic_source"] = "profiles/r03_traffic.json (rocprofv3 --pmc, per launch)"
        except Exception:
            pass
        it_bytes = ALGO_BYTES_PER_EVAL * value / world / 1e9
        roof["iteration_hbm_GBs_per_gpu"] = it_bytes
        roof["iteration_hbm_frac"] = it_bytes / HBM_PEAK_GBS
        out = {
            "metric": "element-residual-grad evals/sec (PINN+GD iteration: MLP fwd+bwd, stiffness, "
                      "assembly, residual, Adam), collinear truss",
            "value": value, "unit": "element-evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "warmup_iterations_run": n_warm, "graph_captured_before_timing": world == 1,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            # arithmetic type of the path: f32 throughout, unless the optional bf16-operand MLP was switched on
            # (PINNFEM_MLP_DTYPE=bf16: hidden-layer products on bf16 operands, everything else f32)
            "dtype": "f32" if mlp_dtype == "f32" else "bf16 (hidden-layer products; f32 elsewhere)",
            "data": "synthetic" + (" (ONE-GPU REHEARSAL: not a valid multi-GPU number)" if rehearsal else ""),
            "config": {"workload": f"{args.workload} shape (nets {widths}, E and A evaluated), "
                                   f"{n_local} elements per GPU, collinear 2-D truss h=1, "
                                   f"alpha_data=100 with measurements at every node, load factor 0.1",
                       "elements_total": total_elems, "parallelism": f"elements sharded x{world}"},
            "repeat": ({"regions": len(rep_ms), "steps_each": args.steps, "ms_per_step_each": [float(x) for x in rep_ms],
                        "ms_per_step_median": float(np.median(rep_ms)),
                        "ms_per_step_min": float(np.min(rep_ms)), "ms_per_step_max": float(np.max(rep_ms)),
                        "value_at_median": total_elems * 1e3 / float(np.median(rep_ms))} if rep_ms else None),
            "ms_per_step_eager_with_events": dt_events / args.steps * 1e3,
            "kernel_ms": {n: float(m) for n, m in zip(names, slot_ms)},
            "fused_launches": {"forward_young+area": bool(fused & _capi.PF_FUSED_FORWARD),
                               "backward_young+area": bool(fused & _capi.PF_FUSED_BACKWARD),
                               "theta_update_in_next_forward (graph)": bool(fused & _capi.PF_FUSED_THETA_UPDATE),
                               "displacement_ping_pong (graph)": bool(fused & _capi.PF_FUSED_U_PINGPONG),
                               "displacement_update_in_next_forward (graph)": bool(fused & _capi.PF_FUSED_U_UPDATE),
                               "replays_chained (a replay hands its last iteration's updates and bookkeeping to the next one)":
                                   bool(world == 1 and all(getattr(eng, "_chain_graphs", {}).get(f) for f in (2, 3))),
                               "note": "a fused launch is booked on the first of its two kernel_ms slots"},
            "roofline": roof,
        }
        if world > 1:
            from pinn_fem_amd.dist import shard_driver_info
            out["config"].update(shard_driver_info(eng.backend))
            if also_c3 is not None:
                name = "configs[3]: 1e7-element bar sharded over the ranks" if not rehearsal else "configs[3] rehearsal (1e5 total)"
                out["also"] = {name: also_c3}
        if world == 1 and not args.no_also:
            # BASELINE.json configs[1] (example3 shape, E = NN, 10^5 elements) for the record; the headline
            # `value` above stays the 10^6-element configuration the metric is quoted on
            out["also"] = {"configs[1]: ex3 shape, 1e5 elements, 1 GPU": side_config("ex3", 100_000, dev, args.steps),
                           # a genuinely 2-D truss (Warren girder, node degree 4) of the headline's size
                           "ex4 shape, Warren girder, 1e6 elements, 1 GPU": side_config("ex4", 1_000_000, dev, args.steps,
                                                                                         mesh="warren")}
            # metric (ii) of SURVEY.md section 8(d): GD iterations to tolerance on the synthetic bar
            out["also"]["iters_to_tol"] = iters_to_tol(dev)
        if not args.no_cpu_baseline and world == 1:
            threads = args.cpu_threads or min(64, _usable_cpus())
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_sample, args.cpu_iters, threads)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        from pinn_fem_amd.dist import destroy_rccl_comms
        destroy_rccl_comms()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
