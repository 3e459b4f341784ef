"""Multi-GPU: independent problem instances shard across ranks, one process per GPU.

The path has no data-path exchange: a closed-loop simulation / Monte-Carlo draw / warm-started MPC step is
solved entirely on one GPU (reference: examples/server_heat/mpc_simulation.jl:38-48 runs
its M simulations one after the other).  Every rank builds the same problem data from the same deterministic
generator, takes a contiguous block of the instances, solves them as one batch, and a single all_gather
(RCCL over xGMI on GPUs, gloo on CPUs) collects one fixed-size record per instance:
    [u0 (nu doubles), iterations, status, xi1, xi2, solve_ms]            -> nu + 5 doubles.

``mpc_simulation_sharded`` shards whole closed-loop simulations the same way (BASELINE.json config 4: the M simulations
of examples/server_heat/mpc_simulation.jl:38-48,164-177, each 20 warm-started solves, 8 per GPU on an 8-GPU node).
"""
from __future__ import annotations

import numpy as np


def shard_range(n_instances: int, world_size: int, rank: int):
    """Contiguous block of instances of `rank`: instance i lives on rank floor(i * world_size / n_instances)."""
    lo = -(-rank * n_instances // world_size)
    hi = -(-(rank + 1) * n_instances // world_size)
    return lo, min(hi, n_instances)


def instance_record(model, results, nu: int) -> np.ndarray:
    """[batch, nu + 5]: first-stage input u0 = z[u_inds[:nu]] (mpc_simulation.jl:164) and the solve summary."""
    z = model.solver_state.z.reshape(model.batch, -1)
    u0 = z[:, model.solver_state_internal.u_inds[:nu]]
    res = results if isinstance(results, list) else [results]
    tail = np.array([[r["iterations"], r["status"], r["xi1"], r["xi2"], r["solve_ms"]] for r in res], dtype=np.float64)
    return np.concatenate([u0, tail], axis=1)


def gather_records(local: np.ndarray, counts, device=None) -> np.ndarray:
    """all_gather of the per-instance records (ragged blocks are padded to the largest block)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    width = local.shape[1]
    pad = max(counts)
    buf = torch.zeros((pad, width), dtype=torch.float64, device=device)
    if local.shape[0]:
        buf[: local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local)).to(buf.device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return np.concatenate([o[: counts[r]].cpu().numpy() for r, o in enumerate(out)], axis=0)


def solve_sharded(specs, x0_all: np.ndarray, tol: float = 1e-3, options=None, device_index: int = 0, torch_device=None,
                  **model_kw):
    """Solve len(x0_all) independent instances of one problem, sharded over the ranks of the initialised
    torch.distributed group (or locally when there is none).  Returns the gathered [n_instances, nu + 5] records."""
    import torch.distributed as dist
    from .model import SolverOptions, build_model, solve_model, CP
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    n = int(x0_all.shape[0])
    counts = [shard_range(n, world, r)[1] - shard_range(n, world, r)[0] for r in range(world)]
    lo, hi = shard_range(n, world, rank)
    nu = specs[2].B[0].shape[1]
    if hi > lo:
        model = build_model(*specs, options or SolverOptions(CP, None), batch=hi - lo, device=device_index, **model_kw)
        res = solve_model(model, x0_all[lo:hi], tol=tol)
        local = instance_record(model, res, nu)
        model.close()
    else:
        local = np.zeros((0, nu + 5))
    return gather_records(local, counts, device=torch_device)


def mpc_record(xt: np.ndarray, ut: np.ndarray, res) -> np.ndarray:
    """[batch, nu + 5] per closed-loop simulation: the first applied input u_0 (nu), the total number of solver iterations
    over the steps, the worst solve status, |x_T|_inf of the final state, the number of steps, the summed device time (ms)."""
    B, steps = ut.shape[0], ut.shape[1]
    tail = np.array([[sum(r["iterations"] for r in res[b]), max(r["status"] for r in res[b]), float(np.max(np.abs(xt[b, -1]))),
                      steps, sum(r["solve_ms"] for r in res[b])] for b in range(B)], dtype=np.float64).reshape(B, 5)
    return np.concatenate([ut[:, 0, :], tail], axis=1)


def mpc_simulation_sharded(specs, x0_all: np.ndarray, w_all: np.ndarray, tol: float = 1e-3, options=None, model=None,
                           device_index: int = 0, torch_device=None, **model_kw):
    """Closed-loop simulations (mpc_simulation.jl:38-48: `for m = 1:M` ... `for t = 1:T` warm-started solve, apply u, draw
    w, advance x0) sharded over the ranks of the initialised torch.distributed group: simulation i runs entirely on rank
    floor(i * world / n) (contiguous blocks), through spock_mpc_simulate as one batch per rank; no data-path collective, one
    all_gather of an (nu + 5)-double record per simulation at the end.  ``model``: a prebuilt model whose batch equals this
    rank's block (the reference builds outside its timed region, mpc_simulation.jl:40,48); otherwise one is built here.
    Returns (records [n, nu + 5] gathered on every rank, x_traj, u_traj of the LOCAL block)."""
    import torch.distributed as dist
    from .model import SolverOptions, build_model, mpc_simulation, SP
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    n = int(x0_all.shape[0])
    counts = [shard_range(n, world, r)[1] - shard_range(n, world, r)[0] for r in range(world)]
    lo, hi = shard_range(n, world, rank)
    nu = specs[2].B[0].shape[1]
    nx = specs[2].B[0].shape[0]
    steps = int(np.asarray(w_all).shape[1])
    if hi > lo:
        own = model is None
        if own:
            model = build_model(*specs, options or SolverOptions(SP, None), batch=hi - lo, device=device_index, **model_kw)
        if model.batch != hi - lo:
            raise ValueError("model.batch must equal this rank's block of simulations")
        xt, ut, res = mpc_simulation(model, x0_all[lo:hi], np.asarray(w_all)[lo:hi], tol=tol)
        local = mpc_record(xt, ut, res)
        if own:
            model.close()
    else:
        xt = np.zeros((0, steps + 1, nx)); ut = np.zeros((0, steps, nu)); local = np.zeros((0, nu + 5))
    return gather_records(local, counts, device=torch_device), xt, ut
