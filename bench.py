#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X Chambolle-Pock hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json config 2): server_heat tree, nx=5, nu=2, branching 2, 4096 leaves (reference N=13),
fp64, synthetic deterministic data.  Every rank holds `--batch` independent problem instances (own x0, own
state); a STEP is `--iters` Chambolle-Pock iterations (one evaluation of T = (prox_f, prox_h*) + L, L' + the
termination residuals each) of every instance, run as ONE solve call with tol = 0 so that no instance stops
early: the work per step is fixed.  All inputs are resident in HBM before the timed region (a step uploads
only the nx-double x0 per instance and reads back one 120-byte control block per instance).

value = n_gpus * batch * iters * steps / time  ("Spock iterations per second", whole job).

The JSON line also carries
  roofline      the dominant kernel (subtree phase-B kernel): algorithmic bytes per launch / its average HIP-event
                duration, measured live after the timed region on the stream the kernels run on
  cpu_baseline  the CPU oracle (single-thread C restatement of the reference) on ONE instance of the same
                workload, on this box's host cores (rank 0, N=1 only)
  single_instance  latency-bound regime: iterations/s of ONE instance and ms-to-tol (tol 1e-3, SILENT rule) of
                the reference-faithful nu=5 variant of the same tree, next to the oracle's time for the same solve
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec), ~6.3 TB/s achievable


def algorithmic_bytes(flat):
    """SURVEY.md section 8(d), W = 8 (fp64), per instance and CP iteration."""
    W = 8
    nz, nv = flat.nz, flat.nv
    qd = flat.n * flat.nx + flat.nn * flat.nu
    phase_a = W * (2 * nz + nv + qd)                 # read z, v, write zbar + q/d write
    phase_b = W * (4 * nz + 4 * nv + qd)             # read zbar, z, v, write vbar + q/d read + termination 2(nz+nv)
    return {"sub_a": phase_a, "sub_b": phase_b, "cp_iteration": phase_a + phase_b}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="problem instances per GPU (4.4 MB of state each)")
    ap.add_argument("--iters", type=int, default=100, help="CP iterations per step (even)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")

    import spock_jl_amd as S
    from spock_jl_amd import dist as sdist

    N_REF, NX, NU, D = 13, 5, 2, 2
    B, M = args.batch, args.iters + (args.iters & 1)
    specs = S.get_server_heat_specs(N_REF, NX, D, nu=NU)
    flat = S.flatten_problem(*specs)
    model = S.build_model(*specs, S.SolverOptions(S.CP, None), batch=B, device=local_rank, max_iter=M, graph_iters=M)
    inst0 = rank * B
    x0 = 0.1 * (1.0 + 0.001 * (inst0 + np.arange(B)))[:, None] * np.ones((B, NX))

    def step():
        return S.solve_model(model, x0, tol=0.0)     # exactly M iterations of every instance (status MAX_ITER)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    elapsed = time.perf_counter() - t0
    res = res if isinstance(res, list) else [res]
    assert all(r["iterations"] == M for r in res), "a step must run exactly --iters iterations"
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the path's only collective: the trivial result gather (nu + 5 doubles per instance)
        rec = sdist.gather_records(sdist.instance_record(model, res, NU), [B] * world, device=torch.device("cuda", local_rank))
        assert rec.shape == (B * world, NU + 5)
    value = n_gpus * B * M * args.steps / elapsed

    # ---- roofline of the dominant kernel: HIP events around every launch, same workload, same stream
    bytes_ = algorithmic_bytes(flat)
    prof = model.profile_kernels(40)
    split = model.sizes.split_stage > 0
    dom = max(("sub_a", "crown", "sub_b", "full"), key=lambda k: prof[k][0] * (prof[k][1] > 0))
    dom_ms = prof[dom][0]
    fused = bool(model.sizes.fused_iteration)
    if fused:   # "sub_b" is then the fused SUB_BA kernel: phase B of iteration k + phase A / backward sweep of k + 1 = one whole iteration
        bytes_["sub_b"] = bytes_["cp_iteration"]
    dom_bytes = B * (bytes_[dom] if dom in bytes_ else bytes_["cp_iteration"])
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    # measured HBM bytes per launch of that kernel: PMC counters cannot be read from inside this process; they come
    # from the committed rocprofv3 --pmc passes of this same command (profiles/<round>/pmc_b<batch>_summary.json,
    # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), null when no pass exists for this batch size
    traffic = None
    for rnd in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        f = os.path.join(ROOT, "profiles", rnd, f"pmc_b{B}_summary.json")
        if os.path.exists(f):
            k = json.load(open(f)).get("kernels", {}).get("sub_ba" if (fused and dom == "sub_b") else dom)
            if k:
                traffic = k["hbm_bytes_per_launch_corrected"]
            break
    roofline = {"bound": "hbm", "kernel": {"sub_a": "spock_group_kernel<MODE_SUB_A>", "sub_b": "spock_group_kernel<MODE_SUB_BA>" if fused else "spock_group_kernel<MODE_SUB_B>",
                                             "crown": "spock_group_kernel<MODE_CROWN>", "full": "spock_group_kernel<MODE_FULL>"}[dom],
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic, "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": round(dom_ms, 5),
                "launches_timed": prof[dom][1],
                "all_kernels_avg_ms": {k: round(v[0], 5) for k, v in prof.items() if v[1]},
                "whole_iteration_frac": round(B * bytes_["cp_iteration"] * n_gpus / (elapsed / (M * args.steps)) / 1e9 / (HBM_PEAK_GBS * n_gpus), 4)}
    model.close()

    out = {"metric": "Spock (Chambolle-Pock) iterations/sec on the server_heat tree", "value": round(value, 1),
           "unit": "iterations/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "server_heat nx=5 nu=2 d=2, 4096 leaves (reference N=13), CP iteration incl. termination residuals",
                      "instances_per_gpu": B, "iterations_per_step": M, "nz": flat.nz, "nv": flat.nv,
                      "split_stage": int(model.sizes.split_stage), "algorithm": "CP", "fused_iteration": fused,
                      "parallelism": f"{n_gpus} x {B} independent instances, RCCL all_gather of results only"},
           "roofline": roofline}

    if rank == 0 and world == 1:
        from oracle import oracle as O
        if not args.no_cpu_baseline:
            # bounded CPU sample of the same workload: ONE instance, as many CP iterations as fit ~12 s
            o = O.Oracle(flat, O.CP)
            t = time.perf_counter(); o.solve(x0[0], tol=0.0, max_iter=50); per = (time.perf_counter() - t) / 50
            k = max(50, int(12.0 / per))
            t = time.perf_counter(); o.solve(x0[0], tol=0.0, max_iter=k); dt = time.perf_counter() - t
            out["cpu_baseline"] = {"value": round(k / dt, 2), "unit": "iterations/s", "cores": 1, "kind": "port",
                                   "host_cores_available": os.cpu_count(),
                                   "sample": f"{k} CP iterations of one instance of the same workload, single-thread C oracle "
                                             f"(the reference is Julia; no Julia toolchain exists on this box)"}
        if not args.no_single:
            single = {}
            m1 = S.build_model(*specs, S.SolverOptions(S.CP, None), batch=1, device=local_rank, max_iter=2000, graph_iters=100)
            S.solve_model(m1, x0[:1], tol=0.0)
            torch.cuda.synchronize(); t = time.perf_counter(); r = S.solve_model(m1, x0[:1], tol=0.0); dt = time.perf_counter() - t
            single["iterations_per_s"] = round(r["iterations"] / dt, 1)
            single["us_per_iteration_device"] = round(r["solve_ms"] / r["iterations"] * 1e3, 2)
            m1.close()
            specs5 = S.get_server_heat_specs(N_REF, NX, D)           # reference-faithful nu = nx = 5: converges
            flat5 = S.flatten_problem(*specs5)
            m5 = S.build_model(*specs5, S.SolverOptions(S.CP, None), batch=1, device=local_rank)
            S.solve_model(m5, np.full(NX, 0.1), tol=1e-3); m5.reset()
            r5 = S.solve_model(m5, np.full(NX, 0.1), tol=1e-3)
            single["ms_to_tol_1e-3_nu5"] = round(r5["solve_ms"], 3)
            single["iterations_to_tol_nu5"] = r5["iterations"]
            m5.close()
            # the Spock algorithm proper (SuperMann-accelerated CP, SolverOptions(SP, nothing)) on the same convergent tree
            ms = S.build_model(*specs5, S.SolverOptions(S.SP, None), batch=1, device=local_rank)
            S.solve_model(ms, np.full(NX, 0.1), tol=1e-3); ms.reset()
            rs = S.solve_model(ms, np.full(NX, 0.1), tol=1e-3)
            single["sp_ms_to_tol_1e-3_nu5"] = round(rs["solve_ms"], 3)
            single["sp_iterations_to_tol_nu5"] = rs["iterations"]
            single["sp_iterations_per_s"] = round(rs["iterations"] / (rs["solve_ms"] * 1e-3), 1)
            ms.close()
            if not args.no_cpu_baseline:
                osp = O.Oracle(flat5, O.SP)
                t = time.perf_counter(); rosp = osp.solve(np.full(NX, 0.1), tol=1e-3); dtsp = time.perf_counter() - t
                single["cpu_oracle_sp_ms_to_tol_nu5"] = round(dtsp * 1e3, 1)
                single["cpu_oracle_sp_iterations_nu5"] = rosp["iterations"]
                o5 = O.Oracle(flat5, O.CP)
                t = time.perf_counter(); ro = o5.solve(np.full(NX, 0.1), tol=1e-3); dt5 = time.perf_counter() - t
                single["cpu_oracle_ms_to_tol_nu5"] = round(dt5 * 1e3, 1)
                single["cpu_oracle_iterations_nu5"] = ro["iterations"]
            out["single_instance"] = single
            # problems/s: 64 independent instances of the convergent (nu = 5) tree, different x0, solved to tol 1e-3 as one batch
            probs = {}
            X = 0.1 * (1.0 + 0.01 * np.arange(64))[:, None] * np.ones((64, NX))
            for name, alg in (("cp", S.CP), ("sp", S.SP)):
                mb = S.build_model(*specs5, S.SolverOptions(alg, None), batch=64, device=local_rank)
                S.solve_model(mb, X, tol=1e-3); mb.reset()
                torch.cuda.synchronize(); t = time.perf_counter(); rb = S.solve_model(mb, X, tol=1e-3); torch.cuda.synchronize()
                dtb = time.perf_counter() - t
                assert all(q["status"] == 0 for q in rb)
                probs[name] = {"instances": 64, "ms": round(dtb * 1e3, 2), "problems_per_s": round(64 / dtb, 1),
                               "iterations_max": max(q["iterations"] for q in rb)}
                mb.close()
            out["problems_to_tol_1e-3_nu5"] = probs

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
