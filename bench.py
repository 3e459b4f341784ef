#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X Chambolle-Pock / Spock hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a child
`python -m torch.distributed.run ...`; the parent never touches a GPU) and passes rank 0's JSON line through.

Workload of `value` (BASELINE.json config 2): server_heat tree, nx=5, nu=2, branching 2, 4096 leaves (reference N=13),
fp64, synthetic deterministic data.  Every rank holds `--batch` independent problem instances (own x0, own state); a STEP
is `--iters` Chambolle-Pock iterations (one evaluation of T = (prox_f, prox_h*) + L, L' + the termination residuals each)
of every instance, run as ONE solve call with tol = 0 so that no instance stops early: the work per step is fixed.  All
inputs are resident in HBM before the timed region (a step uploads only the nx-double x0 per instance and reads back one
120-byte control block per instance).

    value = n_gpus * batch * iters * steps / time      ("Spock iterations per second", whole job, weak scaling)

The JSON line also carries
  roofline        the dominant kernel (LBA2: the level kernel of the two bottom stages N-3 and N-2): `bytes_per_launch` = the entry-by-entry count of what that
                  kernel must move (`bytes_model` says which count), `avg_launch_ms` its average HIP-event duration measured live
                  after the timed region on the stream the kernels run on, `achieved` / `frac` = that rate against the 8 TB/s HBM peak;
                  `traffic` (`traffic_raw`) = PMC bytes of the same kernel per launch from the committed rocprofv3 passes of the SAME
                  library build (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; null when the committed pass is of
                  another build), `frac_measured` the rate on them; `survey_8d_*` = SURVEY.md 8(d)'s stream model of an unfused
                  implementation, kept beside it (on that model a fused kernel "exceeds" the peak: it counts bytes nobody moves).
                  Whole iteration: `whole_iteration_necessary_bytes` / `whole_iteration_frac` (necessary bytes over the timed
                  region's time per iteration), `traffic_iteration` = PMC bytes of ALL launches of one iteration,
                  `frac_iteration_measured` = those over the same time, `waste_ratio` = traffic_iteration / necessary bytes
                  (re-reads the level structure adds).  tools/pmc_summary.py recomputes all of it from profiles/<round>/*.csv
  parity          after the timed region: the first instance, one of a middle tile and the last one of the batch against the CPU
                  oracle run for the same iterations (per-block relative error of z and v); `parity_max_relerr` is the largest
  cpu_baseline    the CPU oracle (single-thread C restatement of the reference) on ONE instance of the same workload, on
                  this box's host cores (rank 0, N=1 only), plus the same with one oracle process per host core
  single_instance latency-bound regime (N=1 only)
  mpc_cfg4        BASELINE config 4, the metric's "1/2/4/8-GPU problems/sec": closed-loop simulations of
                  examples/server_heat/mpc_simulation.jl (nx = nu = 20, N = 10, SP, 20 warm-started steps each, tol 1e-3)
                  sharded over the ranks through dist.mpc_simulation_sharded -- weak (8 simulations per GPU) and strong
                  (64 in total); problems = closed-loop simulations, solves = simulations x 20
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec), ~6.3 TB/s achievable


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2048, help="problem instances per GPU (4.4 MB of state each; 2048 = 32 tiles of 64 lanes)")
    ap.add_argument("--iters", type=int, default=96, help="CP iterations per step (rounded up to a multiple of 6)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true")
    ap.add_argument("--no-mpc", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle self-check of the headline state (profiling runs)")
    ap.add_argument("--parity-max-iters", type=int, default=4000, help=argparse.SUPPRESS)
    ap.add_argument("--mpc-per-gpu", type=int, default=8)
    ap.add_argument("--mpc-total", type=int, default=64)
    # test hooks (tests/test_bench_contract.py drives the whole N = 2 path on CPUs with the emulation library)
    ap.add_argument("--tree-N", type=int, default=13, help=argparse.SUPPRESS)
    ap.add_argument("--mpc-N", type=int, default=10, help=argparse.SUPPRESS)
    ap.add_argument("--mpc-nx", type=int, default=20, help=argparse.SUPPRESS)
    ap.add_argument("--mpc-steps", type=int, default=20, help=argparse.SUPPRESS)
    ap.add_argument("--lib", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def launch_command(args, argv, port):
    """The command the parent runs for `--gpus N` (N > 1): one process per GPU under torch.distributed.run."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def algorithmic_bytes(flat):
    """SURVEY.md section 8(d), W = 8 (fp64), per instance and CP iteration."""
    W = 8
    nz, nv = flat.nz, flat.nv
    qd = flat.n * flat.nx + flat.nn * flat.nu
    phase_a = W * (2 * nz + nv + qd)                 # read z, v, write zbar + q/d write
    phase_b = W * (4 * nz + 4 * nv + qd)             # read zbar, z, v, write vbar + q/d read + termination 2(nz+nv)
    return {"sub_a": phase_a, "sub_b": phase_b, "cp_iteration": phase_a + phase_b}


def necessary_bytes(flat):
    """What the fused iteration kernel must move per instance and CP iteration, entry by entry (W = 8): it reads the iterate
    (z, v), the s / tau / y part of zbar that the previous launch's phase A produced, and d; it writes zbar, vbar and d.
    (The termination test re-reads nothing, q never leaves LDS / registers, zbar's x / u go from the sweep to phase B in
    registers.)  Smaller than SURVEY 8d's model, which counts an unfused implementation's streams."""
    W = 8
    szy = flat.n + (flat.n - 1) + flat.nn * flat.ny
    d = flat.nn * flat.nu
    return W * ((flat.nz + flat.nv + szy + d) + (flat.nz + flat.nv + d))


def lba_necessary_bytes(flat):
    """Bottom-stage level kernel LBA (spock_ilevel.hpp), per instance and launch, entry by entry (W = 8).  Per node of stage N-2:
    reads   X: x_i, d_i, z.x_i, z.u_i, v7 | R: zbar.y, z.y, zbar.s_i, z.s_i, v1, v2 | C_c: zbar.tau, z.tau, v3..v6 | L_c: z.x_leaf, v11..v14, zbar.s, z.s
    writes  X: u_i, vbar.v7, d_i, q_i | R: vbar.v1, v2, next zbar's y, s_ch, tau_ch | C_c: vbar.v3..v6 | L_c: x_leaf, vbar.v11..v14
    (the roles' re-loads of x_i, d_i, z.x_i, z.u_i hit the L2 and are not counted)."""
    W, nx, nu, ny, d = 8, flat.nx, flat.nu, flat.ny, flat.d
    nxu = nx + nu
    reads = 3 * nxu + (3 * ny + 3) + d * (nxu + 4) + d * (3 * nx + 4)
    writes = (2 * nxu + nu) + (2 * ny + 1 + 2 * d) + d * (nxu + 2) + d * (3 * nx + 2)
    n_bottom = flat.nl // d
    return W * n_bottom * (reads + writes)


def lba2_necessary_bytes(flat):
    """The two bottom stages as one launch, LBA2 (spock_ilevel.hpp), per instance and launch, entry by entry (W = 8).  The nodes of
    stage N-2 as in lba_necessary_bytes minus what now stays in LDS (their x_i is not read back, their q_i not written); per node of
    stage N-3 (children = non-leaf nodes):
    reads   X: x_i, d_i, z.x_i, z.u_i, v7 | R: zbar.y, z.y, zbar.s_i, z.s_i, v1, v2 | C_c: zbar.tau, z.tau, v3..v6
    writes  X: u_i, x of the children, vbar.v7, d_i, q_i | R: vbar.v1, v2, next zbar's y, s_ch, tau_ch | C_c: vbar.v3..v6"""
    W, nx, nu, ny, d = 8, flat.nx, flat.nu, flat.ny, flat.d
    nxu = nx + nu
    reads = 3 * nxu + (3 * ny + 3) + d * (nxu + 4)
    writes = (nu + d * nx + nxu + nu + nx) + (2 * ny + 1 + 2 * d) + d * (nxu + 2)
    n_bottom = flat.nl // d
    n_parent = n_bottom // d
    return lba_necessary_bytes(flat) - W * n_bottom * 2 * nx + W * n_parent * (reads + writes)


def block_relerr(flat, which, a, b):
    """Largest per-block relative error of a z- or v-shaped vector against the oracle's (blocks x|u|s|tau|y, v1..v14; a block's
    scale floored at 1e-3 of max(1, |b|_inf) -- the measure of tests/conftest.py)."""
    import numpy as np
    if which == "z":
        blocks = [(sl.start, sl.stop) for sl in flat.z_slices().values()]
    else:
        off = flat.v_offsets(); names = ["v1", "v2", "v3", "v4", "v5", "v6", "v7", "v11", "v12", "v13", "v14", "end"]
        blocks = [(off[names[i]], off[names[i + 1]]) for i in range(len(names) - 1)]
    g = max(1.0, float(np.max(np.abs(b))))
    worst = 0.0
    for lo, hi in blocks:
        if hi > lo:
            e = float(np.max(np.abs(a[lo:hi] - b[lo:hi]))) / max(float(np.max(np.abs(b[lo:hi]))), 1e-3 * g)
            worst = e if (e > worst or e != e) else worst
    return worst


def lib_fingerprint(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


def disturbances(lo, hi, steps, d):
    import numpy as np
    # the reference draws rand(1:d) (mpc_simulation.jl:170); here numpy PCG64 seeded with the simulation index
    return np.stack([np.random.default_rng(seed).integers(0, d, size=steps) for seed in range(lo, hi)]).astype(np.int32) \
        if hi > lo else np.zeros((0, steps), dtype=np.int32)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # start the ranks as a CHILD process: this parent has not initialised a GPU and never will
        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        proc = subprocess.run(launch_command(args, argv, free_port()), env=env)
        return proc.returncode

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    on_gpu = args.backend == "nccl"
    if on_gpu and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if on_gpu:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    n_gpus = world
    tdev = torch.device("cuda", local_rank) if on_gpu else None
    dev_index = local_rank if on_gpu else 0

    import spock_jl_amd as S
    from spock_jl_amd import dist as sdist
    from spock_jl_amd import _lib as slib

    lib_kw = {"lib_path": args.lib} if args.lib else {}
    N_REF, NX, NU, D = args.tree_N, 5, 2, 2
    B, M = args.batch, ((args.iters + 5) // 6) * 6
    specs = S.get_server_heat_specs(N_REF, NX, D, nu=NU)
    flat = S.flatten_problem(*specs)
    model = S.build_model(*specs, S.SolverOptions(S.CP, None), batch=B, device=dev_index, max_iter=M, graph_iters=M, **lib_kw)
    inst0 = rank * B
    x0 = 0.1 * (1.0 + 0.001 * (inst0 + np.arange(B)))[:, None] * np.ones((B, NX))

    def step():
        return S.solve_model(model, x0, tol=0.0)     # exactly M iterations of every instance (status MAX_ITER)

    def fence():
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    res = res if isinstance(res, list) else [res]
    assert all(r["iterations"] == M for r in res), "a step must run exactly --iters iterations"
    if world > 1:
        # the path's only collective: the trivial result gather (nu + 5 doubles per instance)
        rec = sdist.gather_records(sdist.instance_record(model, res, NU), [B] * world, device=tdev)
        assert rec.shape == (B * world, NU + 5)
    value = n_gpus * B * M * args.steps / elapsed
    layout_ok = bool(getattr(model.sizes, "interleaved", 0)) or B < 320
    assert layout_ok, f"batch {B}: the element-major regime was not selected (layout_reason {getattr(model.sizes, 'layout_reason', '?')}): the roofline labels below describe its kernels"

    # ---- self-check of the headline: three instances of the state the timed region left behind against the CPU oracle run for the
    #      same iterations (warm-up + timed steps, each a solve of M iterations warm-started from the last).  Outside the timed region.
    parity = None
    if rank == 0 and not args.no_parity:
        from oracle import oracle as O
        total_steps = args.warmup + args.steps
        fresh = total_steps * M > args.parity_max_iters      # too long for the oracle: check one fresh step instead
        if fresh:
            model.reset(); step(); total_steps = 1
        Z, V = model.solver_state.z.reshape(B, -1), model.solver_state.v.reshape(B, -1)
        picks = sorted(set([0, (B // 128) * 64 + min(37, B - 1) if B >= 128 else B // 2, B - 1]))
        worst = 0.0
        t = time.perf_counter()
        for b in picks:
            o = O.Oracle(flat, O.CP)
            for _ in range(total_steps):
                o.solve(x0[b], tol=0.0, max_iter=M)
            e = max(block_relerr(flat, "z", Z[b], o.z), block_relerr(flat, "v", V[b], o.v))
            worst = e if (e > worst or e != e) else worst
        parity = {"instances": picks, "iterations_compared": total_steps * M,
                  "state": "one fresh step after the timed region" if fresh else "the state left by warm-up + timed steps",
                  "measure": "per-block relative error of z and v against the CPU oracle (blocks x|u|s|tau|y, v1..v14)",
                  "max_relerr": worst, "oracle_s": round(time.perf_counter() - t, 2)}
        assert worst <= 1e-8, f"headline state differs from the oracle: {parity}"

    # ---- roofline of the dominant kernel: HIP events around every launch, same workload, same stream
    roofline = None
    fused = bool(model.sizes.fused_iteration)
    if on_gpu:
        bytes_ = algorithmic_bytes(flat)
        prof = model.profile_kernels(36)
        dom = max(("sub_a", "crown", "sub_b", "full"), key=lambda k: prof[k][0] * (prof[k][1] > 0))
        dom_ms = prof[dom][0]
        if fused:   # "sub_b" is then the fused kernel: phase B of iteration k + phase A / backward sweep of k + 1 = one whole iteration
            bytes_["sub_b"] = bytes_["cp_iteration"]
        dom_bytes = B * (bytes_[dom] if dom in bytes_ else bytes_["cp_iteration"])
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        nec = B * necessary_bytes(flat) if (fused and dom == "sub_b") else None
        # measured HBM bytes per launch of that kernel: PMC counters cannot be read from inside this process; they come from the
        # committed rocprofv3 --pmc passes of this same command (profiles/<round>/pmc_b<batch>_summary.json, FETCH_SIZE doubled
        # as MI355X_MICROARCH.md prescribes for gfx950) -- taken only when that pass ran the SAME library build as this run
        traffic = traffic_raw = None
        fp = lib_fingerprint(args.lib or slib.DEFAULT_LIB)
        pdir = os.path.join(ROOT, "profiles")
        pmc_kernels, pmc_iter, pmc_file = {}, None, None
        for rnd in sorted(os.listdir(pdir), reverse=True) if os.path.isdir(pdir) else []:
            f = os.path.join(pdir, rnd, f"pmc_b{B}_summary.json")
            if os.path.exists(f):
                summ = json.load(open(f))
                if summ.get("lib_sha16") == fp:
                    pmc_kernels = summ.get("kernels", {}); pmc_iter = summ.get("iteration"); pmc_file = os.path.relpath(f, ROOT)
                break
        k = pmc_kernels.get("sub_ba" if (fused and dom == "sub_b") else dom)
        if k:
            traffic = k["hbm_bytes_per_launch_corrected"]; traffic_raw = k.get("hbm_bytes_per_launch_raw")
        kname = {"sub_a": "SUB_A", "sub_b": "SUB_BA" if fused else "SUB_B", "crown": "CROWN", "full": "FULL"}[dom]
        ilv = bool(getattr(model.sizes, "interleaved", 0))
        # `achieved` / `frac`: bytes the kernel must move (entry-by-entry count) over its measured duration.
        # SURVEY 8d's model counts the streams of an UNFUSED implementation (6 nz + 5 nv + 2 (n nx + nn nu) per iteration): the
        # fused kernel never moves those bytes, so a rate computed on them is not a bandwidth (it would exceed the HBM peak);
        # it is kept as survey_8d_* for comparison with round 1.
        bytes_used = nec if nec else dom_bytes
        model_txt = "necessary bytes of the fused kernel (reads z, v, zbar's s/tau/y, d; writes zbar, vbar, d)" if nec else "SURVEY 8d"
        kernel_txt = f"spock {'element-major ' if ilv else ''}group kernel, mode {kname}"
        seq = model.profile_sequence(24) if ilv else []
        lba = [(n, lv, ms) for (n, lv, ms) in seq if n in ("lba", "lba2")]
        if lba:   # level-streaming kernels: the dominant kernel is the bottom kernel -- LBA2 (stages N-3 and N-2 in one launch) or LBA
            dom_ms = lba[0][2]
            if lba[0][0] == "lba2":
                bytes_used = B * lba2_necessary_bytes(flat)
                model_txt = "necessary bytes of the two-bottom-stages kernel LBA2 (bench.py: lba2_necessary_bytes)"
                kernel_txt = "spock_ilevel_kernel, mode LBA2 (stages N-3 and N-2 in one launch: forward step + phase B + next phase A + backward step)"
            else:
                bytes_used = B * lba_necessary_bytes(flat)
                model_txt = "necessary bytes of the bottom-stage kernel LBA (bench.py: lba_necessary_bytes)"
                kernel_txt = "spock_ilevel_kernel, mode LBA (bottom stage: forward step + phase B + next phase A + backward step)"
            k = pmc_kernels.get(lba[0][0])
            traffic = k["hbm_bytes_per_launch_corrected"] if k else None
            traffic_raw = k.get("hbm_bytes_per_launch_raw") if k else None
        ach = bytes_used / (dom_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": kernel_txt,
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_raw": traffic_raw,
                    "bytes_model": model_txt,
                    "bytes_per_launch": bytes_used,
                    "frac_measured": round(traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                    "frac_measured_raw": round(traffic_raw / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic_raw else None,
                    "survey_8d_bytes_per_launch": dom_bytes, "survey_8d_rate_GBps": round(achieved, 1),
                    "avg_launch_ms": round(dom_ms, 5), "launches_timed": 24 if lba else prof[dom][1], "lib_sha16": fp,
                    "all_kernels_avg_ms": {k: round(v[0], 5) for k, v in prof.items() if v[1]},
                    "iteration_launches_ms": [[n, lv, round(ms, 5)] for (n, lv, ms) in seq],
                    "whole_iteration_necessary_bytes": B * necessary_bytes(flat),
                    "whole_iteration_frac": round(B * necessary_bytes(flat) * n_gpus / (elapsed / (M * args.steps)) / 1e9 / (HBM_PEAK_GBS * n_gpus), 4)}
        # the whole iteration on the counters: PMC bytes of every launch of one iteration (profiles/<round>/pmc_b<B>_summary.json,
        # written by tools/pmc_summary.py from the committed CSVs) over the timed region's time per iteration
        it_s = elapsed / (M * args.steps)
        tr_it = pmc_iter["hbm_bytes_corrected"] if pmc_iter else None
        roofline.update({"traffic_iteration": tr_it, "traffic_iteration_raw": pmc_iter["hbm_bytes_raw"] if pmc_iter else None,
                         "frac_iteration_measured": round(tr_it / it_s / 1e9 / HBM_PEAK_GBS, 4) if tr_it else None,
                         "waste_ratio": round(tr_it / (B * necessary_bytes(flat)), 4) if tr_it else None,
                         "iteration_ms": round(it_s * 1e3, 5), "pmc_summary": pmc_file,
                         "survey_8d_iteration_bytes": B * bytes_["cp_iteration"],
                         "survey_8d_iteration_rate_GBps": round(B * bytes_["cp_iteration"] / it_s / 1e9, 1)})
    model.close()

    out = {"metric": "Spock (Chambolle-Pock) iterations/sec on the server_heat tree", "value": round(value, 1),
           "unit": "iterations/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"server_heat nx=5 nu=2 d=2, {flat.nl} leaves (reference N={N_REF}), CP iteration incl. termination residuals",
                      "instances_per_gpu": B, "iterations_per_step": M, "nz": flat.nz, "nv": flat.nv,
                      "split_stage": int(model.sizes.split_stage), "algorithm": "CP", "fused_iteration": fused,
                      "parallelism": f"{n_gpus} x {B} independent instances, RCCL all_gather of results only"},
           "roofline": roofline}
    if parity is not None:
        out["parity"] = parity
        out["parity_max_relerr"] = parity["max_relerr"]

    # ---- BASELINE config 4: closed-loop MPC simulations sharded over the ranks (the metric's problems/sec leg)
    if not args.no_mpc:
        mspecs = S.get_server_heat_specs(args.mpc_N, args.mpc_nx, 2)
        mflat = S.flatten_problem(*mspecs)
        steps = args.mpc_steps
        # SURVEY 8d's bytes_SP per instance-iteration of this problem: 2 bytes_T + 26 W (nz + nv)
        m_bytes_T = 8 * (4 * mflat.nz + 3 * mflat.nv + 2 * (mflat.n * mflat.nx + mflat.nn * mflat.nu))
        m_bytes_SP = 2 * m_bytes_T + 26 * 8 * (mflat.nz + mflat.nv)
        legs = {}
        for name, total in (("weak", args.mpc_per_gpu * world), ("strong", args.mpc_total)):
            lo, hi = sdist.shard_range(total, world, rank)
            xa = 0.1 * (1.0 + 0.01 * np.arange(total))[:, None] * np.ones((total, args.mpc_nx))
            wa = disturbances(0, total, steps, 2)
            mm = S.build_model(*mspecs, S.SolverOptions(S.SP, None), batch=max(hi - lo, 1), device=dev_index, **lib_kw) if hi > lo else None
            if mm is not None:   # warm-up run (code upload, graph capture), then back to the freshly built model
                S.mpc_simulation(mm, xa[lo:hi], wa[lo:hi], tol=1e-3); mm.reset()
            fence()
            t = time.perf_counter()
            recs, _, _ = sdist.mpc_simulation_sharded(mspecs, xa, wa, tol=1e-3, model=mm, torch_device=tdev)
            fence()
            dt = max_over_ranks(time.perf_counter() - t)
            if mm is not None:
                mm.close()
            assert recs.shape == (total, args.mpc_nx + 5)
            legs[name] = {"simulations": total, "per_gpu": [sdist.shard_range(total, world, r)[1] - sdist.shard_range(total, world, r)[0] for r in range(world)],
                          "steps_per_simulation": steps, "ms": round(dt * 1e3, 2),
                          "problems_per_s": round(total / dt, 2), "solves_per_s": round(total * steps / dt, 1),
                          "sp_iterations_total": int(recs[:, args.mpc_nx].sum()), "all_converged": bool((recs[:, args.mpc_nx + 1] == 0).all()),
                          "max_final_state": float(recs[:, args.mpc_nx + 2].max()),
                          # whole-leg rate on SURVEY 8d's bytes_SP (every SP iteration of every simulation priced at bytes_SP) against the HBM peak of the GPUs in use
                          "bytes_SP_per_instance_iteration": m_bytes_SP,
                          "hbm_frac_on_bytes_SP": round(int(recs[:, args.mpc_nx].sum()) * m_bytes_SP / dt / 1e9 / (HBM_PEAK_GBS * n_gpus), 4)}
        out["mpc_cfg4"] = dict(legs, workload=f"mpc_simulation.jl: server_heat nx=nu={args.mpc_nx}, d=2, N={args.mpc_N}, SP, tol 1e-3, "
                                              f"{steps} warm-started steps per simulation; one all_gather of (nu+5) doubles per simulation",
                               unit="closed-loop simulations/s",
                               driver="steps chained on the device (SP_CTL_M): a finished instance applies u, advances x0 and starts its next warm-started solve without the host")

    if rank == 0 and world == 1 and on_gpu:
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
            # the same with every host core busy: one independent oracle process per core (instances are independent)
            ncore = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            try:   # a container's CPU share (cgroup v2 cpu.max) is what can actually run at once
                q, per = open("/sys/fs/cgroup/cpu.max").read().split()
                if q != "max":
                    ncore = max(1, min(ncore, int(float(q) / float(per))))
            except Exception:
                pass
            ncore = min(ncore, 64)
            kk = max(20, k // 4)
            code = ("import sys,time,numpy as np;sys.path.insert(0,%r);import spock_jl_amd as S;from oracle import oracle as O;"
                    "f=S.flatten_problem(*S.get_server_heat_specs(%d,5,2,nu=2));o=O.Oracle(f,O.CP);o.solve(np.full(5,.1),tol=0.,max_iter=5);"
                    "t=time.perf_counter();o.solve(np.full(5,.1),tol=0.,max_iter=%d);print(time.perf_counter()-t)" % (ROOT, N_REF, kk))
            t = time.perf_counter()
            procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE) for _ in range(ncore)]
            times = [float(p.communicate()[0].decode().strip().splitlines()[-1]) for p in procs]
            out["cpu_baseline"]["all_cores"] = {"cores": ncore, "value": round(ncore * kk / max(times), 1), "unit": "iterations/s",
                                                "sample": f"{ncore} independent oracle processes x {kk} CP iterations, slowest process timed"}
        if not args.no_single:
            single = {}
            m1 = S.build_model(*specs, S.SolverOptions(S.CP, None), batch=1, device=dev_index, max_iter=2000, graph_iters=100)
            S.solve_model(m1, x0[:1], tol=0.0)
            torch.cuda.synchronize(); t = time.perf_counter(); r = S.solve_model(m1, x0[:1], tol=0.0); dt = time.perf_counter() - t
            single["iterations_per_s"] = round(r["iterations"] / dt, 1)
            single["us_per_iteration_device"] = round(r["solve_ms"] / r["iterations"] * 1e3, 2)
            single["kernel_sequence"] = [n for n, _, _ in m1.profile_sequence(2)]   # (sub_ba_r: the role-lane subtree kernel, spock_rgroup.hpp)
            m1.close()
            specs5 = S.get_server_heat_specs(N_REF, NX, D)           # reference-faithful nu = nx = 5: converges
            flat5 = S.flatten_problem(*specs5)
            m5 = S.build_model(*specs5, S.SolverOptions(S.CP, None), batch=1, device=dev_index)
            S.solve_model(m5, np.full(NX, 0.1), tol=1e-3); m5.reset()
            r5 = S.solve_model(m5, np.full(NX, 0.1), tol=1e-3)
            single["ms_to_tol_1e-3_nu5"] = round(r5["solve_ms"], 3)
            single["iterations_to_tol_nu5"] = r5["iterations"]
            single["kernel_sequence_nu5"] = [n for n, _, _ in m5.profile_sequence(2)]
            m5.close()
            # the Spock algorithm proper (SuperMann-accelerated CP, SolverOptions(SP, nothing)) on the same convergent tree
            ms = S.build_model(*specs5, S.SolverOptions(S.SP, None), batch=1, device=dev_index)
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
            # BASELINE config 5 (random linear nx = 64, nu = 16, N = 15, 16 384 leaves): the S1 sweep levels run on the fp64
            # matrix cores (v_mfma_f64_16x16x4_f64).  Flops per CP iteration from SURVEY.md section 8d.
            if args.tree_N >= 13:
                # BASELINE config 5 (random linear nx = 64, nu = 16, N = 15, 16 384 leaves), one instance, both precisions: flat node-local
                # kernels + sweep levels on the matrix cores (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32).  Two rooflines per
                # precision: flops per CP iteration (SURVEY.md 8d's count, diagonal cost roots) against the dense matrix-core peak, and
                # SURVEY 8d's bytes_CP against the HBM peak.  A single instance is bound by neither: it is a chain of 2 x 14 dependent tree
                # levels and five launch boundaries.
                sp5 = S.get_random_linear_specs(15, 64, 16)
                f5 = S.flatten_problem(*sp5)
                nn5, n5, nl5, ny5, dd = f5.nn, f5.n, f5.nl, f5.ny, f5.d
                l_mac = nn5 * ny5 + (n5 - 1) * (64 + 16) + nl5 * 64                       # diagonal cost roots
                s1_mac = nn5 * (16 * 16 + 2 * 64 * 16 + dd * (3 * 64 * 64 + 3 * 64 * 16))
                s2_mac = nn5 * (ny5 + 2 * dd) ** 2
                flops = 2.0 * (4 * l_mac + s1_mac + s2_mac)
                PEAKS = {"f64": 78.6, "f32": 157.3}   # TFLOP/s dense matrix-core peaks (AMD's public MI355X figures; SURVEY.md 8d quotes the fp64 one)
                out["cfg5"] = {"workload": "random linear nx=64 nu=16 d=2 N=15 (16384 leaves), one instance, CP", "flops_per_iteration": flops}
                xc = 0.1 * np.cos(np.arange(64))
                for dt in ("f64", "f32"):
                    W = 8 if dt == "f64" else 4
                    mc = S.build_model(*sp5, S.SolverOptions(S.CP, None), batch=1, device=dev_index, max_iter=60, graph_iters=60, L_norm="auto", dtype=dt)
                    S.solve_model(mc, xc, tol=0.0)
                    rc = S.solve_model(mc, xc, tol=0.0)
                    seq5 = mc.profile_sequence(12)
                    mc.close()
                    ms_it = rc["solve_ms"] / rc["iterations"]
                    b8d = W * (6 * f5.nz + 5 * f5.nv + 2 * (f5.n * f5.nx + f5.nn * f5.nu))
                    out["cfg5"][dt] = {"ms_per_iteration": round(ms_it, 4), "launches_ms": [[n, round(ms, 5)] for n, _, ms in seq5],
                                       "roofline_mfma": {"bound": "mfma", "achieved": round(flops / (ms_it * 1e-3) / 1e12, 3), "peak": PEAKS[dt], "unit": "TFLOP/s",
                                                         "frac": round(flops / (ms_it * 1e-3) / 1e12 / PEAKS[dt], 4)},
                                       "roofline_hbm": {"bound": "hbm", "achieved": round(b8d / (ms_it * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                        "frac": round(b8d / (ms_it * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "bytes_model": "SURVEY 8d bytes_CP", "bytes_per_iteration": b8d}}
                    # the same with eight instances in flight (the sweeps' one-workgroup-per-subtree kernels then fill the chip four times over,
                    # the flat node-local kernels stream): time per iteration of the batch, HBM fraction on SURVEY 8d's bytes_CP x 8
                    B8 = 8
                    mc = S.build_model(*sp5, S.SolverOptions(S.CP, None), batch=B8, device=dev_index, max_iter=24, graph_iters=24, L_norm="auto", dtype=dt)
                    X8 = (1.0 + 0.01 * np.arange(B8))[:, None] * xc[None, :]
                    S.solve_model(mc, X8, tol=0.0)
                    r8 = S.solve_model(mc, X8, tol=0.0)
                    mc.close()
                    ms8 = r8[0]["solve_ms"] / r8[0]["iterations"]
                    out["cfg5"][dt]["batch8"] = {"instances": B8, "ms_per_batch_iteration": round(ms8, 4), "ms_per_instance_iteration": round(ms8 / B8, 4),
                                                 "roofline_hbm": {"bound": "hbm", "achieved": round(B8 * b8d / (ms8 * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                                  "frac": round(B8 * b8d / (ms8 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "bytes_model": "SURVEY 8d bytes_CP x instances"},
                                                 "roofline_mfma": {"bound": "mfma", "achieved": round(B8 * flops / (ms8 * 1e-3) / 1e12, 3), "peak": PEAKS[dt], "unit": "TFLOP/s",
                                                                   "frac": round(B8 * flops / (ms8 * 1e-3) / 1e12 / PEAKS[dt], 4)}}
                out["cfg5"]["ms_per_iteration"] = out["cfg5"]["f64"]["ms_per_iteration"]
                out["cfg5"]["roofline"] = dict(out["cfg5"]["f64"]["roofline_mfma"], kernel="S1 sweep levels on v_mfma_f64_16x16x4_f64 (spock_skernels.hip), node-local phases as flat kernels",
                                               traffic=None, note="a single instance: a chain of 2 x 14 dependent tree levels and five launch boundaries; see f64 / f32 for both rooflines")
            # problems/s: 64 independent instances of the convergent (nu = 5) tree, different x0, solved to tol 1e-3 as one batch
            probs = {}
            X = 0.1 * (1.0 + 0.01 * np.arange(64))[:, None] * np.ones((64, NX))
            for name, alg in (("cp", S.CP), ("sp", S.SP)):
                mb = S.build_model(*specs5, S.SolverOptions(alg, None), batch=64, device=dev_index)
                S.solve_model(mb, X, tol=1e-3); mb.reset()
                torch.cuda.synchronize(); t = time.perf_counter(); rb = S.solve_model(mb, X, tol=1e-3); torch.cuda.synchronize()
                dtb = time.perf_counter() - t
                assert all(q["status"] == 0 for q in rb)
                probs[name] = {"instances": 64, "ms": round(dtb * 1e3, 2), "problems_per_s": round(64 / dtb, 1),
                               "iterations_max": max(q["iterations"] for q in rb)}
                mb.close()
            # the same with a batch that fills the chip: 2048 instances, element-major layout (finished instances freeze lane by lane)
            if args.tree_N >= 13:
                XB = 0.1 * (1.0 + 0.0005 * np.arange(B))[:, None] * np.ones((B, NX))
                mb = S.build_model(*specs5, S.SolverOptions(S.CP, None), batch=B, device=dev_index)
                S.solve_model(mb, XB, tol=1e-3); mb.reset()
                torch.cuda.synchronize(); t = time.perf_counter(); rb = S.solve_model(mb, XB, tol=1e-3); torch.cuda.synchronize()
                dtb = time.perf_counter() - t
                assert all(q["status"] == 0 for q in rb)
                probs["cp_large_batch"] = {"instances": B, "layout": "element-major" if mb.sizes.interleaved else "instance-major",
                                           "ms": round(dtb * 1e3, 2), "problems_per_s": round(B / dtb, 1),
                                           "iterations_max": max(q["iterations"] for q in rb), "iterations_min": min(q["iterations"] for q in rb)}
                mb.close()
                # the Spock algorithm proper in the same batched regime (instance-major: its traffic is dominated by the Anderson vectors,
                # which are layout-independent streams) with its roofline on SURVEY 8d's bytes_SP = 2 bytes_T + 26 W (nz + nv)
                mb = S.build_model(*specs5, S.SolverOptions(S.SP, None), batch=B, device=dev_index)
                S.solve_model(mb, XB, tol=1e-3); mb.reset()
                torch.cuda.synchronize(); t = time.perf_counter(); rb = S.solve_model(mb, XB, tol=1e-3); torch.cuda.synchronize()
                dtb = time.perf_counter() - t
                assert all(q["status"] == 0 for q in rb)
                it_max = max(q["iterations"] for q in rb)
                W8 = 8
                bytes_T = W8 * (4 * flat5.nz + 3 * flat5.nv + 2 * (flat5.n * flat5.nx + flat5.nn * flat5.nu))
                bytes_SP = 2 * bytes_T + 26 * W8 * (flat5.nz + flat5.nv)
                ms_it = rb[0]["solve_ms"] / it_max
                probs["sp_large_batch"] = {"instances": B, "layout": "element-major" if mb.sizes.interleaved else "instance-major",
                                           "ms": round(dtb * 1e3, 2), "problems_per_s": round(B / dtb, 1), "iterations_max": it_max,
                                           "iterations_min": min(q["iterations"] for q in rb), "ms_per_batch_iteration": round(ms_it, 3),
                                           "roofline": {"bound": "hbm", "bytes_model": "SURVEY 8d bytes_SP per instance-iteration", "bytes_per_instance_iteration": bytes_SP,
                                                        "achieved": round(B * bytes_SP / (ms_it * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                        "frac": round(B * bytes_SP / (ms_it * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                        "note": "instances retire lane by lane over the solve, so the full batch moves fewer bytes than B x bytes_SP in the late iterations: an upper estimate"}}
                mb.close()
            out["problems_to_tol_1e-3_nu5"] = probs

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
