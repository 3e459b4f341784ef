"""bench.py's contract, checked on CODE (no GPU): the N > 1 launch path runs for real on CPUs -- `python bench.py --gpus 2`
with no WORLD_SIZE must start two ranks itself (gloo + the emulation library stand in for RCCL + the HIP library through
bench.py's hidden test hooks) and print ONE JSON line with n_gpus = 2 -- plus the fields of the last committed GPU line."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "libspock_emu.so")

CONTRACT = (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
            ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict))


def _check_contract(out):
    for k, t in CONTRACT:
        assert isinstance(out[k], t), k
    assert "vs_baseline" in out and out["vs_baseline"] is None        # BASELINE.md publishes no number for this metric
    assert out["scaling"] == "weak" and out["dtype"] == "f64" and "workload" in out["config"] and "model" not in out["config"]


def test_launch_command_and_world_size_check():
    sys.path.insert(0, ROOT)
    import bench
    args = bench.parse_args(["--gpus", "4", "--steps", "2"])
    cmd = bench.launch_command(args, ["--gpus", "4", "--steps", "2"], 29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    # a rank whose WORLD_SIZE disagrees with --gpus refuses to run (it used to print n_gpus = 1 silently)
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


@pytest.mark.parametrize("gpus", [1, 2])
def test_bench_runs_n_ranks_and_prints_one_line(gpus):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emu")], stdout=subprocess.DEVNULL)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--backend", "gloo", "--lib", EMU,
           "--tree-N", "6", "--batch", "2", "--iters", "6", "--steps", "1", "--warmup", "0", "--mpc-N", "4", "--mpc-nx", "5",
           "--mpc-steps", "3", "--mpc-per-gpu", "2", "--mpc-total", "5", "--no-cpu-baseline", "--no-single"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    _check_contract(out)
    assert out["n_gpus"] == gpus and out["config"]["instances_per_gpu"] == 2
    # the headline checks itself: instances of the timed state against the oracle, per-block relative error
    assert out["parity"]["iterations_compared"] == 6 and out["parity_max_relerr"] <= 1e-9 and out["parity"]["instances"] == [0, 1]
    mpc = out["mpc_cfg4"]
    assert mpc["weak"]["simulations"] == 2 * gpus and mpc["strong"]["simulations"] == 5
    assert sum(mpc["strong"]["per_gpu"]) == 5 and len(mpc["strong"]["per_gpu"]) == gpus
    assert mpc["weak"]["all_converged"] and mpc["strong"]["all_converged"] and mpc["weak"]["problems_per_s"] > 0


def test_committed_bench_line_has_the_contract_fields():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_b*.json")))
    assert files, "no committed bench line under profiles/"
    out = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    _check_contract(out)
    r = out["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and (r["traffic"] is None or r["traffic"] > 0)
    c = out["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == out["unit"] and c["sample"]
    assert out["value"] > 50 * c["value"]        # north_star: >= 50x the CPU path's iterations/s
    assert r["frac"] >= 0.4                      # north_star: >= 40 % of the HBM roofline
