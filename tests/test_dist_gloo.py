"""N > 1 path on CPUs: world_size-2 gloo run of the instance sharding + result gather (spock.jl_amd/dist.py).
The per-rank compute uses the emulation harness (test infrastructure) because this machine has no GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["SPOCK_ROOT"])
import torch.distributed as dist
import spock_jl_amd as S
from spock_jl_amd import dist as sd
dist.init_process_group("gloo")
specs = S.get_server_heat_specs(5, 5, 2)
x0 = 0.1 * (1.0 + 0.1 * np.arange(5))[:, None] * np.ones((5, 5))
rec = sd.solve_sharded(specs, x0, tol=1e-3, lib_path=os.environ["SPOCK_EMU"], split_stage=2)
# closed-loop simulations sharded the same way (BASELINE config 4's split, mpc_simulation.jl:38-48)
mspecs = S.get_server_heat_specs(4, 5, 2)
mx0 = 0.1 * (1.0 + 0.1 * np.arange(5))[:, None] * np.ones((5, 5))
mw = np.stack([np.random.default_rng(seed).integers(0, 2, size=4) for seed in range(5)]).astype(np.int32)
mrec, xt, ut = sd.mpc_simulation_sharded(mspecs, mx0, mw, tol=1e-3, options=S.SolverOptions(S.CP, None), lib_path=os.environ["SPOCK_EMU"])
lo, hi = sd.shard_range(5, dist.get_world_size(), dist.get_rank())
assert xt.shape == (hi - lo, 5, 5) and ut.shape == (hi - lo, 4, 5)
if dist.get_rank() == 0:
    np.save(os.environ["SPOCK_OUT"], rec)
    np.save(os.environ["SPOCK_OUT"] + ".mpc.npy", mrec)
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_ranges():
    sys.path.insert(0, ROOT)
    import spock_jl_amd  # noqa: F401
    from spock_jl_amd.dist import shard_range
    for n, w in ((64, 8), (5, 2), (3, 4), (7, 3)):
        blocks = [shard_range(n, w, r) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(blocks[r][1] == blocks[r + 1][0] for r in range(w - 1))
    assert [shard_range(64, 8, r) for r in (0, 7)] == [(0, 8), (56, 64)]


def test_two_rank_gloo_matches_single_process(tmp_path, S, O):
    emu = os.path.join(ROOT, "tests", "emu", "libspock_emu.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emu")], stdout=subprocess.DEVNULL)
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = str(tmp_path / "rec.npy")
    env = dict(os.environ, SPOCK_ROOT=ROOT, SPOCK_EMU=emu, SPOCK_OUT=out, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)], env=env, timeout=300)
    rec = np.load(out)
    assert rec.shape == (5, 5 + 5)
    specs = S.get_server_heat_specs(5, 5, 2)
    flat = S.flatten_problem(*specs)
    x0 = 0.1 * (1.0 + 0.1 * np.arange(5))[:, None] * np.ones((5, 5))
    for i in range(5):
        o = O.Oracle(flat, O.CP)
        r = o.solve(x0[i], tol=1e-3)
        u0 = o.z[flat.z_slices()["u"]][:5]
        assert int(rec[i, 5]) == r["iterations"] and int(rec[i, 6]) == r["status"]
        assert np.max(np.abs(rec[i, :5] - u0)) <= 1e-9
    # the sharded closed loops: every simulation equals the same loop driven through the oracle, whichever rank ran it
    mrec = np.load(out + ".mpc.npy")
    assert mrec.shape == (5, 5 + 5)
    mspecs = S.get_server_heat_specs(4, 5, 2)
    mflat = S.flatten_problem(*mspecs)
    A, B = mspecs[2].A, mspecs[2].B
    mw = np.stack([np.random.default_rng(seed).integers(0, 2, size=4) for seed in range(5)])
    for i in range(5):
        o = O.Oracle(mflat, O.CP)
        x = x0[i].copy()
        its = 0
        for t in range(4):
            r = o.solve(x, tol=1e-3)
            u = o.z[mflat.z_slices()["u"]][:5].copy()
            if t == 0:
                assert np.max(np.abs(mrec[i, :5] - u)) <= 1e-9
            its += r["iterations"]
            x = A[mw[i, t]] @ x + B[mw[i, t]] @ u
        assert int(mrec[i, 5]) == its and int(mrec[i, 6]) == 0 and int(mrec[i, 8]) == 4
        assert abs(mrec[i, 7] - np.max(np.abs(x))) <= 1e-9
