import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def S():
    import spock_jl_amd
    return spock_jl_amd


@pytest.fixture(scope="session")
def O():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def kat_cp():
    return np.loadtxt(os.path.join(GOLDEN, "residuals_cp.txt"))


@pytest.fixture(scope="session")
def kat_sp():
    return np.loadtxt(os.path.join(GOLDEN, "residuals_sp.txt"))


# ---- error measure -------------------------------------------------------------------------------------------------------
# Iterates are compared BLOCK BY BLOCK (z = x | u | s | tau | y, implicit_l.jl:106-158; v = v1 | .. | v7 | v11 | .. | v14,
# cpock.jl:51-60): the error of a block is taken relative to the magnitude of THAT block, so a wrong entry in a small-magnitude
# block (tau, the risk multipliers, ...) is seen at full size instead of being measured against the largest entry of the whole
# vector.  A block's scale is floored at 1e-3 of the vector's scale (itself floored at 1, as before) so that blocks that are
# zero up to rounding do not turn eps-sized differences into O(1) "relative" errors.
# Vectors are recognised by their length: every FlatProblem registers its nz / nv block partition when it is created.
_LAYOUTS = {}
BLOCK_FLOOR = 1e-3


def register_layout(flat):
    zs = flat.z_slices()
    off = flat.v_offsets()
    names = ["v1", "v2", "v3", "v4", "v5", "v6", "v7", "v11", "v12", "v13", "v14", "end"]
    parts = {flat.nz: [(k, (sl.start, sl.stop)) for k, sl in zs.items()],
             flat.nv: [(names[i], (off[names[i]], off[names[i + 1]])) for i in range(len(names) - 1)]}
    for n, blocks in parts.items():
        if n in _LAYOUTS and _LAYOUTS[n] != blocks:
            _LAYOUTS[n] = None          # two problems with the same length and different partitions: fall back to the whole vector
        elif n not in _LAYOUTS:
            _LAYOUTS[n] = blocks


def _install_layout_hook():
    from spock_jl_amd import problem
    if getattr(problem.FlatProblem, "_layout_hook", False):
        return
    orig = problem.FlatProblem.__post_init__

    def post_init(self):
        orig(self)
        register_layout(self)
    problem.FlatProblem.__post_init__ = post_init
    problem.FlatProblem._layout_hook = True


_install_layout_hook()


def block_errors(a, b):
    """{block: max|a - b| / max(max|b| over the block, BLOCK_FLOOR * max(1, max|b|))} for a z- or v-shaped array (last axis)."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    g = max(1.0, float(np.max(np.abs(b)))) if b.size else 1.0
    blocks = _LAYOUTS.get(a.shape[-1]) if a.ndim >= 1 and a.shape == b.shape else None
    if not blocks:
        return {"all": float(np.max(np.abs(a - b)) / g) if a.size else 0.0}
    out = {}
    for name, (lo, hi) in blocks:
        if hi <= lo:
            continue
        bb = b[..., lo:hi]
        diff = np.abs(a[..., lo:hi] - bb)
        m = float(np.max(diff)) if not np.isnan(diff).any() else float("nan")
        out[name] = m / max(float(np.max(np.abs(bb))), BLOCK_FLOOR * g)
    return out


def relerr(a, b, blocks=True):
    """Largest per-block relative error (see above); blocks=False: the whole-vector measure max|a-b| / max(1, max|b|)."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    if not blocks:
        return float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b)))))
    e = block_errors(a, b)
    vals = list(e.values())
    return float("nan") if any(v != v for v in vals) else max(vals)
