"""The bench line committed with the round's profiles carries every field of the bench.py contract (no GPU needed: it
checks the JSON that the last GPU run of `python bench.py` produced, profiles/<round>/bench_b*.json)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_b*.json")))
    assert files, "no committed bench line under profiles/"
    out = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(out[k], t), k
    assert "vs_baseline" in out and out["vs_baseline"] is None        # BASELINE.md publishes no number for this metric
    assert out["scaling"] == "weak" and out["dtype"] == "f64" and "workload" in out["config"] and "model" not in out["config"]
    r = out["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and (r["traffic"] is None or r["traffic"] > 0)
    c = out["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == out["unit"] and c["sample"]
    assert out["value"] > 50 * c["value"]        # north_star: >= 50x the CPU path's iterations/s
    assert r["frac"] >= 0.4                      # north_star: >= 40 % of the HBM roofline
