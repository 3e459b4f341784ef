"""Exploration script (not a test): ms per CP iteration of the element-major regime against the split stage (first streamed stage)
and the number of tile groups.   python tools/explore_split.py [lib] [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import spock_jl_amd as S
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else None
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
specs = S.get_server_heat_specs(13, 5, 2, nu=2)
x0 = np.full((B, 5), 0.1) * (1 + 0.001 * np.arange(B))[:, None]
for split in (3, 4, 5, 6, 8):
    for tg in (1, 2, 4):
        os.environ["SPOCK_AMD_ILV_TGROUPS"] = str(tg)
        m = S.build_model(*specs, batch=B, max_iter=96, lib_path=lib, split_stage=split)
        S.solve_model(m, x0, tol=0.0)
        r = S.solve_model(m, x0, tol=0.0)
        line = f"split {split} tgroups {tg}: {r[0]['solve_ms'] / 96:.3f} ms/iter  {B * 96 / r[0]['solve_ms'] * 1e3:9.0f} it/s"
        if tg == 1:
            seq = m.profile_sequence(12)
            line += "  | " + " ".join(f"{n}{lv}={ms*1e3:.0f}" for n, lv, ms in seq)
        print(line, flush=True)
        m.close()
