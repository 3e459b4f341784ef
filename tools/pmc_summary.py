"""Condense the rocprofv3 outputs of tools/profile_round.sh into profiles/<round>/:
    bench_b<B>_kernel_stats.csv      copy of the --stats kernel summary
    pmc_fetch/write_counter_collection.csv   the raw per-dispatch counter rows of the CP kernels (first 400 rows each)
    pmc_b<B>_summary.json            per kernel: mean FETCH_SIZE / WRITE_SIZE (KB) per launch and HBM bytes per launch,
                                     raw and with FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950

    python tools/pmc_summary.py r01 256
"""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, B = sys.argv[1], int(sys.argv[2])
src = os.path.join(ROOT, "gpurun_out", f"prof_{rnd}")
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
MODES = {"1": "sub_a", "2": "crown", "3": "sub_b", "5": "sub_ba", "0": "full"}


IMODES = {"1": "sub_a", "2": "crown_top", "5": "sub_ba", "9": "crown_a", "10": "crown_up", "11": "crown_down", "4": "finalize"}


LMODES = {"12": "lb", "13": "lba", "14": "la", "15": "la0"}   # level-streaming kernels (spock_ilevel.hpp)


def kind(name):
    if "spock_ilevel_kernel" in name:      # spock_ilevel_kernel<StaticDims<...>, MODE, WX>: LB / LA of all streamed stages pooled
        import re
        m = re.search(r">,\s*(\d+),\s*(true|false)>", name)
        return LMODES.get(m.group(1)) if m else None
    if "spock_igroup_kernel" in name:      # element-major regime: spock_igroup_kernel<StaticDims<...>, MODE, WX>
        import re
        m = re.search(r">,\s*(\d+),\s*(true|false)>", name)
        return IMODES.get(m.group(1)) if m else None
    if "spock_group_kernel" not in name:
        return None
    m = name.rsplit(",", 1)[-1].strip().split(">")[0].strip()
    return MODES.get(m)


newest = lambda pat: max(glob.glob(pat, recursive=True), key=os.path.getmtime)   # gpurun_out/ accumulates earlier runs
shutil.copy(newest(os.path.join(src, "stats", "**", "*kernel_stats.csv")), os.path.join(dst, f"bench_b{B}_kernel_stats.csv"))
import hashlib
_h = hashlib.sha256(open(os.path.join(ROOT, "spock.jl_amd", "libspock_amd.so"), "rb").read()).hexdigest()[:16]
out = {"config": f"SPOCK_AMD_ILV_TGROUPS=1 bench.py --batch {B} (server_heat nx=5 nu=2 d=2 N=13), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes",
       "kernels": {}, "batch": B, "lib_sha16": _h}
for ctr, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    f = newest(os.path.join(src, sub, "**", "*counter_collection.csv"))
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if kind(r["Kernel_Name"])]
    with open(os.path.join(dst, f"{sub}_counter_collection.csv"), "w", newline="") as g:
        w = csv.DictWriter(g, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(keep[:400])
    acc = {}
    for r in keep:
        if r["Counter_Name"] != ctr:
            continue
        acc.setdefault(kind(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        e = out["kernels"].setdefault(k, {})
        e[f"{ctr}_KB_mean"] = sum(v) / len(v); e[f"{ctr}_launches"] = len(v)
for k, e in out["kernels"].items():
    f, w = e.get("FETCH_SIZE_KB_mean", 0.0), e.get("WRITE_SIZE_KB_mean", 0.0)
    e["hbm_bytes_per_launch_raw"] = int((f + w) * 1024)
    e["hbm_bytes_per_launch_corrected"] = int((2 * f + w) * 1024)
json.dump(out, open(os.path.join(dst, f"pmc_b{B}_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
