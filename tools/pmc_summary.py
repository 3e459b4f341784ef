"""Condense the rocprofv3 outputs of tools/profile_round.sh into profiles/<round>/:
    bench_b<B>_kernel_stats.csv      copy of the --stats kernel summary
    pmc_fetch/write_counter_collection.csv   the raw per-dispatch counter rows of the CP kernels (first 400 rows each)
    pmc_b<B>_summary.json            per kernel: mean FETCH_SIZE / WRITE_SIZE (KB) per launch and HBM bytes per launch,
                                     raw and with FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950

    pmc_b<B>_summary.json also carries "iteration": the launches of ONE CP iteration (every kernel kind that runs once or more per
    iteration, its launches per iteration = its launch count / the bottom kernel's), their PMC bytes summed -- what bench.py's
    roofline.traffic_iteration / frac_iteration_measured / waste_ratio are computed from -- and "per_kind_table", one row per
    kernel kind: launches per iteration, mean duration (from the kernel_stats.csv), bytes per launch, achieved TB/s.

    python tools/pmc_summary.py r03 2048                 condense gpurun_out/prof_r03 into profiles/r03
    python tools/pmc_summary.py r03 2048 --recompute     recompute the summary from the CSVs already committed under profiles/r03
"""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, B = sys.argv[1], int(sys.argv[2])
RECOMPUTE = "--recompute" in sys.argv[3:]
src = os.path.join(ROOT, "gpurun_out", f"prof_{rnd}")
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
MODES = {"1": "sub_a", "2": "crown", "3": "sub_b", "5": "sub_ba", "0": "full"}


IMODES = {"1": "sub_a", "2": "crown_top", "5": "sub_ba", "9": "crown_a", "10": "crown_up", "11": "crown_down", "4": "finalize"}


LMODES = {"12": "lb", "13": "lba", "14": "la", "15": "la0", "20": "lba2"}   # level-streaming kernels (spock_ilevel.hpp)


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
stats_csv = os.path.join(dst, f"bench_b{B}_kernel_stats.csv")
if not RECOMPUTE:
    shutil.copy(newest(os.path.join(src, "stats", "**", "*kernel_stats.csv")), stats_csv)
import hashlib
_h = hashlib.sha256(open(os.path.join(ROOT, "spock.jl_amd", "libspock_amd.so"), "rb").read()).hexdigest()[:16]
if RECOMPUTE and os.path.exists(os.path.join(dst, f"pmc_b{B}_summary.json")):
    _h = json.load(open(os.path.join(dst, f"pmc_b{B}_summary.json"))).get("lib_sha16", _h)   # the build the committed passes ran
out = {"config": f"SPOCK_AMD_ILV_TGROUPS=1 bench.py --batch {B} (server_heat nx=5 nu=2 d=2 N=13), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes",
       "kernels": {}, "batch": B, "lib_sha16": _h}
for ctr, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    f = os.path.join(dst, f"{sub}_counter_collection.csv") if RECOMPUTE else newest(os.path.join(src, sub, "**", "*counter_collection.csv"))
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if kind(r["Kernel_Name"])]
    if not RECOMPUTE:   # every dispatch of the CP kernels is kept: the summary must be recomputable from the committed file
        with open(os.path.join(dst, f"{sub}_counter_collection.csv"), "w", newline="") as g:
            w = csv.DictWriter(g, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(keep)
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
# ---- one iteration: every kind that is launched at least once per iteration of the bottom kernel (lba2 / lba, or sub_ba / sub_b / full)
ks = out["kernels"]
bottom = next((k for k in ("lba2", "lba", "sub_ba", "sub_b", "full") if k in ks), None)
dur = {}   # mean duration per kind from the --stats pass (ns -> ms), pooled over the kernel names of a kind
if os.path.exists(stats_csv):
    tot, cnt = {}, {}
    for r in csv.DictReader(open(stats_csv)):
        k = kind(r["Name"])
        if k:
            tot[k] = tot.get(k, 0.0) + float(r["TotalDurationNs"]); cnt[k] = cnt.get(k, 0) + int(r["Calls"])
    dur = {k: tot[k] / cnt[k] / 1e6 for k in tot}
if bottom:
    nb = ks[bottom]["FETCH_SIZE_launches"]
    per_it, table = {}, []
    raw = corr = 0.0
    for k, e in ks.items():
        lp = e.get("FETCH_SIZE_launches", 0) / nb
        if lp < 0.99:          # prologue / finalize kernels: not part of an iteration
            continue
        lp = round(lp)
        per_it[k] = lp
        raw += lp * e["hbm_bytes_per_launch_raw"]; corr += lp * e["hbm_bytes_per_launch_corrected"]
        ms = dur.get(k)
        table.append({"kind": k, "launches_per_iteration": lp, "avg_ms": round(ms, 5) if ms else None,
                      "hbm_bytes_per_launch_corrected": e["hbm_bytes_per_launch_corrected"],
                      "achieved_TBps": round(e["hbm_bytes_per_launch_corrected"] / (ms * 1e-3) / 1e12, 3) if ms else None})
    out["iteration"] = {"launches": per_it, "hbm_bytes_raw": int(raw), "hbm_bytes_corrected": int(corr),
                        "ms_sum_of_launches": round(sum(t["launches_per_iteration"] * t["avg_ms"] for t in table if t["avg_ms"]), 5)}
    out["per_kind_table"] = sorted(table, key=lambda t: -(t["avg_ms"] or 0) * t["launches_per_iteration"])
json.dump(out, open(os.path.join(dst, f"pmc_b{B}_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

# ---- BASELINE config 5 (one instance, tools/prof_flat64.py 1 cfg5) and config 4's 64 closed loops: kernel stats + counters --------
def _condense_aux():
    import re
    def short(name):
        name = re.sub(r"spock::|void ", "", name)
        m = re.match(r"(spock_\w+)<(.*)>\(", name)
        return f"{m.group(1)}<{m.group(2)}>" if m else name.split("(")[0]
    for tag in ("cfg5", "mpc64"):
        st = os.path.join(dst, f"{tag}_kernel_stats.csv")
        if not RECOMPUTE:
            cands = glob.glob(os.path.join(src, f"{tag}_stats", "**", "*kernel_stats.csv"), recursive=True)
            if not cands:
                continue
            shutil.copy(max(cands, key=os.path.getmtime), st)
        if not os.path.exists(st):
            continue
        kern = {}
        for r in csv.DictReader(open(st)):
            if "spock" in r["Name"]:
                kern[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ms": round(float(r["AverageNs"]) / 1e6, 5), "pct": float(r["Percentage"])}
        summ = {"workload": "BASELINE config 5: random linear nx=64 nu=16 N=15, one instance, 12 CP iterations" if tag == "cfg5"
                else "BASELINE config 4: 64 closed-loop simulations x 20 steps, SP, tol 1e-3 (steps chained on the device)", "kernels": kern, "lib_sha16": _h}
        for ctr, sub in (("FETCH_SIZE", f"{tag}_pmc_fetch"), ("WRITE_SIZE", f"{tag}_pmc_write")):
            f = os.path.join(dst, f"{sub}_counter_collection.csv")
            if not RECOMPUTE:
                cands = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
                if not cands:
                    continue
                rows = [r for r in csv.DictReader(open(max(cands, key=os.path.getmtime))) if "spock" in r["Kernel_Name"]]
                with open(f, "w", newline="") as g:
                    w = csv.DictWriter(g, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
            if not os.path.exists(f):
                continue
            acc = {}
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == ctr:
                    acc.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
            for k, v in acc.items():
                kern.setdefault(k, {})[f"{ctr}_KB_mean"] = sum(v) / len(v)
        tot_ms = tot_b = 0.0
        for k, e in kern.items():
            if "FETCH_SIZE_KB_mean" in e and "WRITE_SIZE_KB_mean" in e:
                e["hbm_bytes_per_launch_corrected"] = int((2 * e["FETCH_SIZE_KB_mean"] + e["WRITE_SIZE_KB_mean"]) * 1024)   # FETCH_SIZE x 2 on gfx950
                e["achieved_TBps"] = round(e["hbm_bytes_per_launch_corrected"] / (e["avg_ms"] * 1e-3) / 1e12, 3) if e.get("avg_ms") else None
                if e.get("calls", 0) >= 12:      # the kernels of an iteration (the prologue's run once)
                    tot_ms += e["avg_ms"]; tot_b += e["hbm_bytes_per_launch_corrected"]
        if tot_ms:
            summ["iteration"] = {"ms_sum_of_launches": round(tot_ms, 5), "hbm_bytes_corrected": int(tot_b), "achieved_TBps": round(tot_b / (tot_ms * 1e-3) / 1e12, 3)}
        json.dump(summ, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
        print(json.dumps(summ, indent=1))


_condense_aux()
