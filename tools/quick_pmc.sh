#!/bin/bash
# Quick PMC look at a python command's kernels: mean FETCH_SIZE / WRITE_SIZE (KB) per kernel name, two separate passes
# (the two TCC counters do not fit one pass).  usage: tools/quick_pmc.sh <tag> <python script> [args...]
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/qpmc_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -- python3 "$@" > $OUT/$C.log 2>&1 || { tail -5 $OUT/$C.log; exit 2; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, re, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = max(glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True))
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k][c] = (sum(v) / len(v), len(v))
for k, e in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", (0, 0))[0]):
    name = re.sub(r"spock::", "", k)[:110]
    f = e.get("FETCH_SIZE", (0, 0)); w = e.get("WRITE_SIZE", (0, 0))
    print(f"FETCH {f[0]/1e6:9.3f} GB(KB/1e6) x{f[1]:4d}  WRITE {w[0]/1e6:9.3f} x{w[1]:4d}  raw F+W {(f[0]+w[0])*1024/1e9:8.3f} GB  2F+W {(2*f[0]+w[0])*1024/1e9:8.3f} GB  {name}")
PY
