#!/bin/bash
# Profiles of the bench workload for profiles/<round>/ (run on the GPU box through gpurun, from the repo root):
#   1. rocprofv3 --kernel-trace --stats        -> per-kernel average durations (must agree with bench.py's HIP events)
#   2. rocprofv3 --pmc FETCH_SIZE --kernel-trace  and  --pmc WRITE_SIZE --kernel-trace, separate passes
#      (MI355X_MICROARCH.md: the two TCC counters do not fit one pass; never combined with other trace domains)
#   3. the same three passes for BASELINE config 5 (one instance, 12 CP iterations: tools/prof_flat64.py 1 cfg5) and a --stats pass of
#      config 4's 64 closed-loop simulations (tools/prof_mpc64.py)
# Results land under gpurun_out/prof_<round>/; tools/pmc_summary.py condenses them into profiles/<round>/.
set -o pipefail
R=${1:-r03}
# the element-major regime splits a launch into tile groups on parallel graph branches; for per-kernel figures that line up with
# bench.py's HIP-event durations (whole batch per launch) the profiled runs use ONE tile group
export SPOCK_AMD_ILV_TGROUPS=1
OUT=gpurun_out/prof_$R
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
B="--no-cpu-baseline --no-single --no-mpc --no-parity"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 5 --warmup 1 $B > $OUT/bench_stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --iters 24 $B > $OUT/pmc_fetch.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --iters 24 $B > $OUT/pmc_write.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg5_stats -- python3 tools/prof_flat64.py 1 cfg5 > $OUT/cfg5_stats.log 2>&1 || exit 4
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cfg5_pmc_fetch -- python3 tools/prof_flat64.py 1 cfg5 > $OUT/cfg5_pmc_fetch.log 2>&1 || exit 5
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/cfg5_pmc_write -- python3 tools/prof_flat64.py 1 cfg5 > $OUT/cfg5_pmc_write.log 2>&1 || exit 6
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mpc64_stats -- python3 tools/prof_mpc64.py 64 > $OUT/mpc64_stats.log 2>&1 || exit 7
tail -1 $OUT/bench_stats.log | cut -c1-300
find $OUT -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | xargs ls -la
