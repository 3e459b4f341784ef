#!/bin/bash
# Exploration (not a test): the headline under variants of the level-streaming launch geometry.  Run on the GPU box from the repo root.
B="--no-cpu-baseline --no-single --no-mpc --no-parity --steps 10"
run() { echo "== $*"; env "$@" python3 bench.py $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], r['iteration_ms'], [(n,l,round(m,4)) for n,l,m in r['iteration_launches_ms'] if n in ('lba2','lba','crown')])"; }
run X=0
run SPOCK_AMD_ILV_TGROUPS=2
run SPOCK_AMD_ILV_CHUNK=32
run SPOCK_AMD_ILV_CHUNK=8
run SPOCK_AMD_ILV_CHUNK=4
run SPOCK_AMD_ILV_FUSE2=0
