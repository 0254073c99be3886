#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
run() { tag=$1; shift; args=$1; shift; for rep in 1 2; do timeout -k 10 300 env "$@" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end $args > $O/r3y_$tag.$rep.json 2> $O/r3y_$tag.$rep.err; done; python3 -c "
import json,sys
v=[json.loads(open('$O/r3y_$tag.%d.json'%r).read().strip().splitlines()[-1]) for r in (1,2)]
print('$tag', [x['value'] for x in v], v[1].get('stage_ms_overlapped'))"; }
for st in 3 4 6 8; do for c in 2 3; do run s${st}_c$c "--streams $st" HJD_SYNC_CONC=$c; done; done
run s4_b128_c2 "--streams 4 --batch 128" HJD_SYNC_CONC=2
run s8_b128_c3 "--streams 8 --batch 128" HJD_SYNC_CONC=3
run s8_b128_c4 "--streams 8 --batch 128" HJD_SYNC_CONC=4
run s4_b32_c2 "--streams 4 --batch 32" HJD_SYNC_CONC=2
echo done
