#!/bin/bash
# scheduling sweep on one box: caller streams x front-end concurrency limit
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; T=${TAG:-r4sch}
for ST in ${STREAMS:-3 4 6 8}; do for CC in ${CONCS:-0 2 3}; do
  HJD_SYNC_CONC=$CC timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-end-to-end --streams $ST > $O/${T}_s${ST}_c$CC.json 2> $O/${T}_s${ST}_c$CC.err
  python3 -c "
import json
d=json.loads(open('$O/${T}_s${ST}_c$CC.json').read().strip().splitlines()[-1])
print('streams=$ST conc=$CC', d['value'], d['spread']['further_regions_of_20_steps'], {k: round(v,3) for k,v in d['stage_ms_overlapped'].items()})"
done; done
