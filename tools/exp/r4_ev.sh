#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
for rep in 1 2; do for v in ev noev; do
  A=""; [ $v = noev ] && A="--no-stage-events"
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end $A > $O/r4ev_$v.$rep.json 2> $O/r4ev_$v.$rep.err
  python3 -c "
import json
d=json.loads(open('$O/r4ev_$v.$rep.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['spread']['further_regions_of_20_steps'])"
done; done
