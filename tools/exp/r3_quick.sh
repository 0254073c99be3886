#!/bin/bash
# quick check: GPU tests + default bench twice (+ optional env variants given as arguments "TAG:ENV=.. ENV=..")
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; T=${TAG:-r3q}
[ -z "$NOTESTS" ] && { timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1; tail -4 $O/${T}_tests.log; }
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end $BENCH_ARGS"
run() { tag=$1; shift; for rep in 1 2; do timeout -k 10 300 env "$@" $B > $O/${T}_$tag.$rep.json 2> $O/${T}_$tag.$rep.err; done; python3 -c "
import json,sys
v=[json.loads(open('$O/${T}_$tag.%d.json'%r).read().strip().splitlines()[-1]) for r in (1,2)]
print('$tag', [x['value'] for x in v], v[1].get('stage_ms_isolated'), v[1].get('stage_ms_overlapped'))"; }
run default
for spec in "$@"; do tag=${spec%%:*}; envs=${spec#*:}; run $tag $envs; done
echo done
