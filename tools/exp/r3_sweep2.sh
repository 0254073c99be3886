#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "subsequence_sizes or golden_rgb or dense_records" > $O/r3x_tests.log 2>&1; tail -3 $O/r3x_tests.log
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --streams 4"
run() { tag=$1; shift; for rep in 1 2; do timeout -k 10 300 env "$@" $B > $O/r3x_$tag.$rep.json 2> $O/r3x_$tag.$rep.err; done; python3 -c "
import json,sys
v=[json.loads(open('$O/r3x_$tag.%d.json'%r).read().strip().splitlines()[-1]) for r in (1,2)]
print('$tag', [x['value'] for x in v], v[1].get('stage_ms_isolated'), v[1].get('stage_ms_overlapped'))"; }
for S in 4096 8192 16384; do for pw in 16 32; do for c in 0 2; do
    run S${S}_pw${pw}_c$c HJD_SUBSEQ_BITS=$S HJD_PHASE_WORDS=$pw HJD_SYNC_CONC=$c
done; done; done
echo done
