#!/bin/bash
# experiment (round 3): sweep of subsequence size / window size / front-end limit / IDCT occupancy cap, two runs each
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --streams 4"
run() { tag=$1; shift; for rep in 1 2; do timeout -k 10 300 env "$@" $B > $O/r3w_$tag.$rep.json 2> $O/r3w_$tag.$rep.err; done; python3 -c "
import json,sys
v=[json.loads(open('$O/r3w_$tag.%d.json'%r).read().strip().splitlines()[-1]) for r in (1,2)]
print('$tag', [x['value'] for x in v], v[1].get('stage_ms_overlapped'))"; }
for lib in base w6 w5; do
  L=""; [ $lib != base ] && L="HJD_LIB=$R/gpurun_libs/libhjd_$lib.so"
  for S in 2048 4096; do for pw in 8 16; do for c in 0 2; do
    run ${lib}_S${S}_pw${pw}_c$c $L HJD_SUBSEQ_BITS=$S HJD_PHASE_WORDS=$pw HJD_SYNC_CONC=$c
  done; done; done
done
echo done
