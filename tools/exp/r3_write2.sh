#!/bin/bash
# experiment (round 3): totals inside k_sync, no gate launch
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/r3h_tests.log 2>&1; tail -15 $O/r3h_tests.log
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end"
run() { tag=$1; shift; timeout -k 10 300 env "$@" $B $EXTRA > $O/r3h_$tag.json 2> $O/r3h_$tag.err; python3 -c "
import json,sys
d=json.loads(open('$O/r3h_$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'], d.get('stage_ms_isolated'), d.get('stage_ms_overlapped'))"; }
EXTRA="--streams 4" run x4
EXTRA="--streams 4" run x4_c2 HJD_SYNC_CONC=2
EXTRA="--streams 4" run x4_pw16 HJD_PHASE_WORDS=16
EXTRA="--streams 4" run x4_S4096 HJD_SUBSEQ_BITS=4096
EXTRA="--streams 4" run x4_S4096_pw16 HJD_SUBSEQ_BITS=4096 HJD_PHASE_WORDS=16
EXTRA="--streams 3" run x3
EXTRA="--streams 4 --batch 128" run x4_b128
echo done
