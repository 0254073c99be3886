#!/bin/bash
# experiment (round 3): in-lane warm-up + wave-local checkpointed item walks
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/r3d_tests.log 2>&1; tail -3 $O/r3d_tests.log
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end"
run() { tag=$1; shift; timeout -k 10 300 env "$@" $B $EXTRA > $O/r3d_$tag.json 2> $O/r3d_$tag.err; python3 -c "
import json,sys
d=json.loads(open('$O/r3d_$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'], d.get('stage_ms_isolated'), d.get('stage_ms_overlapped'))"; }
EXTRA="--streams 4" run x4
EXTRA="--streams 4" run x4_w0 HJD_LANE_WARM=0
EXTRA="--streams 4" run x4_w2 HJD_LANE_WARM=2
EXTRA="--streams 4" run x4_w6 HJD_LANE_WARM=6
EXTRA="--streams 4" run x4_w8 HJD_LANE_WARM=8
EXTRA="--streams 4" run x4_S1024 HJD_SUBSEQ_BITS=1024
EXTRA="--streams 4" run x4_S1024_w2 HJD_SUBSEQ_BITS=1024 HJD_LANE_WARM=2
EXTRA="--streams 4" run x4_pw16 HJD_PHASE_WORDS=16
EXTRA="--internal-streams 4" run i4_s1 HJD_STAGGER=1
for w in 0 4; do HJD_LANE_WARM=$w timeout -k 10 200 python3 tools/sync_stats.py > $O/r3d_syncstats_w$w.txt 2>&1; cat $O/r3d_syncstats_w$w.txt; done
echo done
