#!/bin/bash
# experiment (round 3): checkpointed, window-compacted synchronisation
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/r3c_tests.log 2>&1; tail -3 $O/r3c_tests.log
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end"
run() { tag=$1; shift; timeout -k 10 300 env "$@" $B $EXTRA > $O/r3c_$tag.json 2> $O/r3c_$tag.err; python3 -c "
import json,sys
d=json.loads(open('$O/r3c_$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'], d.get('stage_ms_isolated'), d.get('stage_ms_overlapped'))"; }
EXTRA="--streams 4" run x4 HJD_STAGGER=0
EXTRA="--internal-streams 4" run i4_s0 HJD_STAGGER=0
EXTRA="--internal-streams 4" run i4_s1 HJD_STAGGER=1
EXTRA="--streams 4" run x4_S1024 HJD_SUBSEQ_BITS=1024
EXTRA="--streams 4" run x4_pw4 HJD_PHASE_WORDS=4
EXTRA="--streams 4" run x4_pw16 HJD_PHASE_WORDS=16
timeout -k 10 200 python3 tools/sync_stats.py > $O/r3c_syncstats.txt 2>&1; cat $O/r3c_syncstats.txt
timeout -k 10 200 python3 tools/single_image_latency.py > $O/r3c_latency.txt 2>&1; tail -12 $O/r3c_latency.txt
echo done
