#!/bin/bash
# experiment (round 3): front-end concurrency limit, no lane warm-up, block table a block ahead
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/r3f_tests.log 2>&1; tail -3 $O/r3f_tests.log
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end"
run() { tag=$1; shift; timeout -k 10 300 env "$@" $B $EXTRA > $O/r3f_$tag.json 2> $O/r3f_$tag.err; python3 -c "
import json,sys
d=json.loads(open('$O/r3f_$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'], d.get('stage_ms_isolated'), d.get('stage_ms_overlapped'))"; }
EXTRA="--streams 4" run x4
EXTRA="--streams 4" run x4_c1 HJD_SYNC_CONC=1
EXTRA="--streams 4" run x4_c2 HJD_SYNC_CONC=2
EXTRA="--streams 4" run x4_c3 HJD_SYNC_CONC=3
EXTRA="--streams 4 --batch 128" run x4_b128_c2 HJD_SYNC_CONC=2
EXTRA="--streams 4 --batch 128" run x4_b128
EXTRA="--streams 4" run x4_c2_S1024 HJD_SYNC_CONC=2 HJD_SUBSEQ_BITS=1024
EXTRA="--streams 4" run x4_c2_pw16 HJD_SYNC_CONC=2 HJD_PHASE_WORDS=16
timeout -k 10 200 python3 tools/sync_stats.py > $O/r3f_syncstats.txt 2>&1; cat $O/r3f_syncstats.txt
timeout -k 10 200 python3 tools/sync_profile.py 16 > $O/r3f_prof.txt 2>&1; tail -32 $O/r3f_prof.txt
cd /tmp && export TMPDIR=/tmp && HJD_SYNC_CONC=2 rocprofv3 --kernel-trace --output-format csv -d $O/r3f_trace -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end > $O/r3f_trace.log 2>&1
echo done
