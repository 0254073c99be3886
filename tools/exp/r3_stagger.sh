#!/bin/bash
# experiment (round 3): ranges of one 64-image batch as a staggered software pipeline
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end"
run() { tag=$1; shift; env "$@" $B $EXTRA > $O/r3s_$tag.json 2> $O/r3s_$tag.err; python3 -c "
import json,sys
d=json.loads(open('$O/r3s_$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'])"; }
EXTRA="--internal-streams 4" run i4_s0 HJD_STAGGER=0
EXTRA="--internal-streams 4" run i4_s1 HJD_STAGGER=1
EXTRA="--internal-streams 4" run i4_s2 HJD_STAGGER=2
EXTRA="--internal-streams 3" run i3_s1 HJD_STAGGER=1
EXTRA="--internal-streams 2" run i2_s1 HJD_STAGGER=1
EXTRA="--internal-streams 8" run i8_s1 HJD_STAGGER=1
EXTRA="--internal-streams 4" run i4_s1_S1024 HJD_STAGGER=1 HJD_SUBSEQ_BITS=1024
EXTRA="--internal-streams 4 --batch 128" run i4_s1_b128 HJD_STAGGER=1
EXTRA="--streams 4" run x4 HJD_STAGGER=0
python3 tools/sync_stats.py > $O/r3s_syncstats.txt 2>&1; cat $O/r3s_syncstats.txt
cd /tmp && export TMPDIR=/tmp && HJD_STAGGER=1 rocprofv3 --kernel-trace --output-format csv -d $O/r3s_trace -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --internal-streams 4 > $O/r3s_trace.log 2>&1
echo done
