#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r4seg_tests.log 2>&1; tail -5 $O/r4seg_tests.log
cd /tmp; export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/r4seg_prof -o hostile -- python3 $R/tools/config5_bench.py hostile > $O/r4seg_prof.log 2>&1
python3 - <<P
import glob
for f in glob.glob("$O/r4seg_prof/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:3000])
P
