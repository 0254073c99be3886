#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "segment_parallel or restart or bit_flipped or corpus" > $O/r4seg4_t.log 2>&1; tail -3 $O/r4seg4_t.log
timeout -k 10 300 python3 tools/config5_bench.py hostile 2> /dev/null | grep -o '"value.*'
cd /tmp; export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4seg4_prof -o hostile -- python3 $R/tools/config5_bench.py hostile > $O/r4seg4_prof.log 2>&1
python3 - <<P
import glob
for f in glob.glob("$O/r4seg4_prof/**/*kernel_stats.csv", recursive=True):
    print("\n".join(l[:150] for l in open(f).read().splitlines()[:14]))
P
