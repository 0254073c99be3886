#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "segment_parallel or restart" > $O/r4seg_t1.log 2>&1; tail -15 $O/r4seg_t1.log
timeout -k 10 600 python3 tools/config5_bench.py hostile > $O/r4seg_hostile.json 2> $O/r4seg_hostile.err; cut -c1-200 $O/r4seg_hostile.json; grep -o '"value.*' $O/r4seg_hostile.json; tail -3 $O/r4seg_hostile.err
