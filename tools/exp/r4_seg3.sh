#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
for c in 16 32 64 128 256; do echo "cut $c"; HJD_SEG_CUT=$c timeout -k 10 300 python3 tools/config5_bench.py hostile 2> /dev/null | grep -o '"value.*'; done
