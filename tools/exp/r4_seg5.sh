#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
for rep in 1 2 3; do
  echo "default"; timeout -k 10 300 python3 tools/config5_bench.py 2> /dev/null | grep -o '"value.*ms_per_step": [0-9.]*'
  echo "no seg"; HJD_NO_SEG_EMULATION=1 timeout -k 10 300 python3 tools/config5_bench.py 2> /dev/null | grep -o '"value.*ms_per_step": [0-9.]*'
done
