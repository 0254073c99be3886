#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "segment_parallel or restart or bit_flipped or corpus or statuses" > $O/r4seg6_t.log 2>&1; tail -3 $O/r4seg6_t.log
timeout -k 10 300 python3 tools/config5_bench.py hostile 2> /dev/null | grep -o '"value.*ms_per_step": [0-9.]*'
for rep in 1 2; do
  echo "default"; timeout -k 10 300 python3 tools/config5_bench.py 2> /dev/null | grep -o '"value.*ms_per_step": [0-9.]*'
  echo "no seg"; HJD_NO_SEG_EMULATION=1 timeout -k 10 300 python3 tools/config5_bench.py 2> /dev/null | grep -o '"value.*ms_per_step": [0-9.]*'
done
