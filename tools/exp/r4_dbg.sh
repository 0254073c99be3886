#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
timeout -k 10 200 python3 tools/exp/dbg_resync.py > $O/dbg_resync.log 2>&1
HJD_LIB=gpurun_libs/libhjd_norun.so timeout -k 10 200 python3 tools/exp/dbg_resync.py > $O/dbg_resync_norun.log 2>&1
timeout -k 10 120 python3 tools/sync_profile.py 16 > $O/r4b_syncprof.txt 2>&1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end > $O/r4b_bench.json 2> $O/r4b_bench.err
tail -c 600 $O/r4b_bench.json
