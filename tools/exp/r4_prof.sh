#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
for S in 1024 2048 4096; do
  HJD_SUBSEQ_BITS=$S timeout -k 10 120 python3 tools/sync_profile.py 16 > $O/r4c_syncprof_S$S.txt 2>&1
done
echo done
