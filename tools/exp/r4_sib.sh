#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; T=${TAG:-r4sib}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1; tail -3 $O/${T}_tests.log
for sb in 0 3; do echo "HJD_SIBLINGS=$sb"; HJD_SIBLINGS=$sb timeout -k 10 200 python3 tools/single_image_latency.py 2>&1 | grep -v amdgpu | grep -E "auto|2048"; done > $O/${T}_lat.txt
cat $O/${T}_lat.txt
