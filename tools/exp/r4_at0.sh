#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
for a in 2 3 5; do echo "HJD_SIBLINGS=$a"; HJD_SIBLINGS=$a timeout -k 10 200 python3 tools/single_image_latency.py 2>&1 | grep -E "n= 1 S=auto|n= 1 S=4096|n= 2 S=auto|n= 4 S=auto"; done
