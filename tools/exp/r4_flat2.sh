#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
for v in "" NOSWEEP NOMOVE NONE c; do echo "variant $v"; if [ -n "$v" ]; then export HJD_LIB=gpurun_libs/libhjd_$v.so; else unset HJD_LIB; fi
  timeout -k 10 200 python3 tools/single_image_latency.py 2>&1 | grep -v amdgpu | grep -E "n= 1 S=auto|n= 4 S=auto"; done
