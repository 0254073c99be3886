#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
timeout -k 10 400 python3 -m pytest tests -m gpu -x -q -k "sibling or uniform or forced or knobs or golden or full_size or subsequence" > $O/r4flat_t.log 2>&1; tail -3 $O/r4flat_t.log
timeout -k 10 300 python3 tools/partly_uniform.py 2>&1 | grep -v amdgpu | cut -c1-150 > $O/r4flat_pu.txt; cat $O/r4flat_pu.txt
timeout -k 10 300 python3 tools/uniform_frames.py 2>&1 | grep -v amdgpu | cut -c1-150 > $O/r4flat_uf.txt; cat $O/r4flat_uf.txt
for sb in 5 11 rule; do echo "HJD_SIBLINGS=$sb"; if [ $sb = rule ]; then unset HJD_SIBLINGS; else export HJD_SIBLINGS=$sb; fi
  timeout -k 10 200 python3 tools/single_image_latency.py 2>&1 | grep -v amdgpu | grep -E "auto"; done
