#!/bin/bash
# sibling rule (128 workgroups in all) against none, batches of 1..16 4K images; then the GPU suite
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
for sb in 0 rule; do echo "HJD_SIBLINGS=$sb"; if [ $sb = rule ]; then unset HJD_SIBLINGS; else export HJD_SIBLINGS=$sb; fi
  timeout -k 10 200 python3 tools/single_image_latency.py 2>&1 | grep -v amdgpu | grep -E "auto"; done > $O/r4sib2_lat.txt
unset HJD_SIBLINGS
cat $O/r4sib2_lat.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r4sib2_tests.log 2>&1; tail -3 $O/r4sib2_tests.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > $O/r4sib2_bench.json 2> $O/r4sib2_bench.err; cut -c1-300 $O/r4sib2_bench.json
