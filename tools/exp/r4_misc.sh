#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; T=${TAG:-r4m}
timeout -k 10 200 python3 tools/single_image_latency.py > $O/${T}_latency.txt 2>&1
timeout -k 10 200 python3 tools/small_images.py > $O/${T}_small.txt 2>&1
timeout -k 10 300 python3 tools/partly_uniform.py > $O/${T}_partly.txt 2>&1
timeout -k 10 200 python3 tools/config5_bench.py > $O/${T}_config5.json 2> $O/${T}_config5.err
echo done
