#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; T=${1:-x}; cd /tmp; export TMPDIR=/tmp
ONE="--batch 16 --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${T}_fetch -- python3 $R/bench.py $ONE > $O/${T}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${T}_write -- python3 $R/bench.py $ONE > $O/${T}_write.log 2>&1
echo traffic $T done
