#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
python3 bench.py > $O/r04_bench.json 2> $O/r04_bench.err; tail -c 300 $O/r04_bench.json
rm -rf $O/r04_stats; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end > $O/r04_stats.log 2>&1
tail -c 300 $O/r04_stats.log
