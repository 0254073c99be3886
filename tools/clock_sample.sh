#!/bin/bash
# GPU box: sample the shader clock and power while bench.py runs a long timed region (rocm-smi, 20 samples/s).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
( for i in $(seq 1 400); do rocm-smi --showclocks --showpower --csv 2>/dev/null | tail -n +2 | head -1; sleep 0.05; done > $O/clock_samples.csv ) &
SP=$!
sleep 1
python3 $R/bench.py --steps 3000 --warmup 20 --no-cpu-baseline --no-end-to-end > $O/clock_bench.json 2> $O/clock_bench.err
kill $SP 2>/dev/null
wait $SP 2>/dev/null
rocm-smi --showclocks --showpower --csv 2>/dev/null | head -2
wc -l $O/clock_samples.csv
