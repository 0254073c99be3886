#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r3n_tests.log 2>&1; tail -5 $O/r3n_tests.log
timeout -k 10 300 python3 tools/sampling_throughput.py > $O/r3n_sampling.txt 2>&1; cat $O/r3n_sampling.txt
timeout -k 10 300 python3 tools/config5_bench.py > $O/r3n_config5.json 2> $O/r3n_config5.err; tail -c 900 $O/r3n_config5.json; tail -3 $O/r3n_config5.err
timeout -k 10 300 python3 tools/convert_walltime.py > $O/r3n_convert.txt 2>&1; cat $O/r3n_convert.txt
timeout -k 10 300 python3 tools/host_prepare_time.py > $O/r3n_prepare.txt 2>&1; cat $O/r3n_prepare.txt
timeout -k 10 300 python3 tools/single_image_latency.py > $O/r3n_latency.txt 2>&1; head -12 $O/r3n_latency.txt
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end > $O/r3n_bench.json 2>$O/r3n_bench.err; python3 -c "
import json; d=json.loads(open('$O/r3n_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['stage_ms_isolated'])"
echo done
