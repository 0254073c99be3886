set -e
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/kw_tests.log 2>&1 || { tail -40 gpurun_out/kw_tests.log; exit 1; }
tail -2 gpurun_out/kw_tests.log
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-end-to-end > gpurun_out/kw_bench.json
python3 -c "
import json; d=json.loads(open('gpurun_out/kw_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['spread']['further_regions_of_20_steps'], d['stage_ms_isolated'])"
