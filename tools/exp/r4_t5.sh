#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; T=${TAG:-r4t5}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1; tail -3 $O/${T}_tests.log
timeout -k 10 200 python3 tools/config5_bench.py > $O/${T}_config5.json 2> $O/${T}_config5.err
python3 -c "
import json
d=json.loads(open('$O/${T}_config5.json').read().strip().splitlines()[-1]); print('config5', d['value'], d['ms_per_step'])"
