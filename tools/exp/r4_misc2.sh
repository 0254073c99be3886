#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "reference_main or thumbnail or sibling" > $O/r4m2_tests.log 2>&1; tail -2 $O/r4m2_tests.log
timeout -k 10 300 python3 bench.py --content photo --no-cpu-baseline --no-end-to-end > $O/r4m2_photo.json 2> $O/r4m2_photo.err
python3 -c "
import json
d=json.loads(open('$O/r4m2_photo.json').read().strip().splitlines()[-1]); print('photo', d['value'], d['spread']['further_regions_of_20_steps'], d['stage_ms_isolated'])"
STREAMS="4" CONCS="1 2 3 4" TAG=r4m2s bash tools/exp/r4_sched.sh
