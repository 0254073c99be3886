#!/bin/bash
# bench at several subsequence sizes (one box): "S value ms sync_iso sync_ovl"
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; T=${TAG:-r4s}
for S in ${SIZES:-0 1024 2048 4096}; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --subseq-bits $S $BENCH_ARGS > $O/${T}_S$S.json 2> $O/${T}_S$S.err
  python3 -c "
import json
d=json.loads(open('$O/${T}_S$S.json').read().strip().splitlines()[-1])
print('S=$S', d['value'], d['spread']['further_regions_of_20_steps'], d['stage_ms_isolated'], d['stage_ms_overlapped'])"
done
