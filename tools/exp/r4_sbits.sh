#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
for d in 8 16; do for sb in 2048 4096 8192; do
  echo "distinct $d subseq $sb: $(timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --distinct $d --subseq-bits $sb 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['spread']['further_regions_of_20_steps'], r['stage_ms_overlapped']['huff_sync'], r['stage_ms_isolated']['huff_sync'])")"
done; done
