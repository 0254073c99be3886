#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for st in 2 3 4 5 6 8; do
  echo "streams $st: $(timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --streams $st 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['spread']['further_regions_of_20_steps'])")"
done
for b in 96 128; do
  echo "batch $b streams 4: $(timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --batch $b 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['spread']['further_regions_of_20_steps'])")"
done
for sc in 1 2 3; do
  echo "sync_conc $sc: $(HJD_SYNC_CONC=$sc timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['spread']['further_regions_of_20_steps'])")"
done
