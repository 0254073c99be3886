#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
rm -f $O/r5_config5b.txt
for e in "HJD_SYNC_CONC=2" "HJD_SYNC_CONC=0" "HJD_SYNC_CONC=3" "HJD_SYNC_CONC=4"; do
 for m in "--caller-streams 4 --no-stage-events" "--caller-streams 3 --no-stage-events" "--caller-streams 2 --no-stage-events" "--caller-streams 1 --internal-streams 0" "--caller-streams 1 --internal-streams 3" "--caller-streams 1 --internal-streams 4"; do
  echo "== $e $m" >> $O/r5_config5b.txt
  env $e python tools/config5_bench.py $m 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['spread'], j['ms_per_step'])" >> $O/r5_config5b.txt 2>&1
 done
done
cat $O/r5_config5b.txt
