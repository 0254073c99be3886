#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
rm -f $O/r5_conc.txt
for e in "X=0" "HJD_IDCT_CONC=1" "HJD_IDCT_CONC=2" "HJD_IDCT_CONC=1 HJD_SYNC_CONC=0" "HJD_IDCT_CONC=1 HJD_WRITE_CONC=1" "HJD_IDCT_CONC=1 HJD_WRITE_CONC=1 HJD_SYNC_CONC=0" "HJD_WRITE_CONC=1" "HJD_IDCT_CONC=1 HJD_SYNC_CONC=1" "HJD_IDCT_CONC=1 HJD_SYNC_CONC=3"; do
  for m in "--streams 4" ; do
  echo "== $e $m" >> $O/r5_conc.txt
  env $e python bench.py --no-cpu-baseline --no-end-to-end --no-stage-events $m 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['spread'])" >> $O/r5_conc.txt 2>&1
  done
done
cat $O/r5_conc.txt
