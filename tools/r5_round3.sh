#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
python -m pytest tests -x -q -m gpu > $O/r5e_tests.log 2>&1; tail -3 $O/r5e_tests.log
python tools/ab_bench.py --reps 2 --tag dc base= W9=gpurun_libs/libhjd_W9.so fence=gpurun_libs/libhjd_dcfence.so > $O/ab_dc.log 2>&1; tail -5 $O/ab_dc.log
rm -f $O/r5_config5c.txt
for l in "" gpurun_libs/libhjd_W9.so gpurun_libs/libhjd_dcfence.so; do
  for m in "--caller-streams 1 --internal-streams 0" "--caller-streams 1 --internal-streams 1"; do
  echo "== lib=$l $m" >> $O/r5_config5c.txt
  HJD_LIB=${l:+$PWD/$l} python tools/config5_bench.py $m 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['spread'], j['ms_per_step'], j['stage_ms_isolated'])" >> $O/r5_config5c.txt 2>&1
  done
done
cat $O/r5_config5c.txt
