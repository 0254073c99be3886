#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/r5f_tests.log 2>&1; tail -3 $O/r5f_tests.log
python tools/sampling_throughput.py 2>/dev/null | tail -6 > $O/r5_sampling2.txt; cat $O/r5_sampling2.txt
rm -f $O/r5_config5d.txt
for m in "--caller-streams 1 --internal-streams 0" "--caller-streams 1 --internal-streams 0 --hostile" "--caller-streams 1 --internal-streams 1"; do
  echo "== $m" >> $O/r5_config5d.txt
  python tools/config5_bench.py $m 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['spread'], j['ms_per_step'], j['stage_ms_isolated'])" >> $O/r5_config5d.txt 2>&1
done
cat $O/r5_config5d.txt
timeout -k 10 400 python tests/soak_gpu.py --seconds 150 --seed 5005 > $O/r5_soak1.txt 2>&1; tail -3 $O/r5_soak1.txt
