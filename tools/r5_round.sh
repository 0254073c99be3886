#!/bin/bash
# one gpurun call: restart-stream throughput, config 5 in its forms, bench.py with other stream / batch shapes (round 5 experiments)
cd "$(dirname "$0")/.."
O=gpurun_out
python tools/sampling_throughput.py > $O/r5_sampling.txt 2>&1
for m in "--caller-streams 4" "--caller-streams 1 --internal-streams 1" "--caller-streams 1 --internal-streams 0" "--caller-streams 1 --internal-streams 4" "--caller-streams 2" "--caller-streams 4 --hostile" "--caller-streams 1 --internal-streams 0 --hostile"; do
  echo "== $m" >> $O/r5_config5.txt
  python tools/config5_bench.py $m >> $O/r5_config5.txt 2>&1
done
for m in "--streams 4" "--streams 5" "--streams 6" "--streams 8" "--streams 4 --batch 96" "--streams 4 --batch 128" "--streams 8 --batch 128"; do
  echo "== $m" >> $O/r5_shapes.txt
  python bench.py --no-cpu-baseline --no-end-to-end $m 2>&1 | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['spread'], j['stage_ms_overlapped'])" >> $O/r5_shapes.txt 2>&1
done
for m in "--streams 4" "--streams 6" "--streams 8"; do
  echo "== GPU_MAX_HW_QUEUES=8 $m" >> $O/r5_shapes.txt
  GPU_MAX_HW_QUEUES=8 python bench.py --no-cpu-baseline --no-end-to-end $m 2>&1 | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['spread'], j['stage_ms_overlapped'])" >> $O/r5_shapes.txt 2>&1
done
tail -n 40 $O/r5_sampling.txt $O/r5_config5.txt $O/r5_shapes.txt
