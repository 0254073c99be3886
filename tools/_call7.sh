B="python bench.py --no-cpu-baseline --no-end-to-end"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], list(d['spread'].values())[0], d['ms_per_step'], d.get('preheat',{}).get('steps'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
}
for rep in 1 2; do
$B --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/c7_pre_$rep.json; show gpurun_out/c7_pre_$rep.json
$B --steps 20 --warmup 5 --preheat-ms 0 2>/dev/null | tail -1 > gpurun_out/c7_nopre_$rep.json; show gpurun_out/c7_nopre_$rep.json
$B --steps 20 --warmup 5 --batch 128 --launches-per-stream 2 2>/dev/null | tail -1 > gpurun_out/c7_pre_b128l2_$rep.json; show gpurun_out/c7_pre_b128l2_$rep.json
$B --steps 20 --warmup 5 --batch 128 --launches-per-stream 2 --distinct 16 2>/dev/null | tail -1 > gpurun_out/c7_pre_b128l2d16_$rep.json; show gpurun_out/c7_pre_b128l2d16_$rep.json
done
bash tools/profiles_round.sh r05 a > gpurun_out/c7_parta.log 2>&1; tail -3 gpurun_out/c7_parta.log
echo done
