B="python bench.py --no-cpu-baseline --no-end-to-end"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['spread']['further_regions_of_20_steps'], d['ms_per_step'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
}
for rep in 1 2; do
  for v in "T0S0:0:0" "Tdev:0x40000000:0" "Tnof:0x20000000:0" "Snof:0:0x20000000" "Sdev:0:0x40000000" "TSdev:0x40000000:0x40000000"; do
    n=${v%%:*}; r=${v#*:}; t=${r%%:*}; s=${r#*:}
    if [ "$n" = T0S0 ]; then $B 2>/dev/null | tail -1 > gpurun_out/c4_${n}_$rep.json; else HJD_EVF_T=$t HJD_EVF_S=$s $B 2>/dev/null | tail -1 > gpurun_out/c4_${n}_$rep.json; fi
    show gpurun_out/c4_${n}_$rep.json
  done
done
$B --batch 96 2>/dev/null | tail -1 > gpurun_out/c4_b96.json; show gpurun_out/c4_b96.json
$B --batch 128 2>/dev/null | tail -1 > gpurun_out/c4_b128.json; show gpurun_out/c4_b128.json
$B --batch 128 --launches-per-stream 2 2>/dev/null | tail -1 > gpurun_out/c4_b128l2.json; show gpurun_out/c4_b128l2.json
$B --batch 192 --launches-per-stream 2 2>/dev/null | tail -1 > gpurun_out/c4_b192l2.json; show gpurun_out/c4_b192l2.json
for S in 512 1024; do HJD_SUBSEQ_BITS=$S python tools/partly_uniform.py > gpurun_out/c4_pu_S$S.txt 2>&1; done
echo done
