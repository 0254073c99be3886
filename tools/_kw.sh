set -e
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/kw_tests.log 2>&1 || { tail -40 gpurun_out/kw_tests.log; exit 1; }
tail -2 gpurun_out/kw_tests.log
for pw in 16 8; do
echo "PHASE_WORDS=$pw" >> gpurun_out/kw_lat.txt
HJD_PHASE_WORDS=$pw timeout -k 10 300 python3 tools/single_image_latency.py 2>&1 | grep "S=auto\|S=2048\|S=1024" >> gpurun_out/kw_lat.txt
done
cat gpurun_out/kw_lat.txt
timeout -k 10 300 python3 tools/sampling_throughput.py > gpurun_out/kw_sampling.txt 2>&1; cat gpurun_out/kw_sampling.txt
timeout -k 10 200 python3 tools/config5_bench.py --caller-streams 4 --no-stage-events --steps 10 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 caller4', d['value'], d['spread'])"
timeout -k 10 200 python3 tools/config5_bench.py --caller-streams 1 --internal-streams 0 --steps 10 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 caller1', d['value'], d['spread'], d['stage_ms_isolated'])"
timeout -k 10 200 python3 tools/config5_bench.py --caller-streams 4 --hostile --no-stage-events --steps 10 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 hostile caller4', d['value'], d['spread'])"
