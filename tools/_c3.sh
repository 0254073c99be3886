set -e
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r5h_tests.log 2>&1 || { tail -40 gpurun_out/r5h_tests.log; exit 1; }
tail -2 gpurun_out/r5h_tests.log
r() { echo "$1" >> gpurun_out/c3.txt; python3 bench.py $2 --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], list(d['spread'].values())[0])" >> gpurun_out/c3.txt; }
r "config3 one batch, library's split (16 ranges)" "--width 1920 --height 1080 --batch 1024 --steps 5 --warmup 1 --internal-streams -1"
r "config3 one batch, 3 ranges" "--width 1920 --height 1080 --batch 1024 --steps 5 --warmup 1 --internal-streams 3"
r "config3 one batch, 8 ranges" "--width 1920 --height 1080 --batch 1024 --steps 5 --warmup 1 --internal-streams 8"
r "config3 4 caller streams L=4" "--width 1920 --height 1080 --batch 1024 --steps 5 --warmup 1 --launches-per-stream 4"
r "headline one batch, library's split (4 ranges)" "--internal-streams -1"
r "headline one batch, 3 ranges" "--internal-streams 3"
r "headline one batch, 8 ranges" "--internal-streams 8"
r "headline 4 caller streams" ""
cat gpurun_out/c3.txt
for m in "--caller-streams 1 --internal-streams 0" "--caller-streams 4"; do python3 tools/config5_bench.py $m --no-stage-events --steps 10 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5', d['config']['caller_streams'], d['value'], d['spread'])"; done
