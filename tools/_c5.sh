set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r5f_tests.log 2>&1 || { tail -30 gpurun_out/r5f_tests.log; exit 1; }
tail -2 gpurun_out/r5f_tests.log
run() { echo "$1" >> gpurun_out/c5_q.txt; env $3 python3 tools/config5_bench.py $2 --no-stage-events --steps 10 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['spread'])" >> gpurun_out/c5_q.txt; }
run "caller1 auto" "--caller-streams 1 --internal-streams 0" "A=1"
run "internal2" "--caller-streams 1 --internal-streams 2" "A=1"
run "internal3" "--caller-streams 1 --internal-streams 3" "A=1"
run "internal4" "--caller-streams 1 --internal-streams 4" "A=1"
run "internal4 stagger1" "--caller-streams 1 --internal-streams 4" "HJD_STAGGER=1"
run "internal4 sync_conc0" "--caller-streams 1 --internal-streams 4" "HJD_SYNC_CONC=0"
run "caller2" "--caller-streams 2" "A=1"
run "caller4" "--caller-streams 4" "A=1"
run "internal4 hostile" "--caller-streams 1 --internal-streams 4 --hostile" "A=1"
run "internal4 512" "--caller-streams 1 --internal-streams 4 --images 512" "A=1"
b() { echo "bench $1" >> gpurun_out/c5_q.txt; env $2 python3 bench.py --no-cpu-baseline --no-end-to-end $3 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('spread',{}).get('further_regions_of_20_steps'))" >> gpurun_out/c5_q.txt; }
b default A=1
b "internal 4" A=1 "--internal-streams 4"
cat gpurun_out/c5_q.txt
