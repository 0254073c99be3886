set -e
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/r5g_tests.log 2>&1 || { tail -40 gpurun_out/r5g_tests.log; exit 1; }
tail -2 gpurun_out/r5g_tests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
python3 bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err; tail -c 600 gpurun_out/r05_bench.json
