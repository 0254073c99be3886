#!/bin/bash
# round 3: everything that goes under profiles/ (run on the GPU box through gpurun)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R   # NB: remove the build container's gpurun_out/r03_* before merging a new run (tools/collect_round.sh averages what it finds)
python3 bench.py > $O/r03_bench.json 2> $O/r03_bench.err; tail -c 400 $O/r03_bench.json
bash tools/gpu_profile.sh r03 all
cd $R
python3 bench.py --width 1920 --height 1080 --batch 1024 --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/r03_config3_bench.json 2> $O/r03_config3.err; tail -c 300 $O/r03_config3_bench.json
python3 tools/config5_bench.py > $O/r03_config5_bench.json 2> $O/r03_config5.err
python3 tools/config5_bench.py hostile >> $O/r03_config5_bench.json 2>> $O/r03_config5.err
python3 tools/sampling_throughput.py > $O/r03_sampling_throughput.txt 2>&1
python3 tools/single_image_latency.py > $O/r03_small_batch_latency.txt 2>&1
python3 tools/convert_walltime.py > $O/r03_convert_walltime.txt 2>&1
python3 tools/host_prepare_time.py > $O/r03_host_prepare.txt 2>&1
python3 tools/sync_stats.py > $O/r03_sync_stats.txt 2>&1
echo all done
