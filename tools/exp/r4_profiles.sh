#!/bin/bash
# round 4: everything that goes under profiles/ (run on the GPU box through gpurun; then tools/collect_round.sh r04 in the build container)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; T=r04
rm -rf $O/${T}_sq1_b16 $O/${T}_sq2_b16 $O/${T}_sq1_b32 $O/${T}_sq2_b32 $O/${T}_fetch $O/${T}_write $O/${T}_stats
python3 bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; tail -c 300 $O/${T}_bench.json
bash tools/gpu_profile.sh $T all
cd $R
python3 bench.py --content photo --no-cpu-baseline --no-end-to-end > $O/${T}_photo_bench.json 2> $O/${T}_photo.err; tail -c 200 $O/${T}_photo_bench.json
python3 bench.py --width 1920 --height 1080 --batch 1024 --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/${T}_config3_bench.json 2> $O/${T}_config3.err; tail -c 200 $O/${T}_config3_bench.json
python3 tools/config5_bench.py > $O/${T}_config5_bench.json 2> $O/${T}_config5.err
python3 tools/config5_bench.py hostile >> $O/${T}_config5_bench.json 2>> $O/${T}_config5.err
python3 tools/sampling_throughput.py > $O/${T}_sampling_throughput.txt 2>&1
python3 tools/single_image_latency.py > $O/${T}_small_batch_latency.txt 2>&1
python3 tools/convert_walltime.py > $O/${T}_convert_walltime.txt 2>&1
python3 tools/host_prepare_time.py > $O/${T}_host_prepare.txt 2>&1
python3 tools/sync_stats.py > $O/${T}_sync_stats.txt 2>&1
python3 tools/small_images.py > $O/${T}_small_images.txt 2>&1
python3 tools/partly_uniform.py > $O/${T}_partly_uniform.txt 2>&1
python3 tools/uniform_frames.py > $O/${T}_uniform_frames.txt 2>&1
./gpurun_libs/lat_probe > $O/${T}_lat_probe.txt 2>&1
echo all done
