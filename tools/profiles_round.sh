#!/bin/bash
# The round's evidence, collected ON THE GPU BOX (through gpurun, from the repo root):  tools/profiles_round.sh <tag> <part>
#   part a: the default bench (with CPU baseline and end-to-end), the PMC passes and the kernel-trace stats of tools/gpu_profile.sh,
#           a third PMC pass with the fp32 instruction counters (dynamic instruction mix), the kernel timeline
#   part b: the other BASELINE configurations and the side measurements (config 3, config 5 in four forms, photographic content,
#           samplings, restart streams, small batches, small images, host side, file-to-file wall time, uniform / flat content)
#   part c: a soak run (random images against the oracle)
# Writes gpurun_out/<tag>_*; summarise in the build container with tools/collect_round.sh <tag> (on the same sources).
TAG=${1:-r05}; PART=${2:-a}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
J='import sys,json; print(sys.stdin.read().strip().splitlines()[-1])'
if [ "$PART" = a ]; then
  python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; tail -c 400 $O/${TAG}_bench.json; echo
  bash tools/gpu_profile.sh $TAG all
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $O/${TAG}_sq3_b16 -- python3 $R/bench.py --batch 16 --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/${TAG}_sq3_b16.log 2>&1)
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_tl -- python3 $R/bench.py --steps 20 --no-cpu-baseline --no-end-to-end --repeats 0 > $O/${TAG}_tl.log 2>&1)
  python3 tools/timeline.py $O/${TAG}_tl > $O/${TAG}_timeline.txt; rm -rf $O/${TAG}_tl; cat $O/${TAG}_timeline.txt
fi
if [ "$PART" = b ]; then
  # config 3: sixteen batches of 64 on four caller streams; four batches of 256 (each split by the library); ONE batch of 1024 on one stream
  python3 bench.py --width 1920 --height 1080 --batch 1024 --steps 5 --warmup 1 --launches-per-stream 4 --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "$J" > $O/${TAG}_config3_bench.json
  python3 bench.py --width 1920 --height 1080 --batch 1024 --steps 5 --warmup 1 --internal-streams -1 --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "$J" > $O/${TAG}_config3_one_batch.json
  python3 bench.py --width 1920 --height 1080 --batch 1024 --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "$J" > $O/${TAG}_config3_four_batches.json
  rm -f $O/${TAG}_config5_bench.json
  for m in "--caller-streams 4" "--caller-streams 4 --hostile" "--caller-streams 1 --internal-streams 0" "--caller-streams 1 --internal-streams 0 --hostile" "--caller-streams 2" "--caller-streams 1 --internal-streams 4" "--caller-streams 4 --images 512"; do
    python3 tools/config5_bench.py $m 2>/dev/null | python3 -c "$J" >> $O/${TAG}_config5_bench.json
  done
  python3 bench.py --content photo --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "$J" > $O/${TAG}_photo_bench.json
  python3 bench.py --batch 96 --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "$J" > $O/${TAG}_bench_batch96.json
  python3 tools/sampling_throughput.py > $O/${TAG}_sampling_throughput.txt 2>&1
  python3 tools/dri_probe.py > $O/${TAG}_restart_stages.txt 2>&1
  python3 tools/single_image_latency.py > $O/${TAG}_small_batch_latency.txt 2>&1
  python3 tools/small_images.py > $O/${TAG}_small_images.txt 2>&1
  python3 tools/host_prepare_time.py > $O/${TAG}_host_prepare.txt 2>&1
  python3 tools/convert_walltime.py > $O/${TAG}_convert_walltime.txt 2>&1
  python3 tools/sync_stats.py > $O/${TAG}_sync_stats.txt 2>&1
  python3 tools/uniform_frames.py > $O/${TAG}_uniform_frames.txt 2>&1
  python3 tools/partly_uniform.py > $O/${TAG}_partly_uniform.txt 2>&1
  tail -n 3 $O/${TAG}_config5_bench.json | cut -c1-300
fi
if [ "$PART" = c ]; then
  timeout -k 10 ${SOAK_S:-700} python3 tests/soak_gpu.py --seconds ${SOAK_RUN:-600} --seed ${SOAK_SEED:-20261005} --max-side 2400 > $O/${TAG}_soak.txt 2>&1; tail -2 $O/${TAG}_soak.txt
fi
echo profiles $TAG $PART done
