#!/bin/bash
# After `gpurun -- tools/profiles_round.sh <tag> a|b|c`: turn gpurun_out/<tag>_* into the tracked summaries under profiles/.
# usage: tools/collect_round.sh [tag]     (run in the build container, from the repo root, on the sources that were profiled)
T=${1:-r05}; O=gpurun_out; P=profiles
python3 tools/collect_sq.py $P/sq_counters.json 16 3840 2160 194400 $O/${T}_sq1_b16 $O/${T}_sq2_b16 $O/${T}_sq3_b16 > /dev/null && cp $P/sq_counters.json $P/${T}_sq_counters_b16.json
python3 tools/collect_sq.py $P/${T}_sq_counters_b32.json 32 3840 2160 194400 $O/${T}_sq1_b32 $O/${T}_sq2_b32 > /dev/null
python3 tools/collect_traffic.py $O/${T}_fetch $O/${T}_write 16 3840 2160 $P/traffic.json > /dev/null && cp $P/traffic.json $P/${T}_traffic.json
python3 tools/valu_mix.py $P/valu_mix.json > /dev/null
cp $(ls -t $(find $O/${T}_stats -name "*kernel_stats.csv") | head -1) $P/${T}_kernel_stats.csv   # the newest: gpurun merges every call's files into the same local directory
grep '^{"metric"' $O/${T}_stats.log | tail -1 > $P/${T}_bench_under_rocprof.json
tail -1 $O/${T}_bench.json > $P/${T}_bench.json
for f in config3_bench config3_one_batch config3_four_batches photo_bench bench_batch96; do [ -s $O/${T}_$f.json ] && tail -1 $O/${T}_$f.json > $P/${T}_$f.json; done
[ -s $O/${T}_config5_bench.json ] && grep '^{' $O/${T}_config5_bench.json > $P/${T}_config5_bench.json
for f in sampling_throughput restart_stages small_batch_latency small_images convert_walltime host_prepare sync_stats uniform_frames partly_uniform timeline soak; do
  [ -s $O/${T}_$f.txt ] && grep -v "amdgpu.ids" $O/${T}_$f.txt > $P/${T}_$f.txt
done
echo collected $T
