#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: PMC counter passes over bench.py for ONE launch of
# 16 and of 32 images (one stream, nothing beside the kernels), plus a kernel-trace --stats pass of the default
# bench command.  Writes under gpurun_out/<tag>_*; summarise with tools/collect_sq.py / collect_traffic.py.
# usage: tools/gpu_profile.sh <tag> [sq|traffic|stats|all]
set -e
TAG=${1:-r02}
WHAT=${2:-all}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SQ2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
ONE="--streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end"
if [ "$WHAT" = sq ] || [ "$WHAT" = all ]; then
  for B in ${BATCHES:-16 32}; do
    rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/${TAG}_sq1_b$B -- python3 $R/bench.py --batch $B $ONE > $O/${TAG}_sq1_b$B.log 2>&1
    rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/${TAG}_sq2_b$B -- python3 $R/bench.py --batch $B $ONE > $O/${TAG}_sq2_b$B.log 2>&1
  done
fi
if [ "$WHAT" = traffic ] || [ "$WHAT" = all ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -- python3 $R/bench.py --batch 16 $ONE > $O/${TAG}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -- python3 $R/bench.py --batch 16 $ONE > $O/${TAG}_write.log 2>&1
fi
if [ "$WHAT" = stats ] || [ "$WHAT" = all ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end > $O/${TAG}_stats.log 2>&1
fi
echo profile $TAG $WHAT done
