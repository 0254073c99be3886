#!/bin/bash
# build_variant.sh NAME "FLAGS" — an experiment build of libhjd.so with extra -D switches into gpurun_libs/libhjd_NAME.so
# (git-ignored; travels to the GPU box; tools/ab_bench.py NAME=gpurun_libs/libhjd_NAME.so runs it beside the in-tree build)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_libs
P=hls-jpeg-decoder_amd
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-result $2 -shared -x hip \
  $P/csrc/hjd_unstuff.hip $P/csrc/hjd_entropy.hip $P/csrc/hjd_pixels.hip $P/csrc/hjd_seq.hip $P/host/hjd_api.cpp $P/host/hjd_host.cpp $P/host/openjpg_compat.cpp \
  -o gpurun_libs/libhjd_$1.so
