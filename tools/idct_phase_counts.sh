#!/bin/bash
# GPU box: VALU instructions of k_idct_color up to the end of phase 1 / phase 2 / whole kernel.
# Build the variants first (here, no GPU needed):
#   for c in 1 2; do touch include/hjd.h; make -s lib EXTRA=-DHJD_IDCT_CUT=$c && cp hls-jpeg-decoder_amd/lib/libhjd.so hls-jpeg-decoder_amd/lib/libhjd_cut$c.so; done; touch include/hjd.h; make -s lib
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for V in cut1 cut2 full; do
  if [ $V = full ]; then export HJD_LIB=$R/hls-jpeg-decoder_amd/lib/libhjd.so; else export HJD_LIB=$R/hls-jpeg-decoder_amd/lib/libhjd_$V.so; fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/phase_$V -- python3 $R/tools/idct_phase_counts.py > $O/phase_$V.log 2>&1
done
python3 - <<'P'
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT", os.getcwd())+"/gpurun_out"
for v in ("cut1","cut2","full"):
    acc=collections.defaultdict(list)
    for f in glob.glob(O+"/phase_%s/**/*counter_collection.csv"%v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_idct_color" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(v, {k: sum(x)/len(x) for k,x in acc.items()})
P
