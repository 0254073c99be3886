#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
bash tools/exp/r3_ab.sh
cd /tmp; export TMPDIR=/tmp
ONE="--batch 16 --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end"
rm -rf $O/r4w16
HJD_LIB=$R/gpurun_libs/libhjd_prev.so rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/r4w16 -- python3 $R/bench.py $ONE > $O/r4w16.log 2>&1
python3 - <<'P'
import csv,glob,os,collections
O=os.environ.get("GRAFT_REPO_ROOT", os.getcwd())+"/gpurun_out"
acc=collections.defaultdict(list)
for f in glob.glob(O+"/r4w16/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]=="WRITE_SIZE": acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print("RUN16 WRITE_SIZE MB/image", k, sum(v)/len(v)*1024/16/1e6)
P
