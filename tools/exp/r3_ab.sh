#!/bin/bash
# A/B on one box: the in-tree library against gpurun_libs/libhjd_prev.so, interleaved
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end"
for rep in 1 2 3; do
  for v in new prev; do
    if [ $v = prev ]; then export HJD_LIB=gpurun_libs/libhjd_prev.so; else unset HJD_LIB; fi
    timeout -k 10 300 $B > $O/ab_$v.$rep.json 2> $O/ab_$v.$rep.err
  done
done
python3 - <<'P'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT", os.getcwd())+"/gpurun_out"
for v in ("new","prev"):
    r=[json.loads(open(O+"/ab_%s.%d.json"%(v,k)).read().strip().splitlines()[-1]) for k in (1,2,3)]
    print(v, [x["value"] for x in r], [x["spread"]["further_regions_of_20_steps"] for x in r], [x["stage_ms_overlapped"]["huff_sync"] for x in r], [x["stage_ms_isolated"]["huff_sync"] for x in r])
P
