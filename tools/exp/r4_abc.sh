#!/bin/bash
# interleaved bench of library variants on one box: VARS="tag:path ..." (path empty = in-tree)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end $BENCH_ARGS"
for rep in 1 2 3; do for spec in $VARS; do tag=${spec%%:*}; lib=${spec#*:}
  if [ -n "$lib" ]; then export HJD_LIB=$lib; else unset HJD_LIB; fi
  timeout -k 10 300 $B > $O/abc_$tag.$rep.json 2> $O/abc_$tag.$rep.err
done; done
python3 - <<P
import json,os
O="$O"
for spec in "$VARS".split():
    tag=spec.split(":")[0]
    r=[json.loads(open(O+"/abc_%s.%d.json"%(tag,k)).read().strip().splitlines()[-1]) for k in (1,2,3)]
    print(tag, [x["value"] for x in r], [x["spread"]["further_regions_of_20_steps"] for x in r], r[2]["stage_ms_isolated"])
P
