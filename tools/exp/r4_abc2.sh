#!/bin/bash
# is it the sibling rule? new with HJD_SIBLINGS=0 against new and head; plus which sync kernel the bench launches
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end"
for rep in 1 2; do
  unset HJD_LIB; unset HJD_SIBLINGS
  timeout -k 10 300 $B > $O/abd_new.$rep.json 2> $O/abd_new.$rep.err
  HJD_SIBLINGS=0 timeout -k 10 300 $B > $O/abd_new0.$rep.json 2> $O/abd_new0.$rep.err
  HJD_LIB=gpurun_libs/libhjd_c.so timeout -k 10 300 $B > $O/abd_head.$rep.json 2> $O/abd_head.$rep.err
done
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/abd_prof -o abd -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > $O/abd_prof.log 2>&1
python3 - <<P
import json,glob
O="$O"
for tag in ("new","new0","head"):
    r=[json.loads(open(O+"/abd_%s.%d.json"%(tag,k)).read().strip().splitlines()[-1]) for k in (1,2)]
    print(tag, [x["value"] for x in r], r[1]["stage_ms_isolated"])
for f in glob.glob(O+"/abd_prof/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:1500])
P
