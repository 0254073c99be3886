#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 --kernel-trace of bench.py: for the timed steps, the share of wall time during which each kernel
kind is running, and the share during which NO issue-bound kernel (k_idct_color, k_write) is running.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 20 --no-cpu-baseline --no-end-to-end --repeats 0
    python3 tools/timeline.py gpurun_out/tl"""
import collections
import csv
import glob
import re
import sys


def main():
    d = sys.argv[1]
    rows = []
    qmap = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"<.*", "", r["Kernel_Name"].split("(")[0].replace("void ", "")).strip()
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Stream_Id", r.get("Queue_Id", "0"))))
            qmap.setdefault(r.get("Stream_Id", "?"), collections.Counter())[(r.get("Queue_Id", "?"), k)] += 1
    rows.sort()
    if not rows:
        print("no kernel trace found")
        return
    # the timed region: the last 40 % of the k_idct_color launches' span (warm-up and isolated replays lie outside)
    idct = [r for r in rows if r[2] == "k_idct_color"]
    n = len(idct)
    lo, hi = idct[int(n * 0.25)][0], idct[int(n * 0.6)][1]
    sel = [r for r in rows if r[0] >= lo and r[1] <= hi]
    span = hi - lo
    ev = []
    for s, e, k, q in sel:
        ev.append((s, 1, k))
        ev.append((e, -1, k))
    ev.sort()
    active = collections.Counter()
    t_prev = lo
    busy = collections.Counter()
    combos = collections.Counter()
    for t, dlt, k in ev:
        dt = t - t_prev
        if dt > 0:
            kinds = tuple(sorted(x for x in active if active[x] > 0))
            for x in kinds:
                busy[x] += dt
            combos[kinds] += dt
        active[k] += dlt
        t_prev = t
    print("window %.3f ms, %d launches" % (span / 1e6, len(sel)))
    for st in sorted(qmap):  # HIP stream -> hardware queue(s), with the kernels launched there
        byq = collections.defaultdict(list)
        for (q, k), n in qmap[st].items():
            byq[q].append("%s x%d" % (k, n))
        for q in sorted(byq):
            print("  stream %s -> queue %s: %s" % (st, q, ", ".join(sorted(byq[q])[:6])))
    if len(sys.argv) > 2:  # the first N launches of the window, one per line: start and duration in us, kernel, queue
        for s0, e0, k, q in sel[:int(sys.argv[2])]:
            print("    %9.1f  %8.1f  %-18s q%s" % ((s0 - lo) / 1e3, (e0 - s0) / 1e3, k, q))
    for k, v in busy.most_common():
        dur = [e - s for s, e, kk, q in sel if kk == k]
        print("  %-18s running %5.1f %% of the time; %4d launches, mean %.4f ms" % (k, 100.0 * v / span, len(dur), sum(dur) / len(dur) / 1e6))
    heavy = ("k_idct_color", "k_write")
    none = sum(v for c, v in combos.items() if not any(h in c for h in heavy))
    print("  no k_idct_color / k_write running: %.1f %% of the time" % (100.0 * none / span))
    print("  most common combinations:")
    for c, v in combos.most_common(12):
        print("    %5.1f %%  %s" % (100.0 * v / span, " + ".join(c) if c else "(idle)"))


if __name__ == "__main__":
    main()
