#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py into profiles/traffic.json.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, following
/opt/skills/guides/MI355X_MICROARCH.md §HBM: the counters are in KiB, and on gfx950 FETCH_SIZE reports
half the bytes of a wide coalesced streaming read (the kernels here read with 16-byte-per-lane loads);
WRITE_SIZE is exact for 16-byte-per-lane stores and uncalibrated for the 12-byte-per-lane pixel stores
of k_idct_color, so `write_calibrated` says false for that kernel.

usage: collect_traffic.py <fetch_dir> <write_dir> <images_per_launch> <width> <height> <out.json>
"""
import collections
import csv
import glob
import json
import os
import re
import sys

STAGE_OF = {"k_idct_color": "idct_color", "k_sync": "huff_sync", "k_repair_totals": "huff_repair", "k_write": "huff_write",
            "k_unstuff_write": "unstuff", "k_unstuff_count": "unstuff_count", "k_dcapply": "dc_scan", "k_dcsum": "dc_sum", "k_scan": "scan"}


def per_kernel(d, counter):
    tot = collections.defaultdict(float)
    n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = re.sub(r"<.*", "", r["Kernel_Name"].split("(")[0].replace("void ", "")).strip()
            tot[k] += float(r["Counter_Value"])
            n[k] += 1
    return {k: tot[k] / n[k] for k in tot}, n


def main():
    fdir, wdir, ipl, w, h, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    fetch, nf = per_kernel(fdir, "FETCH_SIZE")
    write, nw = per_kernel(wdir, "WRITE_SIZE")
    res = {}
    for k, stage in STAGE_OF.items():
        if k not in fetch or k not in write:
            continue
        res[stage] = {
            "kernel": k, "images_per_launch": ipl, "width": w, "height": h,
            "fetch_size_kib_raw": round(fetch[k], 1), "write_size_kib_raw": round(write[k], 1),
            "hbm_bytes_per_launch": int((2 * fetch[k] + write[k]) * 1024),
            "fetch_correction": "x2 (gfx950 FETCH_SIZE reads half of wide coalesced streams)",
            "write_calibrated": k != "k_idct_color",
            "dispatches_averaged": [nf[k], nw[k]],
        }
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import hjdtools as T
    # stamped with the hash of the kernel sources the counters were collected on: bench.py ignores the file otherwise
    json.dump({"kernel_src_sha16": T.kernel_src_hash(), "stages": res}, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
