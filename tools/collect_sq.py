#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc SQ counter passes (tools/gpu_profile.sh) into profiles/sq_counters.json.

Per kernel, averaged over the dispatches of the pass (bench.py with ONE launch of `images_per_launch`
images on one stream): raw counters, kernel duration, and the derived figures DESIGN.md quotes:

  valu_per_block        SQ_INSTS_VALU (wave-instructions) / 8x8 blocks of the launch
  valu_issue_frac       SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x 2.4 GHz x kernel time): the share of the chip's
                        vector-issue slots the kernel's instructions fill (MI355X_MICROARCH.md: a wave64 VALU
                        instruction occupies its SIMD-32 for 2 cycles)
  waves_per_simd        SQ_WAVE_CYCLES x 4 (the counter is in quad-cycles) / (kernel cycles x 1024)
  lds_conflict_frac     SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait_inst_frac, wait_any_frac   shares of the wave-cycles spent issue-stalled / parked at a wait

usage: collect_sq.py <out.json> <images_per_launch> <width> <height> <blocks_per_image> <pmc_dir> [<pmc_dir> ...]
"""
import collections
import csv
import glob
import json
import os
import re
import sys

STAGE_OF = {"k_idct_color": "idct_color", "k_sync": "huff_sync", "k_repair": "huff_repair_first", "k_repair_totals": "huff_repair", "k_write": "huff_write",
            "k_unstuff_write": "unstuff", "k_unstuff_count": "unstuff_count", "k_dcapply": "dc_scan", "k_dcsum": "dc_sum",
            "k_scan": "scan", "k_gate": "repair_gate"}
CLK = 2.4e9
SIMDS = 1024


def main():
    out, ipl, w, h, bpi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(collections.Counter)
    dur = collections.defaultdict(list)
    meta = {}
    for d in sys.argv[6:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                k = re.sub(r"<.*", "", r["Kernel_Name"].split("(")[0].replace("void ", "")).strip()
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[k][r["Counter_Name"]] += 1
                key = (f, r["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
                    dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                    meta[k] = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "lds_block_size": int(r["LDS_Block_Size"]),
                               "workgroup_size": int(r["Workgroup_Size"])}
    res = {}
    for k, stage in STAGE_OF.items():
        if k not in tot:
            continue
        c = {n: tot[k][n] / cnt[k][n] for n in tot[k]}
        t_ns = sum(dur[k]) / len(dur[k])
        cyc = t_ns * 1e-9 * CLK
        ent = {"kernel": k, "images_per_launch": ipl, "width": w, "height": h, "dispatches_averaged": len(dur[k]),
               "kernel_ns_under_pmc": round(t_ns), "counters": {n: round(v) for n, v in sorted(c.items())}}
        ent.update(meta[k])
        if "SQ_INSTS_VALU" in c:
            ent["valu_insts_per_launch"] = round(c["SQ_INSTS_VALU"])
            ent["valu_per_block"] = round(c["SQ_INSTS_VALU"] / (ipl * bpi), 2)
            ent["valu_issue_frac"] = round(c["SQ_INSTS_VALU"] * 2 / (SIMDS * cyc), 4)
        if "SQ_INSTS_LDS" in c:
            ent["lds_insts_per_block"] = round(c["SQ_INSTS_LDS"] / (ipl * bpi), 2)
        if "SQ_WAVE_CYCLES" in c:
            ent["waves_per_simd"] = round(c["SQ_WAVE_CYCLES"] * 4 / (cyc * SIMDS), 2)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            ent["lds_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
            ent["lds_active_frac_of_cu_cycles"] = round(c["SQ_LDS_IDX_ACTIVE"] / (256 * cyc), 4)
        if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_INST_ANY" in c:
            ent["wait_inst_frac"] = round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4)
            ent["wait_any_frac"] = round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)
        res[stage] = ent
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import hjdtools as T
    # stamped with the hash of the kernel sources the counters were collected on: bench.py ignores the file otherwise
    json.dump({"kernel_src_sha16": T.kernel_src_hash(), "stages": res}, open(out, "w"), indent=1, sort_keys=True)
    for s, e in res.items():
        print(s, {a: b for a, b in e.items() if a not in ("counters",)})


if __name__ == "__main__":
    main()
