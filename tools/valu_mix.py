#!/usr/bin/env python3
"""Static vector-instruction mix of every kernel in libhjd.so -> profiles/valu_mix.json.

For each kernel symbol of the gfx950 code objects: the VALU instructions of its ISA by issue class, and the average
issue cost per instruction under two price lists:

  nominal   what the hardware cannot beat (/opt/skills/guides/MI355X_MICROARCH.md: a wave64 VALU instruction occupies its
            SIMD-32 for 2 cycles; the packed fp32 forms v_pk_{mul,add,fma}_f32 carry two results per lane = 4 cycles, which
            is what the 157 TF vector peak corresponds to; transcendentals 8)
  measured  profiles/r02_valu_rate.txt (tools/valu_rate.hip, 8 waves per SIMD, independent chains): moves 2.4; two-operand
            integer / fp32 add / mul 2.9; shifts, three-operand forms, conversions, compares, integer multiplies 4.4;
            packed fp32 5.0

bench.py prices a kernel's measured SQ_INSTS_VALU with the NOMINAL cost of its own mix: a fraction of a bound no kernel can
exceed.  The mix is static (every instruction of the kernel body counted once), which under-weights the hot loops; the
file says so.  Stamped with the hash of the kernel sources (hjdtools.kernel_src_hash): bench.py ignores a stale file.

usage: valu_mix.py [out.json]      (runs anywhere: it only disassembles the library)
"""
import collections
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
CHEAP2 = ("v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_f32", "v_sub_f32", "v_subrev_f32",
          "v_mul_f32", "v_add_co_u32", "v_sub_co_u32", "v_addc_co_u32", "v_not_b32", "v_min_u32", "v_max_u32", "v_min_f32", "v_max_f32")


def classify(op):
    if op.startswith("v_pk_") and op.endswith("_f32"):
        return "packed_f32"
    if op.startswith(TRANS):
        return "transcendental"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "move"
    if op.split("_e32")[0].split("_e64")[0].split("_dpp")[0].split("_sdwa")[0] in CHEAP2:
        return "simple"
    return "other"


NOMINAL = {"packed_f32": 4.0, "transcendental": 8.0, "move": 2.0, "simple": 2.0, "other": 2.0}
MEASURED = {"packed_f32": 5.0, "transcendental": 8.0, "move": 2.4, "simple": 2.9, "other": 4.4}


def main():
    import hjdtools as T
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "valu_mix.json")
    so = os.path.join(ROOT, "hls-jpeg-decoder_amd", "lib", "libhjd.so")
    res = {"kernel_src_sha16": T.kernel_src_hash(), "mix": "static (every instruction of the kernel body once) and, where the executed fp32 counters exist, dynamic", "kernels": {}}
    with tempfile.TemporaryDirectory() as td:
        shutil.copy(so, td)
        subprocess.run([OBJDUMP, "--offloading", os.path.join(td, "libhjd.so")], check=True, capture_output=True)
        for f in sorted(os.listdir(td)):
            if "amdgcn" not in f:
                continue
            dis = subprocess.run([OBJDUMP, "-d", os.path.join(td, f)], check=True, capture_output=True, text=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    name = m.group(1)
                    dm = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                    cur = re.sub(r"\(.*", "", dm.replace("void ", "")).strip()
                    if not cur.startswith("k_"):
                        cur = None
                    else:
                        res["kernels"].setdefault(cur, collections.Counter())
                    continue
                if cur is None:
                    continue
                t = line.strip().split()
                if t and t[0].startswith("v_") and not t[0].startswith("v_mfma"):
                    res["kernels"][cur][classify(t[0])] += 1
                    if re.match(r"v_(add|sub|subrev|mul)_f32", t[0]):
                        res["kernels"][cur]["_scalar_f32_addmul"] += 1
    # DYNAMIC mix (round 5): profiles/sq_counters.json now carries, per kernel, the EXECUTED fp32 add and multiply
    # instructions (SQ_INSTS_VALU_ADD_F32 / _MUL_F32, a third PMC pass).  Of those the packed share is taken from the ISA (packed
    # against scalar fp32 add / mul instructions of the kernel): executed packed = (ADD_F32 + MUL_F32) x that share, each priced at
    # 4 cycles, everything else at 2.  The static average stays in the file for comparison; bench.py uses the dynamic one.
    dyn = {}
    try:
        sq = json.load(open(os.path.join(ROOT, "profiles", "sq_counters.json")))
        if sq.get("kernel_src_sha16") == res["kernel_src_sha16"]:
            for st in sq.get("stages", {}).values():
                c = st.get("counters", {})
                if "SQ_INSTS_VALU_ADD_F32" in c and c.get("SQ_INSTS_VALU"):
                    dyn[st["kernel"]] = (c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_MUL_F32"], c["SQ_INSTS_VALU"])
    except Exception:
        pass
    for k, c in list(res["kernels"].items()):
        scalar = c.pop("_scalar_f32_addmul", 0)
        n = sum(c.values())
        ent_dyn = {}
        if k in dyn or k.split("<")[0] in dyn:
            fp, tot = dyn.get(k, dyn.get(k.split("<")[0]))
            share = c.get("packed_f32", 0) / max(1, c.get("packed_f32", 0) + scalar)
            packed_exec = fp * share
            ent_dyn = {"executed_fp32_addmul_per_valu": round(fp / tot, 4), "packed_share_of_fp32_addmul_static": round(share, 4),
                       "nominal_cycles_per_inst_dynamic": round((2.0 * (tot - packed_exec) + 4.0 * packed_exec) / tot, 3)}
        res["kernels"][k] = {"valu_static": n, "by_class": dict(sorted(c.items())), **ent_dyn,
                             "nominal_cycles_per_inst": round(sum(NOMINAL[x] * v for x, v in c.items()) / max(n, 1), 3),
                             "measured_cycles_per_inst": round(sum(MEASURED[x] * v for x, v in c.items()) / max(n, 1), 3)}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res["kernels"].items()):
        print(k, v)


if __name__ == "__main__":
    main()
