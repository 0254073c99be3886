#!/usr/bin/env python3
"""A/B runs of bench.py over several builds of libhjd.so on ONE box, interleaved (boxes differ by a few per cent; only
runs of the same call compare):

    python3 tools/ab_bench.py [--reps 3] [--check] [--bench-args "..."] name=path/to/libhjd.so ...

`name=` alone (empty path) is the in-tree library.  --check first runs a parity subset of tests/test_gpu_parity.py on every
build (HJD_LIB selects the library the Python package loads).  Prints one line per run and a summary table; writes the
summary as JSON to gpurun_out/ab_<tag>.json."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = "golden_rgb or full_size_synthetic or restart_fixtures or random_corpus or coefficients or mixed_batch or uniform_frames or thumbnail or byte_stuffing"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--tag", default="run")
    ap.add_argument("--bench-args", default="")
    ap.add_argument("--env", default="", help="extra environment for every run: A=1,B=2")
    ap.add_argument("builds", nargs="+")
    a = ap.parse_args()
    builds = [(b.split("=", 1)[0], b.split("=", 1)[1] if "=" in b else "") for b in a.builds]
    extra = dict(kv.split("=", 1) for kv in a.env.split(",") if kv)
    res = {n: [] for n, _ in builds}
    ok = {}
    for n, p in builds:
        env = dict(os.environ, **extra)
        if p:
            env["HJD_LIB"] = os.path.join(ROOT, p)
        if a.check:
            r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-x", "-q", "-m", "gpu", "-k", CHECK], cwd=ROOT, env=env,
                               capture_output=True, text=True)
            ok[n] = r.returncode == 0
            print("check %-10s %s" % (n, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]), flush=True)
    for rep in range(a.reps):
        for n, p in builds:
            if a.check and not ok[n]:
                continue
            env = dict(os.environ, **extra)
            if p:
                env["HJD_LIB"] = os.path.join(ROOT, p)
            r = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-end-to-end"] + a.bench_args.split(), cwd=ROOT, env=env,
                               capture_output=True, text=True)
            if r.returncode != 0:
                print("bench %-10s FAILED: %s" % (n, r.stderr[-400:]), flush=True)
                continue
            j = json.loads(r.stdout.strip().splitlines()[-1])
            sp = list(j["spread"].values())[0]
            res[n].append({"value": j["value"], "spread": sp, "iso": j["stage_ms_isolated"], "ovl": j["stage_ms_overlapped"]})
            print("bench %-10s rep %d: %8.1f  spread %s  iso %s" % (n, rep, j["value"], sp, {k: round(v, 3) for k, v in j["stage_ms_isolated"].items()}), flush=True)
    print("\n%-12s %10s %10s %10s   isolated stage ms (mean)" % ("build", "value", "spread1", "spread2"))
    summ = {}
    for n, _ in builds:
        rs = res[n]
        if not rs:
            continue
        mean = lambda xs: sum(xs) / len(xs)
        iso = {k: round(mean([r["iso"][k] for r in rs]), 4) for k in rs[0]["iso"]}
        summ[n] = {"value": round(mean([r["value"] for r in rs]), 1), "spread": [round(mean([r["spread"][i] for r in rs]), 1) for i in range(len(rs[0]["spread"]))],
                   "iso": iso, "runs": len(rs), "parity_subset_ok": ok.get(n)}
        print("%-12s %10.1f %s   %s" % (n, summ[n]["value"], " ".join("%10.1f" % x for x in summ[n]["spread"]), iso))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(summ, open(os.path.join(ROOT, "gpurun_out", "ab_%s.json" % a.tag), "w"), indent=1)


if __name__ == "__main__":
    main()
