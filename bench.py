#!/usr/bin/env python3
"""bench.py — Mpixels/s JPEG->RGB on a synthetic 3840x2160 4:2:0 q85 baseline batch (BASELINE.json).

One "step" = one pass of the whole hot path (unstuff, Huffman synchronise + decode, dequantise,
IDCT, upsample, YCbCr->RGB) over one batch of JPEGs that is already resident in HBM; outputs
stay in HBM.  One process per GPU; images are independent, so ranks share nothing on the data
path (weak scaling: the per-GPU batch is fixed).  torch is used for the process group (barrier
and the max-over-ranks of the elapsed time) and for the HIP stream handle only.

Prints ONE JSON line (rank 0).
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(jpgs, w, h):
    """The reference's CPU path on this box's host cores, bounded sample.  kind 'reference' = the
    compiled reference itself (oracle/_ref/ref_big, built in the build container from the reference
    sources, single-threaded like the reference); 'port' = our C restatement (oracle/)."""
    import hjdtools as T
    ncores = os.cpu_count()
    model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    exe = T.ref_binary(big=True)
    sample = jpgs[:2]
    with tempfile.TemporaryDirectory() as td:
        if exe:
            t0 = time.time()
            for i, j in enumerate(sample):
                jp, bp = os.path.join(td, "%d.jpg" % i), os.path.join(td, "%d.bmp" % i)
                open(jp, "wb").write(j)
                T.ref_convert(jp, bp, big=True, timeout=600)
            dt = time.time() - t0
            kind = "reference"
        else:
            sample = jpgs[:8]
            t0 = time.time()
            for j in sample:
                rc, rgb, _, _, _ = T.oracle_decode(j)
                T.oracle_bmp(rgb)
            dt = time.time() - t0
            kind = "port"
    return {"value": round(len(sample) * w * h / dt / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": kind,
            "sample": "%d x %dx%d 4:2:0 q85 JPEG -> BMP, single thread on %s (host has %d cores), incl. file I/O"
                      % (len(sample), w, h, model, ncores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--quality", type=int, default=85)
    ap.add_argument("--distinct", type=int, default=4, help="distinct synthetic images replicated to fill the batch")
    ap.add_argument("--subseq-bits", type=int, default=0)
    ap.add_argument("--streams", type=int, default=4, help="split the per-GPU batch over this many HIP streams")
    ap.add_argument("--internal-streams", type=int, default=0,
                    help="instead: ONE batch, split inside the library (hjd_options.streams); no per-stage timing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the decode path has no CPU fallback")
    # rehearsal knobs (not used by the driver): several ranks on one GPU need gloo and a fixed device
    backend = os.environ.get("HJD_BENCH_BACKEND", "nccl")
    if "HJD_BENCH_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["HJD_BENCH_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            try:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            except Exception as e:  # the data path needs no collective: agreeing on the clock works over gloo too
                print("bench: RCCL init failed (%s); using gloo for the barrier/clock agreement" % e, file=sys.stderr)
                backend = "gloo"
                dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend=backend)

    import hjdtools as T
    hjd = importlib.import_module("hls-jpeg-decoder_amd")

    W, H = args.width, args.height
    distinct = [T.synth_jpeg(W, H, 1000 + 17 * rank + k, args.quality, 2, 2) for k in range(max(1, args.distinct))]
    jpgs = [distinct[i % len(distinct)] for i in range(args.batch)]

    if args.internal_streams > 1:
        args.streams = 1
    dec = hjd.Decoder(local_rank, subseq_bits=args.subseq_bits, streams=args.internal_streams)
    # parse + H2D: inputs are resident in HBM from here on.  The per-GPU batch may be split into
    # sub-batches that run on separate HIP streams, so that one sub-batch's latency-bound
    # synchronisation tail overlaps another's IDCT (same total work per step).
    ns = max(1, min(args.streams, args.batch))
    parts = [jpgs[k::ns] for k in range(ns)]
    batches = [dec.prepare(p, hjd.OUT_RGB) for p in parts]
    for b in batches:
        assert all(s == 0 for s in b.parse_status), b.parse_status
    tot = [b.totals() for b in batches]
    pixels, alg_bytes, scan_bytes = (sum(t[k] for t in tot) for k in range(3))
    tstreams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(ns - 1)]
    streams = [ts.cuda_stream for ts in tstreams]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        for b, sh in zip(batches, streams):
            b.run(sh)

    for _ in range(args.warmup):
        step()
    barrier()
    for b, sh in zip(batches, streams):
        st = b.finish(sh)
        assert all(s == 0 for s in st), st

    if args.internal_streams <= 1:
        for b in batches:
            b.enable_timing(args.steps)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0

    stages_all = []
    for b, sh in zip(batches, streams):
        st = b.finish(sh)
        assert all(s == 0 for s in st), st
        if b.resync_count() != 0:
            raise SystemExit("bench: the timed steps needed a host-driven re-synchronisation (%d rounds): "
                             "their outputs were not final inside the timed region" % b.resync_count())
        stg, runs = b.collect_timing()
        stages_all.append(stg)
    # per-launch stage durations: average over the sub-batches' launches (each launch covers 1/ns of the batch)
    stages = [(stages_all[0][k][0], sum(sa[k][1] for sa in stages_all) / ns) for k in range(len(stages_all[0]))] \
        if stages_all and stages_all[0] else []
    launch_alg_bytes = alg_bytes / ns

    # parity of what was just timed: compare images of the batch with the CPU oracle (checker only)
    parity = "ok"
    for b, p in ((batches[0], parts[0]), (batches[-1], parts[-1])):
        i = len(p) - 1
        rc, orgb, _, _, _ = T.oracle_decode(p[i])
        got = b.download(i)
        if rc != 0 or got.shape != orgb.shape or not (got == orgb).all():
            parity = "MISMATCH"
    if parity != "ok":
        raise SystemExit("bench: output differs from the oracle: " + parity)

    shard = importlib.import_module("hls-jpeg-decoder_amd.shard")
    dt, total_pixels = shard.agree(dt, pixels, device="cuda" if backend == "nccl" else "cpu")  # MAX of time, SUM of pixels

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = total_pixels * args.steps / dt / 1e6
        stage_ms = {n: round(ms, 4) for n, ms in stages}
        dom = max(stages, key=lambda x: x[1]) if stages else ("none", 0.0)
        kernel_of = {"unstuff": "k_unstuff_count+k_unstuff_write", "huff_sync": "k_sync", "huff_repair": "k_repair (x2)", "scan": "k_scan",
                     "huff_write": "k_write", "idct_color": "k_idct_color", "mask_clear": "hipMemsetAsync",
                     "reset": "hipMemsetAsync"}
        achieved = launch_alg_bytes / (dom[1] * 1e-3) / 1e9 if dom[1] > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get(dom[0])
                if ent and ent.get("images_per_launch") * ns == args.batch and ent.get("width") == W and ent.get("height") == H:
                    traffic = ent["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        out = {
            "metric": "Mpixels/s JPEG->RGB (4:2:0 baseline batch)",
            "value": round(value, 1),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8 entropy/pixel bytes, int16 coefficients, f32 IDCT+colour (reference arithmetic)",
            "data": "synthetic",
            "config": {"workload": "batch of %d synthetic %dx%d 4:2:0 q%d baseline JPEGs per GPU, resident in HBM, "
                                   "RGB out to HBM" % (args.batch, W, H, args.quality),
                       "images_per_gpu": args.batch, "distinct_images": len(distinct), "width": W, "height": H,
                       "scan_bytes_per_image": scan_bytes // max(1, args.batch),
                       "algorithmic_bytes_per_image": alg_bytes // max(1, args.batch), "parallelism": "images sharded, no collectives"},
            "roofline": {"bound": "hbm", "kernel": kernel_of.get(dom[0], dom[0]), "stage": dom[0],
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel_ms": round(dom[1], 4), "algorithmic_bytes_per_launch": int(launch_alg_bytes),
                         "launches_per_step": ns,
                         "pipeline_frac": round(alg_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            "stage_ms": stage_ms,
            "timed_runs": runs,
            "parity": parity,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(jpgs, W, H)
        print(json.dumps(out), flush=True)

    for b in batches:
        b.close()
    dec.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
