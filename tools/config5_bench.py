#!/usr/bin/env python3
"""BASELINE config 5: mixed 4:4:4 / 4:2:2 / 4:2:0 (/ 4:4:0 / 4:1:1) subsampling with restart-interval markers, mixed sizes, on one
MI355X.  128 images (sizes 3840x2160 ... 250x187, half of them with DRI), resident in HBM, RGB out to HBM; every distinct image is
checked against the oracle (bit-exact) after the timed steps.  Measured the way the headline is (bench.py): the 128 images are
partitioned BY COST (shard.partition over the pixel counts) into --caller-streams batches, each run on a HIP stream of its own;
stage hipEvents inside the timed region; the same launches once more alone on the GPU.  Prints one JSON line.

    python3 tools/config5_bench.py [--hostile] [--caller-streams 4] [--internal-streams 1] [--steps 20]

--hostile: restart intervals of 7 MCUs and of every MCU, where the reference fires its resync checks early:
16 of the 128 images leave the clean-resync domain and go through the emulation kernels.
--caller-streams 1 --internal-streams N: ONE batch, split inside the library (hjd_options.streams; 0 = by batch size)."""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# --hw-queues N: the HIP runtime's GPU_MAX_HW_QUEUES (default 4 per process), set before the runtime loads.  Four caller streams
# beside the library's own want more than four hardware queues: streams that share one run one behind the other.
if "--hw-queues" in sys.argv:
    os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[sys.argv.index("--hw-queues") + 1]
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import hjdtools as T  # noqa: E402

hjd = importlib.import_module("hls-jpeg-decoder_amd")
shard = importlib.import_module("hls-jpeg-decoder_amd.shard")
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hostile", action="store_true")
    ap.add_argument("--caller-streams", type=int, default=4)
    ap.add_argument("--internal-streams", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--no-stage-events", action="store_true")
    ap.add_argument("--dirty-mode", type=int, default=0, help="1: restart streams outside the clean-resync domain are refused (no emulation launches)")
    ap.add_argument("--hw-queues", type=int, default=0, help="GPU_MAX_HW_QUEUES for this process (0: the runtime's default, 4)")
    ap.add_argument("--preheat-ms", type=float, default=150.0, help="untimed steps before the timed region (0: none)")
    ap.add_argument("--images", type=int, default=128, help="batch size (a multiple of 32: the 32 distinct images repeated)")
    a = ap.parse_args()
    sizes = [(3840, 2160), (1920, 1080), (1280, 720), (1000, 700), (777, 333), (640, 480), (512, 512), (250, 187)]
    samp = [(1, 1), (2, 1), (2, 2), (1, 1), (2, 1), (2, 2), (1, 2), (4, 1)]
    specs = []
    for k in range(32):  # 32 distinct images: size x sampling x (no DRI | DRI)
        w, h = sizes[k % 8]
        hf, vf = samp[(k // 2) % 8]
        mcux = -(-w // (8 * hf))
        # restart intervals an encoder would choose (one MCU row, four rows, two rows, 16 MCUs); --hostile: also 7 MCUs and every MCU
        rst = 0 if k % 2 == 0 else ([mcux, 4 * mcux, 7, 1] if a.hostile else [mcux, 4 * mcux, 2 * mcux, 16])[(k // 8) % 4]
        # busy content (intervals then end in blocks of 8 bits and more — the clean-resync domain); with 16-MCU intervals
        # the smooth-chroma image 27 needs sigma 12 for that (--hostile keeps sigma 7 everywhere)
        specs.append((w, h, 9000 + k, 85, hf, vf, rst, 12.0 if (rst == 16 and not a.hostile) else 7.0))
    src = [T.synth_jpeg(*s) for s in specs]
    N = max(32, a.images // 32 * 32)
    jpgs = [src[k % 32] for k in range(N)]
    px = [specs[k % 32][0] * specs[k % 32][1] for k in range(N)]
    pixels = sum(px)
    ns = max(1, a.caller_streams)
    parts = shard.partition(px, ns)  # by cost, as the ranks of a node are dealt their shards
    # the caller streams before the library creates any of its own: a hardware queue each (bench.py)
    torch.zeros(1, device="cuda:0")  # the default stream's hardware queue exists before any other stream is created
    torch.cuda.synchronize()
    tstreams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(max(1, a.caller_streams) - 1)]
    for ts in tstreams:  # ... and USED once: the runtime binds a stream to its hardware queue at the stream's first command
        with torch.cuda.stream(ts):
            torch.zeros(1, device="cuda:0")
    torch.cuda.synchronize()
    dec = hjd.Decoder(0, streams=a.internal_streams, dirty_mode=a.dirty_mode)
    batches = [dec.prepare([jpgs[i] for i in p]) for p in parts]
    for b in batches:
        assert all(s == 0 for s in b.parse_status)
    streams = [t.cuda_stream for t in tstreams]

    def step():
        for b, sh in zip(batches, streams):
            b.run(sh)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    for b, sh in zip(batches, streams):
        b.finish(sh)
    # pre-heat, as bench.py does (profiles/r05_warmup.txt): the GPU needs some tens of milliseconds of load after the seconds of
    # host-side setup above before it runs at its working clocks
    pre_t0 = time.perf_counter()
    while (time.perf_counter() - pre_t0) * 1e3 < a.preheat_ms:
        for _ in range(5):
            step()
        torch.cuda.synchronize()
    internal = a.internal_streams != 1  # stage events use the single-range form of the run: not with a split inside the library
    if not a.no_stage_events and not internal:
        for b in batches:
            b.enable_timing(a.steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    spread = []
    for _ in range(a.repeats):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        spread.append((time.perf_counter() - t1) / a.steps)
    sts, stages = [], []
    nseq = nres = nseg = 0
    for b, sh in zip(batches, streams):
        sts.append(b.finish(sh))
        nseq += b.sequential_count()
        nseg += b.segment_count()
        nres += b.resync_count()
        stg, _ = b.collect_timing() if (not a.no_stage_events and not internal) else ([], 0)
        stages.append(stg)
    ovl = {}
    if stages and stages[0]:
        for k, (name, _) in enumerate(stages[0]):
            ovl[name] = round(sum(s[k][1] for s in stages) / ns, 4)
    iso = {}
    for b in batches:  # the same launches alone on the GPU
        for _ in range(2):
            torch.cuda.synchronize()
            for name, ms in b.run_timed(streams[0]):
                iso[name] = iso.get(name, 0.0) + ms / (2 * ns)
        b.finish(streams[0])
    ok = 0
    want = {}
    for k in range(32):
        want[k] = T.oracle_decode(src[k])
    for p, b, st in zip(parts, batches, sts):
        seen = set()
        for i, g in enumerate(p):
            k = g % 32
            if k in seen:
                continue
            seen.add(k)
            orc, orgb = want[k][0], want[k][1]
            if orc == 0:
                assert st[i] == 0, (specs[k], st[i])
                got = b.download(i)
                assert np.array_equal(got, orgb), (specs[k], int((got != orgb).any(axis=2).sum()))
                ok += 1
            else:
                assert st[i] != 0, specs[k]
    tot = [b.totals() for b in batches]
    alg, scan = sum(t[1] for t in tot), sum(t[2] for t in tot)
    dom = max(((n, v) for n, v in ovl.items() if n != "sync_wait"), key=lambda x: x[1]) if ovl else max(iso.items(), key=lambda x: x[1])
    launch_alg = alg / ns
    out = {"metric": "Mpixels/s JPEG->RGB",
           "config": {"workload": "BASELINE config 5: %d images, 8 sizes 3840x2160..250x187, 4:4:4 / 4:2:2 / 4:2:0 / 4:4:0 / 4:1:1, half of them with "
                                  "DRI (intervals of %s), resident in HBM, RGB to HBM; partitioned by pixel count into %d batch(es), one HIP stream each"
                                  % (N, "one MCU row, four rows, 7 MCUs, 1 MCU" if a.hostile else "one MCU row, four rows, two rows, 16 MCUs", ns),
                      "images": N, "distinct_images": 32, "caller_streams": ns, "internal_streams": a.internal_streams, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "default (4)"),
                      "with_dri": sum(1 for s in specs if s[6]) * (N // 32), "scan_bytes": scan, "algorithmic_bytes": alg},
           "value": round(pixels / dt / 1e6, 1), "unit": "Mpixels/s", "ms_per_step": round(dt * 1e3, 4), "steps": a.steps,
           "spread": [round(pixels / x / 1e6, 1) for x in spread],
           "stage_ms_overlapped": ovl, "stage_ms_isolated": {k: round(v, 4) for k, v in iso.items()},
           "roofline": {"bound": "hbm", "stage": dom[0], "kernel_ms": round(dom[1], 4), "achieved": round(launch_alg / (dom[1] * 1e-3) / 1e9, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(launch_alg / (dom[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        "algorithmic_bytes_per_launch": int(launch_alg), "pipeline_frac": round(alg / dt / 1e9 / HBM_PEAK_GBS, 5),
                        "timing": "hipEvents around the stage inside the timed region" if ovl else "isolated replay"},
           "sequential_emulations": nseq, "segment_parallel_emulations": nseg, "resyncs": nres,
           "parity": "bit-exact vs oracle on %d distinct (batch, image) pairs; the others are refused by both" % ok}
    print(json.dumps(out))
    for b in batches:
        b.close()
    dec.close()


if __name__ == "__main__":
    main()
