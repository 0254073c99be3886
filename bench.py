#!/usr/bin/env python3
"""bench.py — Mpixels/s JPEG->RGB on a synthetic 3840x2160 4:2:0 q85 baseline batch (BASELINE.json).

One "step" = one pass of the whole hot path (unstuff, Huffman synchronise + decode, dequantise,
IDCT, upsample, YCbCr->RGB) over one batch of JPEGs that is already resident in HBM; outputs
stay in HBM.  One process per GPU.  ONE global batch of N x --batch images is partitioned over the
N ranks by hls-jpeg-decoder_amd/shard.py::partition (images are independent: no collective on
the data path; weak scaling: the per-GPU share is fixed as N grows).  torch is used for the
process group (barrier, MAX of the elapsed time, SUM of the pixels) and the HIP stream handle only.

`python bench.py --gpus N` with N > 1 and no RANK in the environment starts the N ranks itself:
the parent never touches a GPU, spawns N fresh children with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set, relays rank 0's JSON line and exits non-zero if any rank failed.  Under
`python -m torch.distributed.run ... bench.py --gpus N` the environment is already set.

Prints ONE JSON line (rank 0).
"""
import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
# VALU issue, G wave-instructions/s: nominal = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (the guide's
# figure); measured = the issue cost of the instruction kinds these kernels are made of, on this chip, 8 waves per SIMD,
# independent chains (tools/valu_rate.hip, profiles/r02_valu_rate.txt), in cycles per wave-instruction per SIMD at 2.4 GHz:
# v_mov 2.4; v_and / v_add_u32 / v_add_f32 / v_mul_f32 2.8-3.1; shifts, v_lshl_add, v_bfe, v_ffbl, conversions, v_trunc, v_fma,
# v_max3, compares, integer multiplies 4.2-4.7; v_pk_mul_f32 / v_pk_add_f32 5.0.  The kernels' mix (a third packed, half of
# the 4.2+ class, a sixth cheap) averages about 4.3.
VALU_PEAK_NOMINAL_GINST = 1024 * 2.4 / 2


def host_cores():
    """CPU cores this process may actually use (affinity mask and cgroup quota, whichever is smaller)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def _ref_one(args):
    import hjdtools as T
    j, td, i = args
    jp, bp = os.path.join(td, "%d.jpg" % i), os.path.join(td, "%d.bmp" % i)
    open(jp, "wb").write(j)
    return T.ref_convert(jp, bp, big=True, timeout=900)


def _port_one(j):
    import hjdtools as T
    rc, rgb, _, _, _ = T.oracle_decode(j)
    T.oracle_bmp(rgb)
    return rc


def cpu_baseline(distinct, w, h):
    """The reference's CPU path on this box's host cores, bounded sample: (i) single thread, like
    the reference itself, and (ii) one image per core over all usable host cores.  kind 'reference' = the
    compiled reference (oracle/_ref/ref_big, built in the build container from the reference's sources);
    'port' = our C restatement (oracle/) when that binary is absent."""
    import hjdtools as T
    from concurrent.futures import ProcessPoolExecutor
    cores, model = host_cores(), cpu_model()
    have_ref = T.ref_binary(big=True) is not None
    kind = "reference" if have_ref else "port"
    with tempfile.TemporaryDirectory() as td:
        sample = distinct[:2]
        t0 = time.time()
        for i, j in enumerate(sample):
            rc = _ref_one((j, td, i)) if have_ref else _port_one(j)
            assert rc == 0
        dt1 = time.time() - t0
        work = [distinct[i % len(distinct)] for i in range(cores)]
        t0 = time.time()
        with ProcessPoolExecutor(max_workers=cores) as ex:
            rcs = list(ex.map(_ref_one, [(j, td, 100 + i) for i, j in enumerate(work)])) if have_ref \
                else list(ex.map(_port_one, work))
        dtn = time.time() - t0
        assert all(r == 0 for r in rcs)
    return {"value": round(len(sample) * w * h / dt1 / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": kind,
            "sample": "%d x %dx%d 4:2:0 q85 JPEG -> BMP, single thread on %s, incl. file I/O" % (len(sample), w, h, model),
            "all_cores": {"value": round(len(work) * w * h / dtn / 1e6, 3), "unit": "Mpixels/s", "cores": cores,
                          "sample": "%d images, one process per core (%d usable of %d)" % (len(work), cores, os.cpu_count() or 0)}}


def PG_TIMEOUT_S():
    """Timeout of the process group (rendezvous and every collective), seconds: a rank that is missing makes the job
    fail after this long, not after torch's 10-30 min default."""
    return float(os.environ.get("HJD_BENCH_PG_TIMEOUT", "120"))


def spawn_ranks(n, argv):
    """Parent of a self-launched multi-rank run.  Touches no GPU API (the children must be able to
    initialise their devices freshly); returns the exit code for the whole job.  ALL children are polled: as soon
    as one exits non-zero the others (which may be sitting in the rendezvous or a barrier waiting for it) are
    terminated and the job fails within seconds, not after the process group's 10-30 min timeout."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    out0 = tempfile.TemporaryFile()  # rank 0's stdout goes to a file: nobody has to drain a pipe while polling
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    rcs = [None] * n
    failed = False
    while any(c is None for c in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        if any(c not in (None, 0) for c in rcs):
            failed = True
            break
        time.sleep(0.05)
    if failed:
        for r, p in enumerate(procs):
            if rcs[r] is None:
                p.terminate()
        deadline = time.time() + 5.0
        for r, p in enumerate(procs):
            if rcs[r] is None:
                try:
                    p.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
                rcs[r] = "terminated"
        bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
        print("bench: ranks failed: %s" % bad, file=sys.stderr)
        return 1
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    return 0


def dry_rank(args, rank, world):
    """HJD_BENCH_DRY=1 — rehearsal of the multi-rank plumbing WITHOUT a GPU (gloo): rendezvous, the partition of the one
    global batch, barrier, MAX-of-time / SUM-of-pixels agreement and rank 0's line, for world sizes this pool does not let
    a builder run on a card (BASELINE config 4: 8 ranks).  Decodes nothing and says so: "dry": true, value from a made-up clock."""
    import torch.distributed as dist
    shard = importlib.import_module("hls-jpeg-decoder_amd.shard")
    backend, seen = "none", 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = shard.fail_fast("init_process_group", shard.init_group, "gloo", None, PG_TIMEOUT_S())
        seen = shard.fail_fast("rank count", shard.ranks_seen, "cpu")
    n_global = world * args.batch
    parts = shard.partition([args.width * args.height] * n_global, world)
    assert sorted(sum(parts, [])) == list(range(n_global))
    mine = parts[rank]
    if os.environ.get("HJD_BENCH_HANG_RANK") == str(rank):  # test knob: this rank never reaches the barrier
        time.sleep(3600)
    if world > 1:
        shard.fail_fast("barrier", dist.barrier)
    own_dt = 0.001 * (rank + 1) * args.steps
    dt, px = shard.fail_fast("clock agreement", shard.agree, own_dt,
                             float(len(mine) * args.width * args.height), device="cpu")
    per_rank = shard.fail_fast("per-rank times", shard.gather_per_rank, rank, -1, own_dt, args.steps, device="cpu")
    if rank == 0:
        print(json.dumps({"metric": "Mpixels/s JPEG->RGB (4:2:0 baseline batch)", "dry": True, "n_gpus": world, "steps": args.steps,
                          "backend": backend, "ranks_seen": seen, "per_rank": per_rank,
                          "value": round(px * args.steps / dt / 1e6, 1), "ms_per_step": round(dt / args.steps * 1e3, 4),
                          "config": {"images_per_gpu": len(mine), "global_images": n_global}}), flush=True)
    if world > 1:
        shard.fail_fast("barrier", dist.barrier)
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--quality", type=int, default=85)
    ap.add_argument("--content", choices=("synthetic", "photo"), default="synthetic",
                    help="photo: tests/golden/Lenna.jpg (decoded by this library) mirror-tiled to the frame size, re-encoded "
                         "with per-image optimised Huffman tables — not the driver's workload, a second data point")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic images per GPU, replicated to fill its share (16: profiles/r04_bench_distinct16.json)")
    ap.add_argument("--subseq-bits", type=int, default=0)
    ap.add_argument("--auto-split", action="store_true", help="hjd_options.streams = 0 (the library's own choice by batch size) beside the caller streams")
    ap.add_argument("--launches-per-stream", type=int, default=1, help="batches per caller stream and step (the rank's share is dealt to streams x this many batches)")
    ap.add_argument("--streams", type=int, default=4, help="split the per-GPU share over this many HIP streams")
    ap.add_argument("--internal-streams", type=int, default=0,
                    help="instead: ONE batch, split inside the library (hjd_options.streams); no per-stage timing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-stage-events", action="store_true", help="diagnostic: no hipEvents between the kernels of the timed steps")
    ap.add_argument("--stage-events", choices=("one", "all"), default="one",
                    help="which launches of a step carry stage hipEvents in the reported region: those of ONE stream (default: every "
                         "step has one instrumented launch) or of all (no measurable difference: profiles/r05_warmup.txt)")
    ap.add_argument("--preheat-ms", type=float, default=150.0,
                    help="untimed steps run for this long BEFORE the --warmup steps: the GPU comes out of the minutes of host-side setup at its "
                         "idle clocks and needs some tens of milliseconds of load to reach its working state (a 5-step warm-up is 8 ms); "
                         "0 = none.  Reported in the line as `preheat`.")
    ap.add_argument("--repeats", type=int, default=2, help="further timed regions of --steps steps after the reported one: their values are listed as `spread`")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU" % (args.gpus, world))
    if os.environ.get("HJD_BENCH_FAIL_RANK") == str(rank):  # test knob: this rank dies before the rendezvous
        raise SystemExit("bench: rank %d told to fail (HJD_BENCH_FAIL_RANK)" % rank)
    if os.environ.get("HJD_BENCH_DRY") == "1":
        return dry_rank(args, rank, world)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the decode path has no CPU fallback")
    # rehearsal knobs (not used by the driver): several ranks on one GPU need gloo and a fixed device
    backend = os.environ.get("HJD_BENCH_BACKEND", "nccl")
    if "HJD_BENCH_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["HJD_BENCH_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    import hjdtools as T
    hjd = importlib.import_module("hls-jpeg-decoder_amd")
    shard = importlib.import_module("hls-jpeg-decoder_amd.shard")
    seen = 1
    # HJD_BENCH_GROUP_AT_ONE=1 (rehearsal, not used by the driver): a process group even for ONE rank, so that the RCCL branch —
    # eager communicator on the rank's device, barrier, all-reduce, all-gather — executes on a one-GPU box
    grouped = world > 1 or os.environ.get("HJD_BENCH_GROUP_AT_ONE") == "1"

    def join_group():
        # explicit timeout; RCCL comes up eagerly on this rank's device (a rank that cannot reach its peers fails HERE);
        # the backend really in use and the number of ranks that answered an all-reduce of ones go into the JSON line
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        be = shard.fail_fast("init_process_group", shard.init_group, backend, local_rank, PG_TIMEOUT_S())
        assert dist.get_world_size() == args.gpus
        n = shard.fail_fast("rank count", shard.ranks_seen, "cuda" if be == "nccl" else "cpu")
        if n != world:
            raise SystemExit("bench: %d ranks answered, %d expected" % (n, world))
        return be, n
    # The group is joined AFTER the caller's streams and the decoder's exist (below) unless HJD_BENCH_GROUP_EARLY=1: RCCL creates
    # streams of its own, and a stream takes its hardware queue at its first command — the four queues of the process go to the
    # streams that carry the kernels, RCCL's (idle inside the timed region) share.
    if not grouped:
        backend = "none"
    elif os.environ.get("HJD_BENCH_GROUP_EARLY") == "1":
        backend, seen = join_group()

    # ONE global batch of world x batch images, image g = synthetic source number g mod (distinct x world);
    # every rank computes the same partition (cost = pixels) and decodes its own shard.
    W, H = args.width, args.height
    n_global = world * args.batch
    n_src = max(1, args.distinct) * world
    parts_global = shard.partition([W * H] * n_global, world)
    mine = parts_global[rank]
    assert sorted(sum(parts_global, [])) == list(range(n_global))
    src_ids = sorted({g % n_src for g in mine})
    if args.internal_streams > 1 or args.internal_streams < 0:  # the split is the library's (-1: its own choice by batch size): one caller stream
        args.streams = 1
    # The caller streams FIRST, before the library creates any of its own: the HIP runtime gives the first four streams of a process
    # (the default stream and the next three) a hardware queue each and lets every later one share the least used — two caller
    # streams that share a queue run one behind the other (round 5, rocprofv3 Queue_Id: four fresh streams created after the
    # decoder sat on three queues, 288 Gpixel/s against 323; BASELINE config 5: 131 against 169).  INTEGRATION.md §streams.
    nstreams = max(1, min(args.streams, len(mine)))
    torch.zeros(1, device="cuda:%d" % local_rank)  # the default stream's hardware queue exists before any other stream is created
    torch.cuda.synchronize()
    tstreams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nstreams - 1)]
    for ts in tstreams:  # ... and USED once: the runtime binds a stream to its hardware queue at the stream's first command
        with torch.cuda.stream(ts):
            torch.zeros(1, device="cuda:%d" % local_rank)
    torch.cuda.synchronize()
    dec = hjd.Decoder(local_rank, subseq_bits=args.subseq_bits, streams=0 if (args.internal_streams < 0 or args.auto_split) else max(1, args.internal_streams))  # 1 = no split inside the library, 0 = the library's choice
    if grouped and os.environ.get("HJD_BENCH_GROUP_EARLY") != "1":
        backend, seen = join_group()
    if args.content == "photo":
        import numpy as np
        pb = dec.prepare([open(os.path.join(ROOT, "tests", "golden", "Lenna.jpg"), "rb").read()], hjd.OUT_RGB)
        pb.run()
        assert pb.finish() == [0]
        tile = pb.download(0)
        del pb
        th, tw, _ = tile.shape
        quad = np.concatenate([np.concatenate([tile, tile[:, ::-1]], 1), np.concatenate([tile[::-1], tile[::-1, ::-1]], 1)], 0)

        def photo(s):  # mirror tiling (no seams), another origin per source
            oy, ox = (s * 97) % (2 * th), (s * 211) % (2 * tw)
            big = np.tile(quad, ((H + oy) // (2 * th) + 1, (W + ox) // (2 * tw) + 1, 1))
            return T.encode_jpeg_ex(np.ascontiguousarray(big[oy:oy + H, ox:ox + W]), args.quality, 2, 2,
                                    huff_mode=T.HUFF_FOUR_OPTIMISED)
        sources = {s: photo(s) for s in src_ids}
    else:
        sources = {s: T.synth_jpeg(W, H, 1000 + s, args.quality, 2, 2) for s in src_ids}
    jpgs = [sources[g % n_src] for g in mine]
    # parse + H2D: inputs are resident in HBM from here on.  The per-GPU share may be split into
    # sub-batches that run on separate HIP streams, so that one sub-batch's latency-bound
    # synchronisation tail overlaps another's IDCT (same total work per step).
    # ns = batches = launches of every kernel per step; --launches-per-stream L > 1: L batches one behind the other on every stream
    ns = max(1, min(nstreams * max(1, args.launches_per_stream), len(jpgs)))
    parts = [jpgs[k::ns] for k in range(ns)]
    part_src = [[mine[i] % n_src for i in range(k, len(jpgs), ns)] for k in range(ns)]
    batches = [dec.prepare(p, hjd.OUT_RGB) for p in parts]
    for b in batches:
        assert all(s == 0 for s in b.parse_status), b.parse_status
    tot = [b.totals() for b in batches]
    pixels, alg_bytes, scan_bytes = (sum(t[k] for t in tot) for k in range(3))
    streams = [tstreams[j % nstreams].cuda_stream for j in range(ns)]

    def barrier():
        if grouped:  # a failed barrier ends this rank (non-zero) at once; the launcher ends the others
            if backend == "nccl":
                shard.fail_fast("barrier", dist.barrier, device_ids=[local_rank])
            else:
                shard.fail_fast("barrier", dist.barrier)
        torch.cuda.synchronize()

    def step():
        for b, sh in zip(batches, streams):
            b.run(sh)

    # Everything that is not a step happens BEFORE the warm-up steps — the stage events are created here and the warm-up runs use
    # the first of them — so that between the last warm-up step and the first timed one lies a barrier and nothing else: with the
    # statuses fetched and 340 events created in between, the timed region began on a GPU that had been idle for a millisecond
    # or two (the first region of a run was 5 % below the following ones, with or without stage events).
    timed_batches = [] if (args.internal_streams > 1 or args.internal_streams < 0 or args.no_stage_events) else (batches[:1] if args.stage_events == "one" else batches)
    # Pre-heat (untimed, before the W warm-up steps): the same step, repeated until --preheat-ms have passed.  The setup above is
    # seconds of host work during which the GPU sits at its idle clocks; measured (round 5): with --warmup 5 the timed region ran
    # 4-5 % below the regions after it, with --warmup 40 level with them, with or without stage events (profiles/r05_warmup.txt).
    preheat_steps, preheat_t0 = 0, time.perf_counter()
    while (time.perf_counter() - preheat_t0) * 1e3 < args.preheat_ms:
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        preheat_steps += 5
    for b in timed_batches:
        b.enable_timing(args.steps + args.warmup)
    for _ in range(args.warmup):
        step()
    barrier()
    for b in timed_batches:
        b.collect_timing()  # the warm-up runs' intervals are dropped; the events are free for the timed steps
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0

    # the same region again (not the reported value): how much one box varies from region to region
    spread_dt = []
    for _ in range(max(0, args.repeats)):
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        spread_dt.append(time.perf_counter() - t1)

    stages_all = []
    runs = 0
    for b, sh in zip(batches, streams):
        st = b.finish(sh)
        assert all(s == 0 for s in st), st
        if b.resync_count() != 0:
            raise SystemExit("bench: the timed steps needed a host-driven re-synchronisation (%d rounds): "
                             "their outputs were not final inside the timed region" % b.resync_count())
        if b in timed_batches:
            stg, runs = b.collect_timing()
            stages_all.append(stg)
    # per-launch stage durations INSIDE the timed region: other streams' kernels share the GPU (overlapped)
    stages_ovl = [(stages_all[0][k][0], sum(sa[k][1] for sa in stages_all) / len(stages_all)) for k in range(len(stages_all[0]))] \
        if stages_all and stages_all[0] else []
    launch_alg_bytes = alg_bytes / ns

    # the same launches once more, ALONE on the GPU (one stream, nothing beside them): kernel times proper
    iso_acc = {}
    iso_order = []
    ISO_REP = 3
    for b in batches:
        for _ in range(ISO_REP):
            torch.cuda.synchronize()
            for name, ms in b.run_timed(streams[0]):
                if name not in iso_acc:
                    iso_acc[name] = 0.0
                    iso_order.append(name)
                iso_acc[name] += ms
        assert all(s == 0 for s in b.finish(streams[0]))
    stages_iso = [(n, iso_acc[n] / (ISO_REP * len(batches))) for n in iso_order]

    # parity of what was just timed: reference-produced hashes where the manifest has this input, else the oracle
    parity = "ok"
    parity_vs = set()
    for b, p, srcs in ((batches[0], parts[0], part_src[0]), (batches[-1], parts[-1], part_src[-1])):
        i = len(p) - 1
        got = b.download(i)
        ent = T.large_golden(W, H, 1000 + srcs[i], args.quality, 2, 2, p[i])
        if ent is not None:
            parity_vs.add("reference hash")
            if hashlib.sha256(got.tobytes()).hexdigest() != ent["rgb_sha256"]:
                parity = "MISMATCH"
        else:
            parity_vs.add("oracle")
            rc, orgb, _, _, _ = T.oracle_decode(p[i])
            if rc != 0 or got.shape != orgb.shape or not (got == orgb).all():
                parity = "MISMATCH"
    if parity != "ok":
        raise SystemExit("bench: output differs from the reference: " + parity)

    # end to end on this rank (never `value`): host JPEG bytes -> host RGB bytes through the pipelined entry point
    e2e = None
    if rank == 0 and not args.no_end_to_end and hasattr(dec, "decode_host_pipelined_bench"):
        e2e = dec.decode_host_pipelined_bench(jpgs, W, H)

    adev = "cuda" if backend == "nccl" else "cpu"
    per_rank = shard.fail_fast("per-rank times", shard.gather_per_rank, rank, local_rank, dt, args.steps, device=adev, alone_too=grouped)  # before MAX replaces dt
    dt, total_pixels = shard.fail_fast("clock agreement", shard.agree, dt, pixels, device=adev, alone_too=grouped)  # MAX of time, SUM of pixels
    spread_dt = [shard.fail_fast("clock agreement", shard.agree, x, 0.0, device=adev)[0] for x in spread_dt]

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = total_pixels * args.steps / dt / 1e6
        ovl = dict(stages_ovl)
        iso = dict(stages_iso)
        kernel_of = {"unstuff": "k_unstuff_count+k_unstuff_write", "huff_sync": "k_sync", "huff_repair": "k_repair_totals",
                     "huff_write": "k_write", "dc_scan": "k_dcsum+k_dcapply", "idct_color": "k_idct_color"}
        images_per_launch = len(jpgs) // ns
        # The dominant kernel and its duration come from INSIDE the timed region (hipEvents on the stream each launch goes
        # to; other streams' kernels run beside it) — the same quantity `rocprofv3 --kernel-trace --stats` of this command
        # averages (profiles/r03_kernel_stats.csv).  The isolated replay (same launches, alone on the GPU) is a named extra.
        timing = "overlapped" if stages_ovl else "isolated"
        src = stages_ovl if stages_ovl else stages_iso
        # "sync_wait" is no kernel: the interval a run waits for another run's synchronisation (hjd_decoder::sync_conc)
        dom = max((x for x in src if x[0] != "sync_wait"), key=lambda x: x[1]) if src else ("none", 0.0)
        src_hash = T.kernel_src_hash()
        dropped = []

        def committed(fname):
            """per-stage PMC summaries under profiles/ (tools/gpu_profile.sh on the same launch shape).  A summary is used
            only when it was collected on the kernel sources that are running now (stamp = hash of csrc/)."""
            try:
                d = json.load(open(os.path.join(ROOT, "profiles", fname)))
            except Exception:
                dropped.append(fname + ": absent")
                return {}
            if d.get("kernel_src_sha16") != src_hash:
                dropped.append("%s: collected on kernel sources %s, running %s" % (fname, d.get("kernel_src_sha16"), src_hash))
                return {}
            return {k: v for k, v in d.get("stages", {}).items()
                    if v.get("images_per_launch") == images_per_launch and v.get("width") == W and v.get("height") == H}
        tj, sj = committed("traffic.json"), committed("sq_counters.json")
        try:
            mix = json.load(open(os.path.join(ROOT, "profiles", "valu_mix.json")))
            if mix.get("kernel_src_sha16") != src_hash:
                dropped.append("valu_mix.json: stale")
                mix = {}
        except Exception:
            mix = {}
        SIMD_CYC_PER_MS = 1024 * 2.4e9 * 1e-3

        def own_traffic(name):
            """HBM bytes per launch of this stage's kernels, from THEIR OWN FETCH_SIZE / WRITE_SIZE counters"""
            keys = {"unstuff": ("unstuff", "unstuff_count"), "dc_scan": ("dc_scan", "dc_sum")}.get(name, (name,))
            vals = [tj[k]["hbm_bytes_per_launch"] for k in keys if k in tj]
            return sum(vals) if vals else None

        per_kernel = {}
        valu_step = 0
        valu_nominal_cycles_step = 0.0
        for name, ms in (stages_iso or stages_ovl):
            if name == "sync_wait":
                continue
            t_ovl = ovl.get(name)
            tr = own_traffic(name)
            ent = {"kernel": kernel_of.get(name, name), "ms_isolated": round(iso[name], 4) if name in iso else None,
                   "ms_overlapped": round(t_ovl, 4) if t_ovl is not None else None, "traffic": tr,
                   # this kernel's own measured HBM bytes over its own duration, as a share of the HBM peak
                   "hbm_frac": round(tr / (t_ovl * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if tr and t_ovl else None,
                   "hbm_frac_isolated": round(tr / (iso[name] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if tr and iso.get(name) else None}
            parts = [(k, v) for k, v in sj.items() if k == name or (name == "unstuff" and k == "unstuff_count") or (name == "dc_scan" and k == "dc_sum")]
            vi = sum(v.get("valu_insts_per_launch", 0) for _, v in parts)
            if vi and iso.get(name):
                # nominal issue cycles of this kernel's own instruction mix (profiles/valu_mix.json: 2 cycles per wave64
                # instruction, 4 for packed fp32, 8 for transcendentals — costs the hardware cannot beat)
                def nominal_of(kernel):  # the DYNAMIC mix where the executed fp32 counters exist (tools/valu_mix.py), else the static one
                    m = mix.get("kernels", {}).get(kernel) or next((v2 for k2, v2 in mix.get("kernels", {}).items() if k2.split("<")[0] == kernel), {})
                    return m.get("nominal_cycles_per_inst_dynamic", m.get("nominal_cycles_per_inst", 2.0))
                nom = sum(v.get("valu_insts_per_launch", 0) * nominal_of(v.get("kernel")) for _, v in parts)
                ent["valu_insts"] = vi
                ent["cycles_per_valu_inst_per_simd_isolated"] = round(SIMD_CYC_PER_MS * iso[name] / vi, 2)
                ent["valu_frac_of_nominal_issue"] = round(nom / (SIMD_CYC_PER_MS * iso[name]), 4)
                valu_step += vi * ns
                valu_nominal_cycles_step += nom * ns
            per_kernel[name] = ent
        achieved = launch_alg_bytes / (dom[1] * 1e-3) / 1e9 if dom[1] > 0 else 0.0
        roof = {"bound": "hbm", "kernel": kernel_of.get(dom[0], dom[0]), "stage": dom[0],
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": per_kernel.get(dom[0], {}).get("traffic"),
                "timing": timing + (": hipEvents around the stage inside the timed region (the launches of %s), other streams' kernels beside it" % ("one of the %d batches of a step, %d caller streams" % (ns, nstreams) if len(timed_batches) == 1 and ns > 1 else "every batch")
                                    if timing == "overlapped" else ": no stage events in the timed region"),
                "kernel_ms": round(dom[1], 4),
                "kernel_ms_isolated": round(iso.get(dom[0], 0.0), 4),
                "frac_isolated": round(launch_alg_bytes / (iso[dom[0]] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if iso.get(dom[0]) else None,
                "algorithmic_bytes_per_launch": int(launch_alg_bytes), "images_per_launch": images_per_launch,
                "launches_per_step": ns,
                "pipeline_frac": round(alg_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "per_kernel": per_kernel}
        if dropped:
            roof["profiles_not_used"] = dropped
        if valu_step:  # what actually bounds the path: fp32/int VALU issue (instruction counts from the committed PMC summary)
            g = valu_step / (ms_step * 1e-3) / 1e9
            roof["valu"] = {"bound": "valu", "achieved": round(g, 1), "peak": VALU_PEAK_NOMINAL_GINST, "unit": "G wave-instr/s",
                            "frac": round(valu_nominal_cycles_step / (SIMD_CYC_PER_MS * ms_step), 4),
                            "frac_meaning": "nominal issue cycles of the step's instructions (each kernel's own EXECUTED mix: packed fp32 from the SQ_INSTS_VALU_ADD/MUL_F32 counters) / SIMD cycles of the step",
                            "wave_insts_per_step": valu_step, "cycles_per_inst_per_simd": round(1024 * 2.4 / g, 2),
                            "cost_table_measured": "profiles/r02_valu_rate.txt", "mix": "profiles/valu_mix.json",
                            "counters": "profiles/sq_counters.json (stamp %s = the kernel sources running)" % src_hash,
                            "scope": "whole step, all kernels, overlapped streams"}
        out = {
            "metric": "Mpixels/s JPEG->RGB (4:2:0 baseline batch)",
            "value": round(value, 1),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "backend": backend,        # process group the ranks met over: "nccl" (= RCCL), "gloo", or "none" for one rank
            "ranks_seen": seen,        # an all-reduce of ones over that group
            "per_rank": per_rank,      # every rank's own time per step (value uses the MAX): a straggler is visible here
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8 entropy/pixel bytes, int16 coefficients, f32 IDCT+colour (reference arithmetic)",
            "data": "synthetic" if args.content == "synthetic" else "one photograph (tests/golden/Lenna.jpg), mirror-tiled and re-encoded",
            "config": {"workload": "batch of %d %s %dx%d 4:2:0 q%d baseline JPEGs per GPU (one global batch of %d "
                                   "partitioned over %d rank(s)), resident in HBM, RGB out to HBM"
                                   % (args.batch, "synthetic" if args.content == "synthetic" else
                                      "photographic (Lenna 512x512 mirror-tiled, optimised Huffman tables)", W, H, args.quality,
                                      n_global, world),
                       "images_per_gpu": len(jpgs), "distinct_images": len(sources), "width": W, "height": H,
                       "scan_bytes_per_image": scan_bytes // max(1, len(jpgs)),
                       "algorithmic_bytes_per_image": alg_bytes // max(1, len(jpgs)), "parallelism": "images sharded, no collectives"},
            "roofline": roof,
            "stage_ms_isolated": {n: round(ms, 4) for n, ms in stages_iso},
            "stage_ms_overlapped": {n: round(ms, 4) for n, ms in stages_ovl},
            "timed_runs": runs,
            "preheat": {"ms": args.preheat_ms, "steps": preheat_steps,
                        "note": "untimed steps before the --warmup steps, until the GPU has left its idle clocks; --preheat-ms 0 turns them off"},
            "spread": {"further_regions_of_%d_steps" % args.steps: [round(total_pixels * args.steps / x / 1e6, 1) for x in spread_dt],
                       "note": "same process, same batches, measured after the reported region; box to box the same build differs by a few per cent more"},
            "parity": parity,
            "parity_vs": sorted(parity_vs),
        }
        if e2e is not None:
            out["end_to_end"] = e2e
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline([sources[s] for s in src_ids], W, H)
        print(json.dumps(out), flush=True)

    for b in batches:
        b.close()
    dec.close()
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
