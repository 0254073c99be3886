"""Soak run (not collected by pytest): test_random_corpus's property over fresh seeds for as long as asked.

    python3 tests/soak_gpu.py --seconds 480 [--seed N] [--max-side 1600]

Every round draws a batch of random baseline JPEGs (sizes 1 .. max-side, every sampling with hf*vf <= 4, quality 5..100,
smooth to noisy content, no / short / long restart intervals, two or three DQTs, the four Huffman table modes), a random
subsequence size and random tuning knobs, decodes the batch through the C ABI on the GPU and compares every image with the CPU
oracle (run in a process pool): identical bytes wherever the reference's algorithm produces an image, an error status where it
would not terminate or overrun.  Prints one line per round and a summary; exit code 1 on the first difference (the offending
JPEG is written to gpurun_out/soak_fail_*.jpg)."""
import argparse
import importlib
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import hjdtools as T  # noqa: E402

SAMPLINGS = [(1, 1), (2, 1), (1, 2), (2, 2), (4, 1), (1, 4)]
KNOBS = [{}, {"HJD_PHASE_WORDS": "2"}, {"HJD_PHASE_WORDS": "4", "HJD_WRITE_WINDOWS": "2"}, {"HJD_PHASE_WORDS": "8", "HJD_WRITE_WINDOWS": "4"},
         {"HJD_PHASE_WORDS": "32"}, {"HJD_SYNC_OVERLAP": "0"}, {"HJD_SYNC_OVERLAP": "1"}, {"HJD_SYNC_CONC": "0", "HJD_STAGGER": "1"},
         # round 4: sibling trajectories forced on / off, the thumbnails' kernels off one by one
         {"HJD_SIBLINGS": "5"}, {"HJD_SIBLINGS": "2", "HJD_PHASE_WORDS": "8"}, {"HJD_SIBLINGS": "0"}, {"HJD_NO_SEQ_SYNC": "1"},
         {"HJD_SIBLINGS": "11"}, {"HJD_SIBLINGS": "11", "HJD_SYNC_OVERLAP": "1"}, {"HJD_SIBLINGS": "8", "HJD_PHASE_WORDS": "4"},   # all 2 nb entering classes: entry adoption, chain rounds
         {"HJD_NO_SEQ_SYNC": "1", "HJD_NO_SMALL_SYNC": "1"},
         # dirty restart streams: the one-lane emulation instead of the segment-parallel one; segments cut off early (more fall back)
         {"HJD_NO_SEG_EMULATION": "1"}, {"HJD_SEG_CUT": "8"}]


def split_knobs(hjd, knobs):
    """(environment proper, hjd_options.debug_flags): the test switches are options of the decoder (include/hjd.h: HJD_DBG_*)."""
    env = dict(knobs)
    f = 0
    if env.pop("HJD_NO_SMALL_SYNC", None):
        f |= hjd.DBG_NO_SMALL_SYNC
    if env.pop("HJD_NO_SEQ_SYNC", None):
        f |= hjd.DBG_NO_SEQ_SYNC
    if env.pop("HJD_NO_SEG_EMULATION", None):
        f |= hjd.DBG_NO_SEG_EMULATION
    if "HJD_SIBLINGS" in env:
        f |= hjd.dbg_siblings(int(env.pop("HJD_SIBLINGS")))
    if "HJD_SEG_CUT" in env:
        f |= hjd.dbg_seg_cut(int(env.pop("HJD_SEG_CUT")))
    return env, f


def make(args):
    seed, max_side = args
    rng = np.random.default_rng(seed)
    big = rng.random() < 0.15
    w = int(rng.integers(1, (max_side if big else 320) + 1))
    h = int(rng.integers(1, (max_side if big else 320) + 1))
    hf, vf = SAMPLINGS[int(rng.integers(0, len(SAMPLINGS)))]
    q = int(rng.integers(5, 101))
    sigma = float(rng.choice([0.0, 0.5, 3.0, 8.0, 30.0, 80.0]))
    nm = -(-w // (8 * hf)) * -(-h // (8 * vf))
    rst = int(rng.choice([0, 0, 0, 1, int(rng.integers(1, nm + 2)), -(-w // (8 * hf)), 2 * -(-w // (8 * hf))]))
    three, mode = bool(rng.integers(0, 2)), int(rng.choice([T.HUFF_STD, T.HUFF_SIX_PERMUTED, T.HUFF_SIX_OPTIMISED, T.HUFF_FOUR_OPTIMISED]))
    while True:
        try:
            jpg = T.encode_jpeg_ex(T.synth_rgb(w, h, seed, sigma), q, hf, vf, min(rst, 65535), three_dqt=three, huff_mode=mode)
            break
        except AssertionError:  # a coefficient the baseline tables have no code for (noise at quality ~100): milder content
            sigma, q = sigma / 2, max(5, q - 10)
    rc, orgb, _, _, _ = T.oracle_decode(jpg)
    return jpg, rc, (orgb if rc == 0 else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=int(time.time()))
    ap.add_argument("--max-side", type=int, default=1600)
    ap.add_argument("--batch", type=int, default=48)
    ap.add_argument("--workers", type=int, default=min(14, os.cpu_count() or 1))
    a = ap.parse_args()
    T.enc_lib()
    T.oracle_lib()  # built before the pool forks
    hjd = importlib.import_module("hls-jpeg-decoder_amd")
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    t_end = time.time() + a.seconds
    rnd = ok = refused = 0
    rng = np.random.default_rng(a.seed)
    print("soak: seed %d, %.0f s, batches of %d, sides up to %d" % (a.seed, a.seconds, a.batch, a.max_side), flush=True)
    with ProcessPoolExecutor(a.workers) as pool:
        nxt = pool.map(make, [(a.seed * 1000003 + k, a.max_side) for k in range(a.batch)], chunksize=2)
        while True:
            items = list(nxt)
            rnd += 1
            if time.time() < t_end:  # the next round's images and oracle outputs are made while this one decodes
                nxt = pool.map(make, [(a.seed * 1000003 + rnd * 4096 + k, a.max_side) for k in range(a.batch)], chunksize=2)
            else:
                nxt = None
            knobs = KNOBS[int(rng.integers(0, len(KNOBS)))]
            bits = int(rng.choice([0, 0, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384]))
            for k in list(os.environ):
                if k.startswith("HJD_") and k not in ("HJD_BENCH_BACKEND",):
                    del os.environ[k]
            env, flags = split_knobs(hjd, knobs)
            os.environ.update(env)
            streams, launches = int(rng.choice([0, 0, 2, 3])), int(rng.choice([0, 0, 2, 3]))
            dec = hjd.Decoder(0, subseq_bits=bits, streams=streams, sync_launches=launches, debug_flags=flags)
            no_ri = rng.random() < 0.3   # round 5: the HW/SW cut as an unmodified reference host drives it — parsed headers, Ri withheld
            if no_ri:
                jds = []
                for it in items:
                    pst, jd = hjd.parse_header(it[0])
                    jd.restart_mcus = 0
                    jds.append((pst, jd))
                good_idx = [k for k, (pst, _) in enumerate(jds) if pst == 0]
                b = dec.prepare_parsed([jds[k][1] for k in good_idx], keep_alive=[it[0] for it in items])
                b.run()
                bst = b.finish()
                st, outs = [jds[k][0] for k in range(len(items))], [None] * len(items)
                for i, k in enumerate(good_idx):
                    st[k] = bst[i]
                    outs[k] = b.download(i) if bst[i] == 0 else None
                b.close()
                knobs = dict(knobs, no_ri=1)
            else:
                st, outs = dec.decode([it[0] for it in items])
            dec.close()
            for k, (jpg, rc, orgb) in enumerate(items):
                good = (st[k] == 0 and np.array_equal(outs[k], orgb)) if rc == 0 else st[k] != 0
                if not good:
                    p = os.path.join(out_dir, "soak_fail_%d_%d.jpg" % (rnd, k))
                    open(p, "wb").write(jpg)
                    print("soak: DIFFERENCE round %d image %d: oracle rc %d, status %d, knobs %s, bits %d -> %s"
                          % (rnd, k, rc, st[k], knobs, bits, p), flush=True)
                    sys.exit(1)
                ok += rc == 0
                refused += rc != 0
            print("soak: round %d ok (S=%s, streams %d, launches %d, %s): %d identical, %d refused so far"
                  % (rnd, bits or "auto", streams, launches, knobs or "default", ok, refused), flush=True)
            if nxt is None:
                break
    print("soak: %d rounds, %d images identical to the oracle, %d refused by both; no difference" % (rnd, ok, refused), flush=True)


if __name__ == "__main__":
    main()
