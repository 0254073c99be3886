"""Uniform frames: periodic bit streams in which a speculative decode may never fall into step (open).
Device time, host re-synchronisations and status of one 3840x2160 image per case."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch  # noqa
import hjdtools as T
hjd = importlib.import_module("hls-jpeg-decoder_amd")
W, H = 3840, 2160
white = np.full((H, W, 3), 255, np.uint8)
half = T.synth_rgb(W, H, 5, 6.0); half[:H // 2] = 200
bars = np.zeros((H, W, 3), np.uint8); bars[:, ::64] = 255  # vertical lines every 64 pixels: periodic per 4 MCUs
cases = {"white": white, "black": np.zeros((H, W, 3), np.uint8), "grey": np.full((H, W, 3), 200, np.uint8), "half flat": half, "bars": bars, "noise": T.synth_rgb(W, H, 6, 6.0)}
cases = {k: (v, 2, 2) for k, v in cases.items()}
cases["white 4:4:4"] = (white, 1, 1)
cases["white 4:2:2"] = (white, 2, 1)
nz = np.clip(T.synth_rgb(W, H, 7, 0.5).astype(np.int16) - 118, 0, 255).astype(np.uint8)  # nearly black, a little texture
cases["dark, faint texture"] = (nz, 2, 2)
cases = {k: v + (T.HUFF_STD,) for k, v in cases.items()}
cases["white, optimised tables"] = (white, 2, 2, T.HUFF_FOUR_OPTIMISED)        # one-bit codes: MCUs of 12 bits, a period of three words
cases["white 4:4:4, optimised"] = (white, 1, 1, T.HUFF_SIX_OPTIMISED)
for name, (rgb, hf, vf, mode) in cases.items():
    j = T.encode_jpeg_ex(rgb, 85, hf, vf, huff_mode=mode)
    for S in (0,):
        d = hjd.Decoder(0, subseq_bits=S)
        b = d.prepare([j])
        t0 = time.perf_counter(); b.run(); st = b.finish(); t1 = time.perf_counter() - t0
        t0 = time.perf_counter(); b.run(); st = b.finish(); t2 = time.perf_counter() - t0
        stg = b.run_timed(); st2 = b.finish()
        out = b.download(0) if st == [0] else None
        ok = out is not None and np.array_equal(out, hjd.__dict__.get("_none", out))
        print("%-26s S=%-5s scan %7d B  status %s  wall %.2f / %.2f ms  resyncs %d  stages %s" % (
            name, S or "auto", len(j), st, t1 * 1e3, t2 * 1e3, b.resync_count(), " ".join("%s=%.3f" % (k, t) for k, t in stg)), flush=True)
        b.close(); d.close()
