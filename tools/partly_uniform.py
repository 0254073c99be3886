"""Partly uniform 4K content (objects on a white background, flat patches, bands), standard and optimised tables."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch  # noqa
import hjdtools as T
hjd = importlib.import_module("hls-jpeg-decoder_amd")
W, H = 3840, 2160
tex = T.synth_rgb(W, H, 11, 8.0)
cases = {}
a = np.full((H, W, 3), 255, np.uint8); a[H // 4:3 * H // 4, W // 4:3 * W // 4] = tex[H // 4:3 * H // 4, W // 4:3 * W // 4]; cases["object on white"] = a
a = np.zeros((H, W, 3), np.uint8); a[H // 3:2 * H // 3, W // 8:W // 2] = tex[H // 3:2 * H // 3, W // 8:W // 2]; cases["object on black"] = a
rng = np.random.default_rng(3); a = np.zeros((H, W, 3), np.uint8)
for y in range(0, H, H // 6):
    for x in range(0, W, W // 8):
        a[y:y + H // 6, x:x + W // 8] = rng.integers(0, 256, 3)
cases["flat patches"] = a
a = np.zeros((H, W, 3), np.uint8)
for k, y in enumerate(range(0, H, 120)): a[y:y + 120] = (k * 37) % 256
cases["bands"] = a
a = tex.copy(); a[:H // 5] = 255; a[-H // 5:] = 255; cases["white margins"] = a
d = hjd.Decoder(0)
for name, rgb in cases.items():
    for mode, mn in ((T.HUFF_STD, "std"), (T.HUFF_FOUR_OPTIMISED, "opt")):
        for hf, vf in ((2, 2), (1, 1)):
            j = T.encode_jpeg_ex(rgb, 85, hf, vf, huff_mode=mode)
            b = d.prepare([j]); b.run(); st = b.finish()
            t0 = time.perf_counter(); b.run(); st = b.finish(); t = time.perf_counter() - t0
            stg = dict(b.run_timed()); b.finish()
            print("%-16s %s %dx%d scan %7d B status %s wall %6.2f ms sync %.3f resyncs %d %s" % (name, mn, hf, vf, len(j), st, t * 1e3, stg["huff_sync"], b.resync_count(), b.sync_stats(0)), flush=True)
            b.close()
