#!/usr/bin/env python3
"""Stage times (launch alone on the GPU) of 64 x 1920x1080 4:2:0 q85 with and without restart markers: where a restart stream costs."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import hjdtools as T
hjd = importlib.import_module("hls-jpeg-decoder_amd")
dec = hjd.Decoder(0, streams=1)
for name, rst in (("no DRI", 0), ("DRI 120 (one MCU row)", 120), ("DRI 16", 16), ("DRI 1", 1)):
    src = [T.synth_jpeg(1920, 1080, 2000 + s, 85, 2, 2, rst) for s in range(4)]
    b = dec.prepare([src[i % 4] for i in range(64)])
    best = {}
    for _ in range(5):
        for n, ms in b.run_timed():
            best[n] = min(best.get(n, 1e9), ms)
        st = b.finish()
    print("%-24s %s  total %.3f  emulated %d" % (name, {k: round(v, 4) for k, v in best.items()}, sum(best.values()), b.sequential_count()), flush=True)
    b.close()
