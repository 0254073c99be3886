#!/usr/bin/env python3
"""Per-image stage times of BASELINE config 5's 32 distinct images (each alone on the GPU, one launch): which images make the
synchronisation kernel of the mixed batch slow.  Prints one line per image."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import hjdtools as T
hjd = importlib.import_module("hls-jpeg-decoder_amd")
hostile = "--hostile" in sys.argv
sizes = [(3840, 2160), (1920, 1080), (1280, 720), (1000, 700), (777, 333), (640, 480), (512, 512), (250, 187)]
samp = [(1, 1), (2, 1), (2, 2), (1, 1), (2, 1), (2, 2), (1, 2), (4, 1)]
S = int(next((a.split("=")[1] for a in sys.argv if a.startswith("S=")), "4096"))
dec = hjd.Decoder(0, subseq_bits=S, debug_flags=hjd.dbg_siblings(0))  # the batch's subsequence size, no sibling trajectories (a batch gets none)
for k in range(32):
    w, h = sizes[k % 8]
    hf, vf = samp[(k // 2) % 8]
    mcux = -(-w // (8 * hf))
    rst = 0 if k % 2 == 0 else ([mcux, 4 * mcux, 7, 1] if hostile else [mcux, 4 * mcux, 2 * mcux, 16])[(k // 8) % 4]
    j = T.synth_jpeg(w, h, 9000 + k, 85, hf, vf, rst, 12.0 if (rst == 16 and not hostile) else 7.0)
    b = dec.prepare([j])
    best = {}
    for _ in range(4):
        for n, ms in b.run_timed():
            best[n] = min(best.get(n, 1e9), ms)
        b.finish()
    st = b.sync_stats(0)
    pc = b.path_counts()
    print("%2d %4dx%-4d %dx%d rst %5d scan %7d B  sync %.3f write %.3f idct %.3f unstuff %.3f  rounds max %2d decodes/sub %.2f nsub %d %s" %
          (k, w, h, hf, vf, rst, len(j), best.get("huff_sync", 0), best.get("huff_write", 0), best.get("idct_color", 0), best.get("unstuff", 0),
           st["iters_max"], st["decodes"] / max(1, st["nsub"]), st["nsub"], "small" if pc["sync_small"] else ("seq" if pc["sync_seq"] else "")), flush=True)
    b.close()
