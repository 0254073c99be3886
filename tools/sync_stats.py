"""Diagnostics: how many refinement iterations / decodes the synchronisation needs per image."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import hjdtools as T
hjd = importlib.import_module("hls-jpeg-decoder_amd")
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
j = T.synth_jpeg(W, H, 1000, 85, 2, 2)
for S in (256, 512, 1024, 2048, 4096):
    d = hjd.Decoder(0, subseq_bits=S)
    b = d.prepare([j]); b.run(); st = b.finish()
    s = b.sync_stats(0)
    nwg = (s["nsub"] + 503) // 504
    print("S=%d status=%s nsub=%d nwg=%d decodes/sub=%.2f refinement rounds per workgroup=%.1f max=%d" % (
        S, st, s["nsub"], nwg, s["decodes"] / s["nsub"], s["iters"] / nwg, s["iters_max"]))
    b.close(); d.close()
