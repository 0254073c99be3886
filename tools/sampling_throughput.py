"""Device throughput of the whole path for other chroma samplings (BASELINE config 5 material): 64 x 1920x1080 per sampling, one
batch on one stream (library defaults), RGB to HBM.  Not the headline metric; a check that no sampling has a slow path."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import hjdtools as T
hjd = importlib.import_module("hls-jpeg-decoder_amd")
dec = hjd.Decoder(0)
W, H = 1920, 1080
for name, hf, vf, rst in (("4:2:0", 2, 2, 0), ("4:2:2", 2, 1, 0), ("4:4:0", 1, 2, 0), ("4:4:4", 1, 1, 0), ("4:1:1", 4, 1, 0), ("4:2:0 + DRI 120", 2, 2, 120)):
    src = [T.synth_jpeg(W, H, 2000 + s, 85, hf, vf, rst) for s in range(4)]
    b = dec.prepare([src[i % 4] for i in range(64)])
    for _ in range(3):
        b.run()
    assert b.finish() == [0] * 64
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        b.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    assert b.finish() == [0] * 64
    print("%-16s %6.1f Gpixel/s  (%.3f ms per 64 images, %.2f MB of scan)" % (name, 64 * W * H / dt / 1e9, dt * 1e3, sum(len(s) for s in src) / 4 * 64 / 1e6), flush=True)
    b.close()
