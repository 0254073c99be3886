"""Device time of ONE 3840x2160 image (BASELINE config 2) for several subsequence sizes."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import hjdtools as T
hjd = importlib.import_module("hls-jpeg-decoder_amd")
j = T.synth_jpeg(3840, 2160, 1000, 85, 2, 2)
for S in (128, 256, 512, 1024, 2048):
    d = hjd.Decoder(0, subseq_bits=S)
    b = d.prepare([j])
    for _ in range(3):
        b.run()
    assert b.finish() == [0]
    best = None
    for _ in range(5):
        st = b.run_timed()
        tot = sum(t for _, t in st)
        if best is None or tot < best[0]:
            best = (tot, st)
    print("S=%4d  total %.3f ms  %s" % (S, best[0], " ".join("%s=%.3f" % (n, t) for n, t in best[1])))
    b.close(); d.close()
