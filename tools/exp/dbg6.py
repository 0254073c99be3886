import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import hjdtools as T
hjd = importlib.import_module("hls-jpeg-decoder_amd")
sizes = [(3840, 2160), (1920, 1080), (1280, 720), (1000, 700), (777, 333), (640, 480), (512, 512), (250, 187)]
samp = [(1, 1), (2, 1), (2, 2), (1, 1), (2, 1), (2, 2), (1, 2), (4, 1)]
d = hjd.Decoder(0)
for k in range(1, 32, 2):
    w, h = sizes[k % 8]; hf, vf = samp[(k // 2) % 8]; mcux = -(-w // (8 * hf))
    rst = [mcux, 4 * mcux, 2 * mcux, 16][(k // 8) % 4]
    j = T.synth_jpeg(w, h, 9000 + k, 85, hf, vf, rst, 7.0)
    orc, orgb, _, _, st = T.oracle_decode(j)
    b = d.prepare([j]); b.run(); s = b.finish()
    print(k, (w, h), (hf, vf), rst, "oracle dirty", st.resyncs_dirty, "gpu status", s, "emulated", b.sequential_count(), "equal", bool(s[0] == 0 and np.array_equal(b.download(0), orgb)))
    b.close()
