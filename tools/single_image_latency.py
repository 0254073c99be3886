"""Device time of small batches of 3840x2160 images (BASELINE config 2 = one image) for fixed subsequence sizes and for
the automatic choice (subseq_bits = 0: hjd_api.cpp::batch_build picks 256..2048 from the batch's scan bytes)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import hjdtools as T
hjd = importlib.import_module("hls-jpeg-decoder_amd")
src = [T.synth_jpeg(3840, 2160, 1000 + s, 85, 2, 2) for s in range(4)]
for n in (1, 2, 4, 8, 16):
    jpgs = [src[i % 4] for i in range(n)]
    for S in (0, 512, 1024, 2048, 4096):
        d = hjd.Decoder(0, subseq_bits=S)
        b = d.prepare(jpgs)
        for _ in range(3):
            b.run()
        assert b.finish() == [0] * n
        best = None
        for _ in range(5):
            st = b.run_timed()
            tot = sum(t for _, t in st)
            if best is None or tot < best[0]:
                best = (tot, st)
        # the same run without the stage events (two events around the whole run on the default stream): what a caller sees
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        whole = None
        for _ in range(8):
            e0.record()
            b.run()
            e1.record()
            e1.synchronize()
            whole = e0.elapsed_time(e1) if whole is None else min(whole, e0.elapsed_time(e1))
        b.finish()
        print("n=%2d S=%4s  total %.3f ms  %s  | one run between two events %.3f ms" % (n, S or "auto", best[0], " ".join("%s=%.3f" % (k, t) for k, t in best[1]), whole), flush=True)
        b.close(); d.close()
