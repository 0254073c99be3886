"""Throughput on batches of small images (per-image fixed costs: one synchronisation workgroup per image at least)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch  # noqa
import hjdtools as T
hjd = importlib.import_module("hls-jpeg-decoder_amd")
for S in (0,) if len(sys.argv) < 2 else (0, 128, 256, 512, 1024, 2048):  # any argument: sweep the subsequence size
  d = hjd.Decoder(0, subseq_bits=S)
  print("S =", S or "auto")
  for (w, h, n) in ((64, 64, 8192), (128, 128, 8192), (256, 256, 4096), (512, 512, 2048), (1024, 768, 512)):
      src = [T.synth_jpeg(w, h, 300 + k, 85, 2, 2) for k in range(16)]
      jpgs = [src[k % 16] for k in range(n)]
      b = d.prepare(jpgs)
      for _ in range(2):
          b.run()
      assert all(s == 0 for s in b.finish())
      torch.cuda.synchronize(); t0 = time.perf_counter()
      for _ in range(5):
          b.run()
      st = b.finish(); dt = (time.perf_counter() - t0) / 5
      stg = b.run_timed(); b.finish()
      print("%4dx%-4d x %5d: %.3f ms per batch = %.1f Gpixel/s, %.2f M images/s  %s" % (w, h, n, dt * 1e3, w * h * n / dt / 1e9, n / dt / 1e6, " ".join("%s=%.3f" % kv for kv in stg)), flush=True)
      b.close()
