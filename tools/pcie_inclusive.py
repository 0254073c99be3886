"""End-to-end (host buffers in, host buffers out) rate of the synchronous convenience path, for DESIGN.md §5:
parse + H2D + all kernels + D2H per batch.  Not the metric; run on the GPU box."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import hjdtools as T
hjd = importlib.import_module("hls-jpeg-decoder_amd")
W, H, N = 3840, 2160, 16
jp = [T.synth_jpeg(W, H, 1000 + k, 85, 2, 2) for k in range(4)]
jpgs = [jp[i % 4] for i in range(N)]
d = hjd.Decoder(0)
d.decode(jpgs[:2])
t0 = time.perf_counter()
st, outs = d.decode(jpgs)
dt = time.perf_counter() - t0
assert all(s == 0 for s in st)
print("host-to-host: %d x %dx%d in %.1f ms = %.0f Mpixels/s (pageable host memory, synchronous copies)" % (N, W, H, dt * 1e3, N * W * H / dt / 1e6))
d.close()
