"""Diagnostics: where a k_sync workgroup's time goes (HJD_SYNC_PROF=1: shader-clock stamps per workgroup)."""
import ctypes, importlib, os, sys
os.environ["HJD_SYNC_PROF"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch  # noqa
import hjdtools as T
hjd = importlib.import_module("hls-jpeg-decoder_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
jpgs = [T.synth_jpeg(3840, 2160, 1000 + (k % 4), 85, 2, 2) for k in range(n)]
d = hjd.Decoder(0, streams=1)
b = d.prepare(jpgs)
for _ in range(3):
    b.run(); st = b.finish()
L = hjd.lib()
L.hjd_debug_sync_profile.restype = ctypes.c_int
L.hjd_debug_sync_profile.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros(1 << 22, np.uint64)
nwg = L.hjd_debug_sync_profile(b._h, buf.ctypes.data, buf.size)
a = buf[:nwg * 32].reshape(nwg, 16, 2).astype(np.int64)
t0 = a[:, 0, 0].min()
print("workgroups", nwg, "kernel span (cycles @100MHz? see below)", a[:, :, 0].max() - t0)
durs = []
for w in range(nwg):
    ev = [(int(a[w, k, 0] - a[w, 0, 0]), int(a[w, k, 1])) for k in range(16) if a[w, k, 0]]
    durs.append((ev[-1][0], w, ev))
durs.sort()
for tag, (tot, w, ev) in (("fastest", durs[0]), ("median", durs[len(durs) // 2]), ("slowest", durs[-1]), ("2nd slowest", durs[-2])):
    print(tag, "wg", w, "start offset", int(a[w, 0, 0] - t0), "total", tot)
    for t, x in ev:
        print("    %9d  %s" % (t, hex(x)))
import collections
print("mean per stage (stamps in order):", np.mean([[e[0] for e in ev[:6]] for _, _, ev in durs if len(ev) >= 6], axis=0))
