"""Wall time of the reference's own entry point for ONE image (BASELINE config 2): hjd_convert_jpg_file (ConvertJpgFile,
openjpg.cpp:593-684) on a 3840x2160 4:2:0 q85 file — first call (allocates the re-usable workspace) and warm calls — beside
the compiled reference (oracle/_ref/ref_big) on the same file, and the device time of the same image (hjd_batch_run_timed)."""
import importlib, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import hashlib
import torch  # noqa
import hjdtools as T
hjd = importlib.import_module("hls-jpeg-decoder_amd")
W, H = 3840, 2160
j = T.synth_jpeg(W, H, 1000, 85, 2, 2)
with tempfile.TemporaryDirectory() as td:
    jp, bp, rp = os.path.join(td, "a.jpg"), os.path.join(td, "a.bmp"), os.path.join(td, "r.bmp")
    open(jp, "wb").write(j)
    ts = []
    for k in range(8):
        t0 = time.perf_counter(); rc = hjd.convert_jpg_file(jp, bp); ts.append((time.perf_counter() - t0) * 1e3)
        assert rc == 1
    bmp = open(bp, "rb").read()
    print("hjd_convert_jpg_file %dx%d: first call %.1f ms, warm calls %s ms (file read + header walk + H2D + kernels + D2H + file write)"
          % (W, H, ts[0], " ".join("%.2f" % t for t in ts[1:])))
    print("  = %.1f Mpixel/s warm (best)" % (W * H / min(ts[1:]) / 1e3))
    if T.ref_binary(big=True):
        t0 = time.perf_counter(); rc = T.ref_convert(jp, rp, big=True); tr = (time.perf_counter() - t0) * 1e3
        same = open(rp, "rb").read()[54:] == bmp[54:]  # the reference's BMP header pads differ only in its two garbage bytes per row: none at this width
        print("reference binary (oracle/_ref/ref_big), same file: %.0f ms, pixel bytes identical: %s" % (tr, same))
    ent = T.large_golden(W, H, 1000, 85, 2, 2, j)
    if ent:
        print("BMP sha256 equals the reference-produced hash:", hashlib.sha256(bmp).hexdigest() == ent.get("bmp_sha256"))
d = hjd.Decoder(0)
b = d.prepare([j])
for _ in range(3):
    b.run(); b.finish()
best = None
for _ in range(5):
    st = b.run_timed()
    tot = sum(ms for _, ms in st)
    if best is None or tot < best[0]:
        best = (tot, st)
print("device time, one image: %.3f ms  " % best[0] + " ".join("%s=%.3f" % x for x in best[1]))
b.close(); d.close()
