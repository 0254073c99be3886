import os, sys, importlib, hashlib
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import hjdtools as T
hjd = importlib.import_module("hls-jpeg-decoder_amd")
d = "tests/golden"
names = ["Lenna.jpg", "syn_250x187_444_q85.jpg"]
jpgs = [open(os.path.join(d, n), "rb").read() for n in names]
os.environ["HJD_VERBOSE"] = "1"
os.environ["HJD_MAX_RESYNC"] = "12"
for bits in (256, 64):
    for ov in (-1, 0):
        dd = hjd.Decoder(0, subseq_bits=bits, sync_launches=1, sync_overlap=ov)
        b = dd.prepare(jpgs)
        b.run()
        try:
            st = b.finish()
            print("bits", bits, "overlap", ov, "status", st, "resyncs", b.resync_count(), flush=True)
        except Exception as e:
            print("bits", bits, "overlap", ov, "FAILED", e, flush=True)
        b.close(); dd.close()
