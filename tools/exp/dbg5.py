import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import hjdtools as T
hjd = importlib.import_module("hls-jpeg-decoder_amd")
spec = (1920, 1080, 9017, 85, 1, 1, 7, 7.0)
j = T.synth_jpeg(*spec)
orc, orgb, ocoef, _, ost = T.oracle_decode(j, want_coefs=True)
print("oracle", orc, ost.resyncs_block_start, ost.resyncs_mid_block, ost.resyncs_dirty)
for kw in ({}, {"subseq_bits": 2048}, {"subseq_bits": 256}, {"dirty_mode": 1}):
    for others in (0, 5):
        d = hjd.Decoder(0, **kw)
        extra = [T.synth_jpeg(640, 480, 77 + q, 85, 2, 2) for q in range(others)]
        b = d.prepare(extra + [j])
        b.run(); st = b.finish()
        k = others
        msg = "opts %s others %d status %s seq %d resync %d" % (kw, others, st[k], b.sequential_count(), b.resync_count())
        if st[k] == 0:
            got = b.download(k)
            bad = (got != orgb).any(axis=2)
            co = b.download_coefs(k, ocoef.shape[0])
            cb = np.argwhere((co != ocoef).any(axis=1)).ravel()
            msg += " wrong pixels %d wrong blocks %d first %s" % (bad.sum(), len(cb), cb[:6].tolist())
            if len(cb):
                q = cb[0]; print("   block", q, "got", co[q][:12].tolist(), "want", ocoef[q][:12].tolist())
        print(msg)
        b.close(); d.close()
