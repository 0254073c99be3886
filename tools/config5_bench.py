"""BASELINE config 5: mixed 4:4:4 / 4:2:2 / 4:2:0 subsampling with restart-interval markers, mixed sizes, ONE batch on one MI355X.
128 images (sizes 3840x2160 ... 250x187, half of them with DRI), resident in HBM, RGB out to HBM; every distinct image is
checked against the oracle (bit-exact) after the timed steps.  Prints one JSON line (-> profiles/r03_config5_bench.json)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import hjdtools as T
hjd = importlib.import_module("hls-jpeg-decoder_amd")
sizes = [(3840, 2160), (1920, 1080), (1280, 720), (1000, 700), (777, 333), (640, 480), (512, 512), (250, 187)]
samp = [(1, 1), (2, 1), (2, 2), (1, 1), (2, 1), (2, 2), (1, 2), (4, 1)]
specs = []
for k in range(32):  # 32 distinct images: size x sampling x (no DRI | DRI)
    w, h = sizes[k % 8]
    hf, vf = samp[(k // 2) % 8]
    mcux = -(-w // (8 * hf))
    # restart intervals an encoder would choose (one MCU row, four rows, two rows, 16 MCUs); with the argument "hostile" also 7 MCUs
    # and every MCU, where the reference fires its resync checks early on tiny chroma blocks and the images
    # go through the sequential emulation
    rst = 0 if k % 2 == 0 else ([mcux, 4 * mcux, 7, 1] if "hostile" in sys.argv else [mcux, 4 * mcux, 2 * mcux, 16])[(k // 8) % 4]
    # busy content (intervals then end in blocks of 8 bits and more — the clean-resync domain); with 16-MCU intervals
    # the smooth-chroma image 27 needs sigma 12 for that, at sigma 7 the reference fires its resync check early on it twice
    # and the image goes through the sequential emulation ("hostile" keeps sigma 7 everywhere)
    specs.append((w, h, 9000 + k, 85, hf, vf, rst, 12.0 if (rst == 16 and "hostile" not in sys.argv) else 7.0))
src = [T.synth_jpeg(*s) for s in specs]
jpgs = [src[k % 32] for k in range(128)]
pixels = sum(specs[k % 32][0] * specs[k % 32][1] for k in range(128))
NSTREAMS = next((int(a.split('=')[1]) for a in sys.argv if a.startswith('streams=')), 0)  # streams=N: the batch split over N streams inside the library (hjd_options.streams)
dec = hjd.Decoder(0, streams=NSTREAMS)
b = dec.prepare(jpgs)
assert all(s == 0 for s in b.parse_status)
for _ in range(3):
    b.run()
st = b.finish()
STEPS = 20
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    b.run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / STEPS
st = b.finish()
ok = 0
for k in range(32):
    orc, orgb, _, _, _ = T.oracle_decode(src[k])
    if orc == 0:
        assert st[k] == 0, (specs[k], st[k])
        got = b.download(k)
        assert np.array_equal(got, orgb), (specs[k], int((got != orgb).any(axis=2).sum()), np.argwhere((got != orgb).any(axis=2))[:3].tolist())
        ok += 1
    else:
        assert st[k] != 0, specs[k]
px, alg, scan = b.totals()
print(json.dumps({"metric": "Mpixels/s JPEG->RGB", "config": {"workload": "BASELINE config 5: one batch of 128 images, 8 sizes 3840x2160..250x187, "
                  "4:4:4 / 4:2:2 / 4:2:0 / 4:4:0 / 4:1:1, half of them with DRI (intervals of %s), resident in HBM, RGB to HBM" % ("one MCU row, four rows, 7 MCUs, 1 MCU" if "hostile" in sys.argv else "one MCU row, four rows, two rows, 16 MCUs"),
                  "images": 128, "distinct_images": 32, "internal_streams": NSTREAMS, "with_dri": sum(1 for s in specs if s[6]) * 4, "scan_bytes": scan},
                  "value": round(pixels / dt / 1e6, 1), "unit": "Mpixels/s", "ms_per_step": round(dt * 1e3, 4), "steps": STEPS,
                  "pipeline_frac_of_hbm_peak": round(alg / dt / 8e12, 5), "sequential_emulations": b.sequential_count(), "resyncs": b.resync_count(),
                  "parity": "bit-exact vs oracle on %d of 32 distinct images (the others are refused by both)" % ok}))
b.close(); dec.close()
