#!/usr/bin/env python3
"""Generate tests/golden/: small JPEG fixtures + what the COMPILED REFERENCE emits for them.

Run in the build container only (needs /root/reference and `make -C oracle ref`).
For every fixture the unmodified reference (oracle/_ref/ref_small, shipped capacity
512x512 / 105 KB) is run under a timeout; where the input exceeds the shipped capacity the
enlarged-capacity build (ref_big) is used and the manifest says so.  Recorded per fixture:

  bmp_sha256         sha256 of the reference's BMP bytes exactly as written
  bmp_sha256_masked  same, with BMP row-pad bytes 2 and 3 zeroed: the reference writes pad
                     bytes by fwrite(&pad, n, 1) from a ONE-byte object (openjpg.cpp:563-567),
                     so pad bytes beyond the first are stack garbage, not part of the contract
  rgb_sha256         sha256 of the top-down RGB payload reconstructed from the BMP
  status             "ok" | "unsupported" | "ref_hangs"
  restart            "none" | "clean" | "dirty"   (judged by the oracle's trace)

The JPEG inputs themselves are committed (encoder output is version dependent): a fixture whose
file already exists under tests/golden/ is NOT regenerated, only re-run through the reference.

Round 2 added: three DQTs with Cb's table != Cr's (the reference dequantises Cb with Cr's table,
loadjpg.cpp:984), six distinct Huffman tables (selectors 0/1/2 for Y/Cb/Cr), a crafted block set
whose dequantised products wrap int16 (loadjpg.cpp:144-152) and saturate Clamp (:83-91), and a
scan of 63 x (+-1) blocks coded with a 1-bit (run 0, size 1) code (2 scan bits per coefficient).
"""
import hashlib
import io
import json
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hjdtools as T  # noqa: E402

GOLD = os.path.join(T.ROOT, "tests", "golden")
TMP = "/tmp/hjd_golden"


def bmp_to_rgb_masked(bmp, w, h):
    pad = (4 - (3 * w) % 4) % 4
    stride = 3 * w + pad
    body = np.frombuffer(bmp, np.uint8)[54:].reshape(h, stride).copy()
    if pad > 1:
        body[:, 3 * w + 1:] = 0
    masked = bmp[:54] + body.tobytes()
    rgb = body[::-1, :3 * w].reshape(h, w, 3)[:, :, ::-1]
    return masked, np.ascontiguousarray(rgb)


def run_ref(jpg_bytes, name):
    os.makedirs(TMP, exist_ok=True)
    jp, bp = os.path.join(TMP, name), os.path.join(TMP, name + ".bmp")
    open(jp, "wb").write(jpg_bytes)
    if os.path.exists(bp):
        os.remove(bp)
    try:
        w, h, hf, vf = T.jpeg_dims(jpg_bytes)
    except Exception:
        w = h = 0
    small_ok = 0 < w <= 512 and 0 < h <= 512 and len(jpg_bytes) <= 105000
    try:
        rc = T.ref_convert(jp, bp, big=not small_ok, timeout=30)
    except subprocess.TimeoutExpired:
        return "hang", None, "small" if small_ok else "big"
    if rc != 0 or not os.path.exists(bp):
        return "fail", None, "small" if small_ok else "big"
    return "ok", open(bp, "rb").read(), "small" if small_ok else "big"


def pil_jpeg(rgb, **kw):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(rgb).save(b, "JPEG", **kw)
    return b.getvalue()


def main():
    os.makedirs(GOLD, exist_ok=True)
    fixtures = []  # (name, bytes, expect)

    lenna = open("/root/reference/data/Lenna.jpg", "rb").read()
    fixtures.append(("Lenna.jpg", lenna, "ok"))

    # own encoder (standard Annex-K tables)
    for name, w, h, q, hf, vf, rst, sig, seed in [
        ("syn_256x192_420_q85.jpg", 256, 192, 85, 2, 2, 0, 6.0, 1),
        ("syn_250x187_420_q85.jpg", 250, 187, 85, 2, 2, 0, 6.0, 2),
        ("syn_250x187_422_q85.jpg", 250, 187, 85, 2, 1, 0, 6.0, 3),
        ("syn_250x187_440_q85.jpg", 250, 187, 85, 1, 2, 0, 6.0, 4),
        ("syn_250x187_444_q85.jpg", 250, 187, 85, 1, 1, 0, 6.0, 5),
        ("syn_128x96_420_q50.jpg", 128, 96, 50, 2, 2, 0, 6.0, 6),
        ("syn_96x64_420_q100.jpg", 96, 64, 100, 2, 2, 0, 12.0, 7),
        ("syn_64x48_411_q85.jpg", 64, 48, 85, 4, 1, 0, 6.0, 8),
        ("syn_17x9_420_q85.jpg", 17, 9, 85, 2, 2, 0, 6.0, 9),
        ("syn_1x1_444_q85.jpg", 1, 1, 85, 1, 1, 0, 6.0, 10),
        ("syn_8x8_420_q85.jpg", 8, 8, 85, 2, 2, 0, 6.0, 11),
        ("syn_256x192_420_q85_rst16.jpg", 256, 192, 85, 2, 2, 16, 6.0, 12),
        ("syn_250x187_444_q85_rst4.jpg", 250, 187, 85, 1, 1, 4, 6.0, 13),
        ("syn_250x187_422_q85_rst1.jpg", 250, 187, 85, 2, 1, 1, 8.0, 14),
        ("syn_128x128_420_q20_smooth.jpg", 128, 128, 20, 2, 2, 0, 0.0, 15),
    ]:
        fixtures.append((name, T.synth_jpeg(w, h, seed, q, hf, vf, rst, sig), "ok"))
    flat = np.full((64, 64, 3), 77, np.uint8)
    fixtures.append(("flat_64x64_444_rst1_dirty.jpg", T.encode_jpeg(flat, 85, 1, 1, 1), "ok"))
    fixtures.append(("flat_64x64_420.jpg", T.encode_jpeg(flat, 85, 2, 2, 0), "ok"))

    # round 2: Cb/Cr quantised differently, six Huffman tables, int16-wrap extremes, 2-bit coefficients
    img2 = T.synth_rgb(250, 187, 31, 6.0)
    fixtures.append(("syn_250x187_420_q85_3dqt.jpg", T.encode_jpeg_ex(img2, 85, 2, 2, 0, True, T.HUFF_STD), "ok"))
    fixtures.append(("syn_250x187_422_q85_3dqt_6huff.jpg", T.encode_jpeg_ex(img2, 85, 2, 1, 0, True, T.HUFF_SIX_OPTIMISED), "ok"))
    fixtures.append(("syn_96x64_444_q60_3dqt_6perm.jpg",
                     T.encode_jpeg_ex(T.synth_rgb(96, 64, 32, 8.0), 60, 1, 1, 0, True, T.HUFF_SIX_PERMUTED), "ok"))
    fixtures.append(("syn_250x187_420_q85_4opt_rst5.jpg", T.encode_jpeg_ex(img2, 85, 2, 2, 5, False, T.HUFF_FOUR_OPTIMISED), "ok"))
    rng = np.random.default_rng(11)
    nb = 16 * 6
    co = np.zeros((nb, 64), np.int16)
    mags = [1, 2, 127, 128, 129, 255, 256, 257, 300, 1000, 1023, 1024, 4095, 8191, 16384, 20000, 32767]
    for b in range(nb):
        k = rng.integers(1, 24)
        idx = rng.choice(np.arange(1, 64), k, replace=False)
        co[b, idx] = rng.choice(mags, k) * rng.choice([-1, 1], k)
    co[:, 0] = rng.integers(-32000, 32001, nb).astype(np.int16)
    qt = np.full((3, 64), 255, np.uint8)
    qt[2, ::3] = 1
    qt[2, 1::7] = 129
    qt[0, 5::4] = 128
    qt[1, :] = 7  # never used by the reference: Cb is dequantised with Cr's table
    fixtures.append(("ext_64x64_420_int16wrap.jpg",
                     T.encode_jpeg_ex(None, 85, 2, 2, 0, True, T.HUFF_SIX_OPTIMISED, qtab=qt, coefs=co, size=(64, 64)), "ok"))
    nb = 8 * 6 * 3
    co = np.zeros((nb, 64), np.int16)
    co[:, 1:] = rng.choice([-1, 1], (nb, 63))
    co[:, 0] = rng.integers(-200, 200, nb)
    qt = np.full((3, 64), 3, np.uint8)
    qt[2] = 5
    fixtures.append(("ext_64x48_444_dense_ones.jpg",
                     T.encode_jpeg_ex(None, 85, 1, 1, 0, True, T.HUFF_SIX_OPTIMISED, qtab=qt, coefs=co, size=(64, 48)), "ok"))

    # Pillow / libjpeg-turbo (optimised Huffman tables, other DQT scalings)
    img = T.synth_rgb(250, 187, 21, 6.0)
    fixtures.append(("pil_250x187_420_q85_opt.jpg", pil_jpeg(img, quality=85, subsampling="4:2:0", optimize=True), "ok"))
    fixtures.append(("pil_250x187_444_q95_opt.jpg", pil_jpeg(img, quality=95, subsampling="4:4:4", optimize=True), "ok"))
    fixtures.append(("pil_250x187_422_q50.jpg", pil_jpeg(img, quality=50, subsampling="4:2:2"), "ok"))
    fixtures.append(("pil_256x192_420_q85_std.jpg", pil_jpeg(T.synth_rgb(256, 192, 22, 6.0), quality=85, subsampling="4:2:0"), "ok"))
    fixtures.append(("pil_progressive.jpg", pil_jpeg(img[:64, :64], quality=85, progressive=True), "unsupported"))
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(img[:64, :64, 0]).save(b, "JPEG", quality=85)
    fixtures.append(("pil_gray.jpg", b.getvalue(), "unsupported"))

    manifest = {}
    for name, data, expect in fixtures:
        path = os.path.join(GOLD, name)
        if os.path.exists(path):
            data = open(path, "rb").read()  # committed bytes are the fixture
        else:
            open(path, "wb").write(data)
        ent = {"bytes": len(data), "sha256": hashlib.sha256(data).hexdigest()}
        if expect == "unsupported":
            ent["status"] = "unsupported"  # reference crashes / prints garbage here: no golden output
            manifest[name] = ent
            print(name, ent["status"])
            continue
        w, h, hf, vf = T.jpeg_dims(data)
        st, bmp, which = run_ref(data, name)
        ent.update({"width": w, "height": h, "hf": hf, "vf": vf, "ref_build": which})
        if st == "hang":
            ent["status"] = "ref_hangs"
        elif st != "ok":
            ent["status"] = "ref_failed"
        else:
            masked, rgb = bmp_to_rgb_masked(bmp, w, h)
            ent["status"] = "ok"
            ent["bmp_sha256"] = hashlib.sha256(bmp).hexdigest()
            ent["bmp_sha256_masked"] = hashlib.sha256(masked).hexdigest()
            ent["rgb_sha256"] = hashlib.sha256(rgb.tobytes()).hexdigest()
            ent["bmp_bytes"] = len(bmp)
        rc, _, _, _, ost = T.oracle_decode(data)
        has_dri = b"\xff\xdd\x00\x04" in data[:1024]
        ent["restart"] = "none" if not has_dri else ("clean" if ost.resyncs_dirty == 0 else "dirty")
        ent["final_stream_index"] = int(ost.final_stream_index)
        manifest[name] = ent
        print(name, ent["status"], ent.get("restart"), which, ent.get("bmp_sha256", "")[:12])
    json.dump(manifest, open(os.path.join(GOLD, "manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
