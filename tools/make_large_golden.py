#!/usr/bin/env python3
"""tests/golden/large_manifest.json: what the COMPILED REFERENCE emits for the full-size inputs of
bench.py and of the full-size GPU tests.

Build container only (needs oracle/_ref/ref_big = the unmodified reference sources with the three
capacity #defines of loadjpg.h:55-57 enlarged, see oracle/Makefile).  The inputs are not committed
(1.3 MB each): they are regenerated from (width, height, seed, quality, sampling) by tools/jpeg_enc.c,
which is deterministic on one libm; every entry carries the sha256 of the JPEG it was computed for,
and a consumer whose regenerated input hashes differently must skip the comparison and say so.

  jpg_sha256   the input bytes
  rgb_sha256   top-down RGB payload reconstructed from the reference's BMP
  bmp_sha256   the reference's BMP file, byte for byte (rows of these sizes need no padding)
"""
import hashlib
import json
import os
import sys
import tempfile
from concurrent.futures import ProcessPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hjdtools as T  # noqa: E402

OUT = os.path.join(T.ROOT, "tests", "golden", "large_manifest.json")

# (width, height, seed, quality, hf, vf)
CASES = [(3840, 2160, 1000 + k, 85, 2, 2) for k in range(8)]      # bench.py, rank 0, --distinct 8
CASES += [(1920, 1080, 500 + k, 85, 2, 2) for k in range(6)]       # test_config3_batch_of_1080p
CASES += [(1920, 1080, 1234 + 1920, 85, 2, 2), (3840, 2160, 1234 + 3840, 85, 2, 2)]  # test_full_size_synthetic


def key(w, h, seed, q, hf, vf):
    return "syn_%dx%d_s%d_q%d_%dx%d" % (w, h, seed, q, hf, vf)


def one(case):
    w, h, seed, q, hf, vf = case
    j = T.synth_jpeg(w, h, seed, q, hf, vf)
    with tempfile.TemporaryDirectory() as td:
        jp, bp = os.path.join(td, "a.jpg"), os.path.join(td, "a.bmp")
        open(jp, "wb").write(j)
        rc = T.ref_convert(jp, bp, big=True, timeout=900)
        assert rc == 0, case
        bmp = open(bp, "rb").read()
    assert (3 * w) % 4 == 0
    body = np.frombuffer(bmp, np.uint8)[54:].reshape(h, 3 * w)
    rgb = np.ascontiguousarray(body[::-1].reshape(h, w, 3)[:, :, ::-1])
    # the restatement must agree here too (it is what the GPU tests fall back to for other seeds)
    orc, orgb, _, _, _ = T.oracle_decode(j)
    assert orc == 0 and np.array_equal(orgb, rgb), ("oracle differs from the compiled reference", case)
    return key(*case), {"width": w, "height": h, "seed": seed, "quality": q, "hf": hf, "vf": vf, "bytes": len(j),
                        "jpg_sha256": hashlib.sha256(j).hexdigest(), "rgb_sha256": hashlib.sha256(rgb.tobytes()).hexdigest(),
                        "bmp_sha256": hashlib.sha256(bmp).hexdigest(), "ref_build": "big"}


def main():
    assert T.ref_binary(big=True), "make -C oracle ref first"
    with ProcessPoolExecutor(max_workers=4) as ex:
        res = dict(ex.map(one, CASES))
    json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT, len(res), "entries")


if __name__ == "__main__":
    main()
