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

# (width, height, seed, quality, hf, vf[, restart interval in MCUs, noise sigma])
CASES = [(3840, 2160, 1000 + k, 85, 2, 2) for k in range(8)]      # bench.py, rank 0, --distinct 8
CASES += [(1920, 1080, 500 + k, 85, 2, 2) for k in range(6)]       # test_config3_batch_of_1080p
CASES += [(1920, 1080, 1234 + 1920, 85, 2, 2), (3840, 2160, 1234 + 3840, 85, 2, 2)]  # test_full_size_synthetic
# round 5: restart streams at full size (the reference's unchanged host + libhjd.so, tests/test_gpu_parity.py:
# test_reference_host_binary_runs_on_the_gpu; hjd_jpeg_decode_hw without Ri) — one MCU row per interval, 4:4:4 with 16-MCU intervals
CASES += [(1920, 1080, 700, 85, 2, 2, 120, 6.0), (1920, 1080, 701, 85, 1, 1, 16, 6.0)]


def key(w, h, seed, q, hf, vf, rst=0, sigma=6.0):
    return "syn_%dx%d_s%d_q%d_%dx%d" % (w, h, seed, q, hf, vf) + ("_rst%d" % rst if rst else "")


def one(case):
    w, h, seed, q, hf, vf = case[:6]
    rst, sigma = (case[6], case[7]) if len(case) > 6 else (0, 6.0)
    j = T.synth_jpeg(w, h, seed, q, hf, vf, rst, sigma)
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
    return key(*case), {"width": w, "height": h, "seed": seed, "quality": q, "hf": hf, "vf": vf, "bytes": len(j), "restart_mcus": rst,
                        "jpg_sha256": hashlib.sha256(j).hexdigest(), "rgb_sha256": hashlib.sha256(rgb.tobytes()).hexdigest(),
                        "bmp_sha256": hashlib.sha256(bmp).hexdigest(), "ref_build": "big"}


def main():
    assert T.ref_binary(big=True), "make -C oracle ref first"
    # entries already in the manifest are kept (the reference does not change); --all recomputes every one
    res = {} if "--all" in sys.argv or not os.path.exists(OUT) else json.load(open(OUT))
    todo = [c for c in CASES if key(*c) not in res]
    with ProcessPoolExecutor(max_workers=4) as ex:
        res.update(dict(ex.map(one, todo)))
    json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT, len(res), "entries")


if __name__ == "__main__":
    main()
