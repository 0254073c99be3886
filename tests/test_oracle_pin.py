"""Pins the CPU oracle (oracle/hjd_oracle.c) against what the compiled reference emitted
for the committed fixtures (tests/golden/manifest.json, made by tools/make_golden.py) and
against the known-answer anchors of SURVEY App. C.  CPU only."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import hjdtools as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _masked_bmp(rgb):
    h, w, _ = rgb.shape
    return T.oracle_bmp(rgb)  # oracle writes zero pad bytes == masked form


def test_manifest_hashes(golden):
    d, man = golden
    n = 0
    for name, ent in sorted(man.items()):
        data = open(os.path.join(d, name), "rb").read()
        assert hashlib.sha256(data).hexdigest() == ent["sha256"], name
        rc, rgb, _, _, st = T.oracle_decode(data)
        if ent["status"] == "unsupported":
            assert rc == T.HJO_ERR_UNSUPPORTED, name
            continue
        assert ent["status"] == "ok", name
        assert rc == T.HJO_OK, name
        assert hashlib.sha256(rgb.tobytes()).hexdigest() == ent["rgb_sha256"], name
        bmp = _masked_bmp(rgb)
        assert len(bmp) == ent["bmp_bytes"], name
        assert hashlib.sha256(bmp).hexdigest() == ent["bmp_sha256_masked"], name
        pad = (4 - 3 * ent["width"] % 4) % 4
        if pad <= 1:  # no undefined pad bytes: the raw reference hash must match too
            assert hashlib.sha256(bmp).hexdigest() == ent["bmp_sha256"], name
        assert st.final_stream_index == ent["final_stream_index"], name
        assert (st.resyncs_dirty == 0) == (ent["restart"] != "dirty"), name
        n += 1
    assert n >= 20


def test_lenna_anchors(golden):
    """Lenna hash, first block, first tile, corner pixels, final stream index."""
    d, man = golden
    data = open(os.path.join(d, "Lenna.jpg"), "rb").read()
    assert hashlib.sha256(data).hexdigest() == "8640b22549789d5aeabebb487b0b6cda05119c3bff8f67699850a0f8177532d7"
    rc, rgb, coefs, planes, st = T.oracle_decode(data, want_coefs=True, want_planes=True)
    assert rc == 0
    assert hashlib.sha256(T.oracle_bmp(rgb)).hexdigest() == \
        "af6996f7f0cb092f8282bd661f95869d6c2c983b364d8722f6ff581eaf5d12e0"
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == \
        "c29f58d3186999276880480f4d5df2212d834ec2ac8b4158488ead1be4f17940"
    first = [130, 5, 8, -5, -1, 3, 0, 1, 0, 2, -1, 1, -1, -2, 0, 0, 1, 1, 1, -1, 1, 0, 0, 0, 0, 0, 1, -1, 1, -1,
             0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1]
    assert coefs[0, :len(first)].tolist() == first and not coefs[0, len(first):].any()
    tile = planes[0, :256].reshape(16, 16)[:8, :8]
    assert tile[0].tolist() == [163, 161, 161, 161, 162, 155, 165, 160]
    assert tile[7].tolist() == [157, 159, 157, 155, 158, 160, 155, 156]
    assert rgb[0, :4].tolist() == [[223, 138, 129], [221, 136, 127], [222, 135, 130], [222, 135, 130]]
    assert rgb[511, :2].tolist() == [[83, 20, 53], [86, 23, 56]]
    assert st.final_stream_index == 104033
    assert st.nblocks == 6144


def test_idct_zero_skip_equals_full_terms():
    """skipping zero coefficients leaves every output byte unchanged, including
    for coefficient magnitudes that overflow int16 after dequantisation."""
    o = T.oracle_lib()
    rng = np.random.default_rng(5)
    a = np.empty((8, 8), np.uint8)
    b = np.empty((8, 8), np.uint8)
    for it in range(3000):
        coef = np.zeros(64, np.int16)
        k = rng.integers(0, 64)
        idx = rng.choice(64, k, replace=False)
        scale = [8, 200, 2047, 32767][it % 4]
        coef[idx] = rng.integers(-scale, scale + 1, k)
        q = rng.integers(1, 256, 64).astype(np.uint8)
        o.hjo_block_idct(coef.ctypes.data, q.ctypes.data, a.ctypes.data, 8, 0)
        o.hjo_block_idct(coef.ctypes.data, q.ctypes.data, b.ctypes.data, 8, 1)
        assert (a == b).all()


APP_B = """
0x1p+0  0x1.f62c08p-1  0x1.d910b8p-1  0x1.a9cc2p-1   0x1.6a2ecap-1  0x1.1ca9eep-1  0x1.886ecap-2  0x1.90f1bep-3
0x1p+0  0x1.a9cc2p-1   0x1.886ecap-2 -0x1.8dbedp-3  -0x1.699b2p-1  -0x1.f6032cp-1 -0x1.d9606ap-1 -0x1.1d5756p-1
0x1p+0  0x1.1ca9eep-1 -0x1.86ed1p-2  -0x1.f6032cp-1 -0x1.6ac236p-1  0x1.8a8b92p-3  0x1.d8706ep-1  0x1.aab36ap-1
0x1p+0  0x1.90f1bep-3 -0x1.d8c0bap-1 -0x1.1d5756p-1  0x1.690742p-1  0x1.aab36ap-1 -0x1.83e8d8p-2 -0x1.f6a4aep-1
0x1p+0 -0x1.8dbedp-3  -0x1.d9606ap-1  0x1.1b4e94p-1  0x1.6b556cp-1 -0x1.a86f2p-1  -0x1.8cf256p-2  0x1.f5869ep-1
0x1p+0 -0x1.1bfc56p-1 -0x1.89f044p-2  0x1.f67ccap-1 -0x1.68731cp-1 -0x1.9a88c4p-3  0x1.da4da2p-1 -0x1.a7851ep-1
0x1p+0 -0x1.a9581p-1   0x1.856b06p-2  0x1.9756cp-3  -0x1.6be866p-1  0x1.f6cc3cp-1 -0x1.d77dbp-1   0x1.1895aep-1
0x1p+0 -0x1.f6032cp-1  0x1.d8706ep-1 -0x1.a86f2p-1   0x1.67dec4p-1 -0x1.19442cp-1  0x1.7f609p-2  -0x1.7a87f6p-3
"""  # what the reference's x86-64 / glibc 2.35 build evaluates, row = x (or y), column = u (or v)


def test_cos_table_is_appendix_b_everywhere():
    """All 64 entries, in every place that holds them: the oracle's table and the table the HIP kernel is compiled with
    (csrc/hjd_pixels.hip: c_cos) equal SURVEY App. B bit for bit."""
    import re
    want = np.array([float.fromhex(x) for x in APP_B.split()], np.float32)
    assert want.size == 64 and want[6 * 8 + 7] == np.float32(float.fromhex("0x1.1895aep-1"))   # the libm-specific entry, not 0x1.1895b0p-1
    t = np.ctypeslib.as_array(T.oracle_lib().hjo_cos_table(), shape=(64,)).copy()
    assert t.dtype == np.float32 and t.tobytes() == want.tobytes()
    src = open(os.path.join(ROOT, "hls-jpeg-decoder_amd", "csrc", "hjd_pixels.hip")).read()
    body = src[src.index("__constant__ float c_cos[64] = {") + len("__constant__ float c_cos[64] = {"):]
    body = body[:body.index("};")]
    dev = np.array([float.fromhex(x.rstrip("f")) for x in re.findall(r"-?0x[0-9a-f.]+p[+-]\d+f", body)], np.float32)
    assert dev.size == 64 and dev.tobytes() == want.tobytes()


def test_cos_table_is_what_the_reference_expression_gives_on_this_libm():
    """The reference computes cosf(((2*x+1) * u * PI) / 16) with PI = 3.14f at run time (loadjpg.cpp:108, :120).  Where the
    local libm is the one the reference hashes were produced on (glibc 2.35: the build container and the GPU box image) all 64
    values must come out as the table; on another libm the comparison is skipped with a message — the table, not the libm, is
    the contract.  tools/gen_costable.c is the same check as a C program."""
    import ctypes
    import ctypes.util
    libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    libm.cosf.restype = ctypes.c_float
    libm.cosf.argtypes = [ctypes.c_float]
    want = np.array([float.fromhex(x) for x in APP_B.split()], np.float32).reshape(8, 8)
    f = np.float32
    got = np.empty((8, 8), np.float32)
    for x in range(8):
        for u in range(8):
            arg = f(f(f((2 * x + 1) * u) * f(3.14)) / f(16))   # int * int, then * float, then / 16: one rounding per operation
            got[x, u] = libm.cosf(float(arg))
    diff = [(x, u) for x in range(8) for u in range(8) if got[x, u].tobytes() != want[x, u].tobytes()]
    if diff:
        pytest.skip("this libm's cosf differs from glibc 2.35's at %s: the hard-coded table stays the contract" % diff)


def test_bmp_header_bytes(golden):
    d, _ = golden
    rc, rgb, _, _, _ = T.oracle_decode(open(os.path.join(d, "Lenna.jpg"), "rb").read())
    bmp = T.oracle_bmp(rgb)
    assert bmp[:54].hex() == "424d36000c000000000036000000280000000002000000020000010018" + "00" * 25
    assert len(bmp) == 786486


@pytest.mark.skipif(T.ref_binary() is None, reason="compiled reference (oracle/_ref) not present")
def test_live_reference_agrees_on_fresh_inputs(tmp_path):
    """Where oracle/_ref exists: run the compiled reference itself on inputs that are NOT in the
    manifest and compare byte for byte (pad bytes >= 2 masked)."""
    cases = [(320, 200, 85, 2, 2, 0, 101), (123, 77, 70, 2, 1, 0, 102), (200, 120, 92, 1, 1, 0, 103),
             (160, 160, 85, 2, 2, 5, 104), (512, 256, 85, 2, 2, 0, 105)]
    for (w, h, q, hf, vf, rst, seed) in cases:
        j = T.synth_jpeg(w, h, seed, q, hf, vf, rst, 7.0)
        jp, bp = str(tmp_path / "a.jpg"), str(tmp_path / "a.bmp")
        open(jp, "wb").write(j)
        try:
            rc = T.ref_convert(jp, bp, big=True, timeout=60)
        except subprocess.TimeoutExpired:
            pytest.fail("reference hung")
        assert rc == 0
        ref = np.frombuffer(open(bp, "rb").read(), np.uint8)
        orc, rgb, _, _, _ = T.oracle_decode(j)
        assert orc == 0
        mine = np.frombuffer(T.oracle_bmp(rgb), np.uint8)
        pad = (4 - 3 * w % 4) % 4
        stride = 3 * w + pad
        diff = np.nonzero(ref != mine)[0]
        diff = [x for x in diff if x < 54 or (x - 54) % stride < 3 * w + 1]
        assert not diff, (w, h, hf, vf, rst)
