"""Test/bench tooling: synthetic image generator + baseline JPEG encoder (tools/jpeg_enc.c)
and a ctypes view of the CPU oracle (oracle/).  Not product code."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ENC = None
_ORACLE = None


def kernel_src_hash():
    """sha256 (16 hex digits) over the kernel sources: profiles/ summaries carry it, bench.py ignores a summary whose stamp
    differs from the sources it runs (the counters then describe another binary)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "hls-jpeg-decoder_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _build(target_dir, so, cmd):
    path = os.path.join(ROOT, target_dir, so)
    if not os.path.exists(path):
        subprocess.run(cmd, shell=True, check=True, cwd=os.path.join(ROOT, target_dir))
    return path


class EncParams(ctypes.Structure):
    """hje_params (tools/jpeg_enc.c)."""
    _fields_ = [("w", ctypes.c_uint), ("h", ctypes.c_uint), ("hf", ctypes.c_uint), ("vf", ctypes.c_uint),
                ("quality", ctypes.c_int), ("qtab", ctypes.c_void_p), ("three_dqt", ctypes.c_int),
                ("huff_mode", ctypes.c_int), ("restart_mcus", ctypes.c_uint), ("rgb", ctypes.c_void_p),
                ("coefs", ctypes.c_void_p)]


HUFF_STD, HUFF_SIX_PERMUTED, HUFF_SIX_OPTIMISED, HUFF_FOUR_OPTIMISED = 0, 1, 2, 3


def enc_lib():
    global _ENC
    if _ENC is None:
        p = _build("tools", "libhjd_enc.so", "gcc -O2 -fPIC -shared -o libhjd_enc.so jpeg_enc.c -lm")
        e = ctypes.CDLL(p)
        e.hje_encode.restype = ctypes.c_size_t
        e.hje_encode.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_int, ctypes.c_uint,
                                 ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p, ctypes.c_size_t]
        e.hje_encode_ex.restype = ctypes.c_size_t
        e.hje_encode_ex.argtypes = [ctypes.POINTER(EncParams), ctypes.c_void_p, ctypes.c_size_t]
        e.hje_synth_rgb.restype = None
        e.hje_synth_rgb.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint64, ctypes.c_double]
        _ENC = e
    return _ENC


def synth_rgb(w, h, seed, sigma=6.0):
    a = np.empty((h, w, 3), np.uint8)
    enc_lib().hje_synth_rgb(a.ctypes.data, w, h, seed, sigma)
    return a


def encode_jpeg(rgb, quality=85, hf=2, vf=2, restart_mcus=0):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    cap = w * h * 3 + 65536
    out = np.empty(cap, np.uint8)
    n = enc_lib().hje_encode(rgb.ctypes.data, w, h, quality, hf, vf, restart_mcus, out.ctypes.data, cap)
    assert 0 < n <= cap, "encoder failed"
    return out[:n].tobytes()


def encode_jpeg_ex(rgb=None, quality=85, hf=2, vf=2, restart_mcus=0, three_dqt=False, huff_mode=HUFF_STD, qtab=None,
                   coefs=None, size=None):
    """General form of the test encoder: three DQTs (Cb and Cr quantised differently), six Huffman
    tables (permuted Annex K or optimised per component), explicit quantiser tables (3 x 64, natural
    order) and explicit coefficient blocks (scan order, zig-zag, [0] = DC value; `size` = (w, h))."""
    P = EncParams()
    keep = []
    if coefs is not None:
        w, h = size
        coefs = np.ascontiguousarray(coefs, dtype=np.int16)
        nm = ((w + 8 * hf - 1) // (8 * hf)) * ((h + 8 * vf - 1) // (8 * vf))
        assert coefs.shape == (nm * (hf * vf + 2), 64), coefs.shape
        P.coefs = coefs.ctypes.data
        keep.append(coefs)
    else:
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        h, w, _ = rgb.shape
        P.rgb = rgb.ctypes.data
        keep.append(rgb)
    if qtab is not None:
        qtab = np.ascontiguousarray(qtab, dtype=np.uint8)
        assert qtab.shape == (3, 64)
        P.qtab = qtab.ctypes.data
        keep.append(qtab)
    P.w, P.h, P.hf, P.vf, P.quality = w, h, hf, vf, quality
    P.three_dqt, P.huff_mode, P.restart_mcus = int(three_dqt), int(huff_mode), restart_mcus
    cap = w * h * 3 + (coefs.size * 4 if coefs is not None else 0) + 65536
    out = np.empty(cap, np.uint8)
    n = enc_lib().hje_encode_ex(ctypes.byref(P), out.ctypes.data, cap)
    assert 0 < n <= cap, "encoder failed (a coefficient without a code, or bad arguments)"
    return out[:n].tobytes()


def synth_jpeg(w, h, seed, quality=85, hf=2, vf=2, restart_mcus=0, sigma=6.0):
    return encode_jpeg(synth_rgb(w, h, seed, sigma), quality, hf, vf, restart_mcus)


# ---- oracle ------------------------------------------------------------------

class OracleStats(ctypes.Structure):
    _fields_ = [("final_stream_index", ctypes.c_size_t), ("resyncs_block_start", ctypes.c_uint),
                ("resyncs_mid_block", ctypes.c_uint), ("resyncs_dirty", ctypes.c_uint), ("nblocks", ctypes.c_ulong),
                ("nonzero_coefs", ctypes.c_ulong)]


HJO_OK, HJO_ERR_NOT_JPEG, HJO_ERR_UNSUPPORTED, HJO_ERR_CORRUPT, HJO_ERR_REF_WOULD_HANG, HJO_ERR_REF_OVERRUN, \
    HJO_ERR_DC_CODE_MISSING = range(7)


def oracle_lib():
    global _ORACLE
    if _ORACLE is None:
        p = os.path.join(ROOT, "oracle", "libhjd_oracle.so")
        if not os.path.exists(p):
            subprocess.run("make -s libhjd_oracle.so", shell=True, check=True, cwd=os.path.join(ROOT, "oracle"))
        o = ctypes.CDLL(p)
        o.hjo_decode.restype = ctypes.c_int
        o.hjo_decode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_void_p, ctypes.POINTER(OracleStats)]
        o.hjo_block_idct.restype = None
        o.hjo_block_idct.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        o.hjo_ycc_to_rgb.restype = None
        o.hjo_bmp_size.restype = ctypes.c_size_t
        o.hjo_bmp_size.argtypes = [ctypes.c_uint, ctypes.c_uint]
        o.hjo_bmp_pack.restype = None
        o.hjo_bmp_pack.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p]
        o.hjo_convert_jpg_file.restype = ctypes.c_int
        o.hjo_convert_jpg_file.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        o.hjo_cos_table.restype = ctypes.POINTER(ctypes.c_float)
        _ORACLE = o
    return _ORACLE


def jpeg_dims(jpg):
    """(width, height, hf, vf) from the SOF0 segment."""
    b = jpg
    p = 2
    while p + 4 <= len(b):
        assert b[p] == 0xFF
        while b[p + 1] == 0xFF:
            p += 1
        m = b[p + 1]
        ln = (b[p + 2] << 8) | b[p + 3]
        if m == 0xC0:
            h = (b[p + 5] << 8) | b[p + 6]
            w = (b[p + 7] << 8) | b[p + 8]
            return w, h, b[p + 11] >> 4, b[p + 11] & 15
        if m == 0xDA:
            break
        p += 2 + (0 if m in (0xD8, 0xD9) else ln)
    raise ValueError("no SOF0")


def oracle_decode(jpg, want_coefs=False, want_planes=False):
    """Returns (status, rgb[h,w,3] or None, coefs or None, planes or None, stats)."""
    o = oracle_lib()
    try:
        w, h, hf, vf = jpeg_dims(jpg)
    except Exception:
        st = OracleStats()
        rc = o.hjo_decode(jpg, len(jpg), None, None, None, ctypes.byref(st))
        return rc, None, None, None, st
    nm = ((w + 8 * hf - 1) // (8 * hf)) * ((h + 8 * vf - 1) // (8 * vf)) if hf and vf else 0
    nb = nm * (hf * vf + 2)
    rgb = np.zeros((h, w, 3), np.uint8)
    coefs = np.zeros((nb, 64), np.int16) if want_coefs else None
    planes = np.zeros((nm, (hf * vf + 2) * 64), np.uint8) if want_planes else None
    st = OracleStats()
    rc = o.hjo_decode(jpg, len(jpg), rgb.ctypes.data, coefs.ctypes.data if want_coefs else None,
                      planes.ctypes.data if want_planes else None, ctypes.byref(st))
    return rc, rgb, coefs, planes, st


def oracle_bmp(rgb):
    o = oracle_lib()
    h, w, _ = rgb.shape
    out = np.empty(o.hjo_bmp_size(w, h), np.uint8)
    rgb = np.ascontiguousarray(rgb)
    o.hjo_bmp_pack(rgb.ctypes.data, w, h, out.ctypes.data)
    return out.tobytes()


def ref_binary(big=True):
    p = os.path.join(ROOT, "oracle", "_ref", "ref_big" if big else "ref_small")
    return p if os.path.exists(p) else None


def ref_convert(jpg_path, bmp_path, big=True, timeout=600):
    """Run the compiled reference (oracle/_ref) — always under a timeout."""
    exe = ref_binary(big)
    assert exe, "oracle/_ref not built (make -C oracle ref; needs /root/reference)"
    return subprocess.run("ulimit -s unlimited; exec %s %s %s > /dev/null" % (exe, jpg_path, bmp_path),
                          shell=True, timeout=timeout).returncode


# ---- reference-produced hashes for the full-size inputs (tools/make_large_golden.py) ----------------

_LARGE = None


def large_golden(w, h, seed, quality, hf, vf, jpg, rst=0):
    """Entry of tests/golden/large_manifest.json for this synthetic input, or None when there is none
    or when the regenerated input is not byte-identical to the one the reference was run on (another
    libm): the caller then falls back to the oracle and says so."""
    import hashlib
    import json
    global _LARGE
    if _LARGE is None:
        p = os.path.join(ROOT, "tests", "golden", "large_manifest.json")
        _LARGE = json.load(open(p)) if os.path.exists(p) else {}
    ent = _LARGE.get("syn_%dx%d_s%d_q%d_%dx%d" % (w, h, seed, quality, hf, vf) + ("_rst%d" % rst if rst else ""))
    if ent is None:
        return None
    if hashlib.sha256(jpg).hexdigest() != ent["jpg_sha256"]:
        print("large_golden: regenerated %dx%d seed %d differs from the input the reference saw; using the oracle"
              % (w, h, seed))
        return None
    return ent
