"""GPU parity: the HIP path (through the C ABI, libhjd.so) against the CPU oracle and the golden
hashes the compiled reference produced.  Bit-exact or fail."""
import ctypes
import hashlib
import importlib
import os

import numpy as np
import pytest

import hjdtools as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hjd():
    import torch  # noqa: F401  (loads the HIP runtime the library binds to)
    return importlib.import_module("hls-jpeg-decoder_amd")


@pytest.fixture(scope="module")
def dec(hjd):
    d = hjd.Decoder(0)
    yield d
    d.close()


def _dbg(hjd, env):
    """Test switches by their old names -> (environment proper, hjd_options.debug_flags).  The switches that select among paths
    that must give the same bytes are options of the decoder (include/hjd.h: HJD_DBG_*), not variables of the process."""
    env = dict(env)
    f = 0
    if env.pop("HJD_NO_SMALL_SYNC", None):
        f |= hjd.DBG_NO_SMALL_SYNC
    if env.pop("HJD_NO_SEQ_SYNC", None):
        f |= hjd.DBG_NO_SEQ_SYNC
    if env.pop("HJD_NO_SEG_EMULATION", None):
        f |= hjd.DBG_NO_SEG_EMULATION
    if "HJD_SIBLINGS" in env:
        f |= hjd.dbg_siblings(int(env.pop("HJD_SIBLINGS")))
    if "HJD_SEG_CUT" in env:
        f |= hjd.dbg_seg_cut(int(env.pop("HJD_SEG_CUT")))
    return env, f


def _ok_fixtures(golden, restart=("none",)):
    d, man = golden
    return [(n, e, open(os.path.join(d, n), "rb").read()) for n, e in sorted(man.items())
            if e["status"] == "ok" and e.get("restart") in restart]


def test_golden_rgb_and_bmp(hjd, dec, golden):
    """Every committed fixture without restart markers: RGB payload and BMP file image equal the
    compiled reference's (hashes) and the oracle's (bytes)."""
    fx = _ok_fixtures(golden)
    assert len(fx) >= 17
    jpgs = [f[2] for f in fx]
    st, rgbs = dec.decode(jpgs, hjd.OUT_RGB)
    st2, bmps = dec.decode(jpgs, hjd.OUT_BMP)
    for (name, ent, data), s, rgb, s2, bmp in zip(fx, st, rgbs, st2, bmps):
        assert s == 0 and s2 == 0, (name, s, s2)
        orc, orgb, _, _, _ = T.oracle_decode(data)
        assert orc == 0
        assert rgb.shape == orgb.shape, name
        assert np.array_equal(rgb, orgb), (name, int((rgb != orgb).sum()))
        assert hashlib.sha256(rgb.tobytes()).hexdigest() == ent["rgb_sha256"], name
        assert bmp.tobytes() == T.oracle_bmp(orgb), name
        assert hashlib.sha256(bmp.tobytes()).hexdigest() == ent["bmp_sha256_masked"], name


def test_lenna_is_the_reference_bmp(hjd, dec, golden, tmp_path):
    """Config 1: data/Lenna.jpg -> BMP through the ConvertJpgFile-shaped entry point."""
    d, man = golden
    out = tmp_path / "out.bmp"
    assert hjd.convert_jpg_file(os.path.join(d, "Lenna.jpg"), str(out)) == 1
    assert hashlib.sha256(out.read_bytes()).hexdigest() == \
        "af6996f7f0cb092f8282bd661f95869d6c2c983b364d8722f6ff581eaf5d12e0"


ROUND2 = ("syn_250x187_420_q85_3dqt.jpg", "syn_250x187_422_q85_3dqt_6huff.jpg", "syn_96x64_444_q60_3dqt_6perm.jpg",
          "ext_64x64_420_int16wrap.jpg", "ext_64x48_444_dense_ones.jpg")  # what they exercise: tests/test_cabi_cpu.py


def test_coefficients_match_oracle(hjd, dec, golden):
    """Entropy-decode stage alone: zig-zag coefficients with DC prediction applied."""
    fx = _ok_fixtures(golden)
    for name, ent, data in fx[:8] + [f for f in fx if f[0] in ROUND2]:
        b = dec.prepare([data])
        b.run()
        assert b.finish() == [0]
        orc, _, ocoef, _, _ = T.oracle_decode(data, want_coefs=True)
        coefs = b.download_coefs(0, ocoef.shape[0])
        b.close()
        assert np.array_equal(coefs, ocoef), name


@pytest.mark.parametrize("w,h,hf,vf,q", [(1920, 1080, 2, 2, 85), (3840, 2160, 2, 2, 85), (1000, 700, 2, 1, 60),
                                          (777, 333, 1, 1, 95), (640, 480, 1, 2, 75), (512, 512, 4, 1, 85)])
def test_full_size_synthetic(hjd, dec, w, h, hf, vf, q):
    data = T.synth_jpeg(w, h, 1234 + w, q, hf, vf)
    st, outs = dec.decode([data])
    assert st == [0]
    orc, orgb, _, _, _ = T.oracle_decode(data)
    assert orc == 0
    assert np.array_equal(outs[0], orgb), int((outs[0] != orgb).sum())
    ent = T.large_golden(w, h, 1234 + w, q, hf, vf, data)  # 1080p and 4K: what the compiled reference emitted
    if ent is not None:
        assert hashlib.sha256(outs[0].tobytes()).hexdigest() == ent["rgb_sha256"]
        st, bmps = dec.decode([data], hjd.OUT_BMP)
        assert st == [0] and hashlib.sha256(bmps[0].tobytes()).hexdigest() == ent["bmp_sha256"]
    else:
        assert (w, h) not in ((1920, 1080), (3840, 2160)), "reference hash for the headline sizes went missing"


def test_bench_inputs_match_reference_hashes(hjd, dec):
    """The exact inputs bench.py decodes on rank 0 (3840x2160 4:2:0 q85, seeds 1000..1007): RGB bytes hash to what
    the compiled reference (oracle/_ref/ref_big, tools/make_large_golden.py) wrote for the same inputs."""
    jpgs = [T.synth_jpeg(3840, 2160, 1000 + k, 85, 2, 2) for k in range(8)]
    ents = [T.large_golden(3840, 2160, 1000 + k, 85, 2, 2, j) for k, j in enumerate(jpgs)]
    if any(e is None for e in ents):
        pytest.skip("regenerated bench inputs differ from the ones the reference was run on (other libm)")
    st, outs = dec.decode(jpgs)
    assert st == [0] * 8
    for o, e in zip(outs, ents):
        assert hashlib.sha256(o.tobytes()).hexdigest() == e["rgb_sha256"]


def test_forced_host_resynchronisation(hjd, golden):
    """hjd_batch_finish's fallback: with no warm-up subsequences and no in-stream repair launches the first pass
    leaves stale start states at every wave border; k_write's induction check must catch them
    (HJD_FLAG_NOT_CONVERGED), the host must run repair rounds until none is left, and the bytes must be the
    same as from a normal run.  Also: a DIRTY verdict from the unstuff stage survives those rounds."""
    d, man = golden
    names = ["Lenna.jpg", "syn_250x187_444_q85.jpg", "pil_250x187_444_q95_opt.jpg", "syn_256x192_420_q85_rst16.jpg",
             "syn_250x187_422_q85_3dqt_6huff.jpg"]
    jpgs = [open(os.path.join(d, n), "rb").read() for n in names]
    # a restart stream with a non-RSTn marker between intervals: flagged DIRTY by k_unstuff_write only
    rst = bytearray(T.synth_jpeg(320, 240, 91, 85, 2, 2, 10, 7.0))
    sos = rst.index(b"\xff\xda")
    k = rst.index(b"\xff\xd0", sos)
    rst[k + 1] = 0xC8
    jpgs.append(bytes(rst))
    for bits in (64, 256):
        dd = hjd.Decoder(0, subseq_bits=bits, sync_launches=1, sync_overlap=-1)
        b = dd.prepare(jpgs)
        b.run()
        st = b.finish()
        assert b.resync_count() > 0, "the fallback was not exercised"
        for n, s, i in zip(names, st, range(len(names))):
            assert s == 0, (n, s)
            assert hashlib.sha256(b.download(i).tobytes()).hexdigest() == man[n]["rgb_sha256"], (n, bits)
        # the marker verdict of the unstuff stage survived the repair rounds: the image went to the sequential emulation
        assert b.sequential_count() == 1
        orc, orgb, _, _, _ = T.oracle_decode(jpgs[-1])
        assert (st[-1] == 0 and np.array_equal(b.download(len(names)), orgb)) if orc == 0 else st[-1] == hjd.HJD_ERR_CORRUPT
        b.close()
        dd.close()


def test_two_bit_coefficients_and_optimised_tables(hjd, dec):
    """Record capacity: a scan whose AC coefficients cost two bits each (1-bit code for run 0 / size 1 plus the
    sign bit) holds 63 records per block in 126 bits.  And an optimised-table noisy image at full density."""
    rng = np.random.default_rng(3)
    w, h = 512, 256
    nb = (w // 16) * (h // 16) * 6
    co = np.zeros((nb, 64), np.int16)
    co[:, 1:] = rng.choice([-1, 1], (nb, 63))
    co[:, 0] = rng.integers(-300, 300, nb)
    qt = np.full((3, 64), 2, np.uint8)
    a = T.encode_jpeg_ex(None, 85, 2, 2, 0, True, T.HUFF_SIX_OPTIMISED, qtab=qt, coefs=co, size=(w, h))
    st0, jd = hjd.parse_header(a)
    assert st0 == 0 and jd.htac[0].bits[1] == 1  # the 1-bit code exists
    noisy = T.encode_jpeg_ex(T.synth_rgb(640, 480, 5, 40.0), 98, 1, 1, 0, True, T.HUFF_SIX_OPTIMISED)
    st, outs = dec.decode([a, noisy])
    assert st == [0, 0], st
    for j, o in zip((a, noisy), outs):
        orc, orgb, _, _, _ = T.oracle_decode(j)
        assert orc == 0 and np.array_equal(o, orgb)


def test_mixed_batch(hjd, dec, golden):
    """Independent images of different sizes / subsamplings / tables in one batch."""
    fx = _ok_fixtures(golden)
    jpgs = [f[2] for f in fx] + [T.synth_jpeg(800, 600, 9, 85, 2, 2), T.synth_jpeg(333, 222, 10, 50, 1, 1)]
    jpgs = jpgs * 2
    st, outs = dec.decode(jpgs)
    assert all(s == 0 for s in st)
    for j, o in zip(jpgs, outs):
        _, orgb, _, _, _ = T.oracle_decode(j)
        assert np.array_equal(o, orgb)


def test_thumbnail_batches_take_the_small_kernels(hjd):
    """Images of a few kilobytes: k_sync_seq (one lane decodes a whole scan), k_sync_small / k_write_small (one-wave workgroups) on a
    side stream beside the large images' kernels.  One batch of 700 thumbnails — five samplings, standard and per-image optimised
    tables (many table sets: many groups), some with restart markers (which keep the workgroup kinds), sizes on both sides of
    every threshold — between three large images; every distinct image equals the oracle, with the small kernels and without."""
    rng = np.random.default_rng(11)
    samp = [(2, 2), (1, 1), (2, 1), (1, 2), (4, 1)]
    distinct = []
    for k in range(60):
        w, h = int(rng.integers(1, 400)), int(rng.integers(1, 300))
        hf, vf = samp[k % 5]
        rgb = T.synth_rgb(w, h, 900 + k, float(rng.choice([2.0, 8.0, 30.0])))
        mode = T.HUFF_STD if k % 3 else T.HUFF_FOUR_OPTIMISED
        distinct.append(T.encode_jpeg_ex(rgb, int(rng.choice([35, 85, 97])), hf, vf, 3 if k % 7 == 0 else 0, True, mode))
    big = [T.synth_jpeg(1920, 1080, 31, 85, 2, 2), T.synth_jpeg(1280, 720, 32, 90, 1, 1), T.synth_jpeg(800, 600, 33, 85, 2, 1, 20, 7.0)]
    jpgs = [big[0]] + [distinct[k % 60] for k in range(350)] + [big[1]] + [distinct[(k * 7) % 60] for k in range(350)] + [big[2]]
    want = {}
    for j in set(jpgs):
        rc, orgb, _, _, _ = T.oracle_decode(j)
        want[j] = (rc, orgb)
    for env in ({}, {"HJD_NO_SEQ_SYNC": "1"}, {"HJD_NO_SEQ_SYNC": "1", "HJD_NO_SMALL_SYNC": "1"}):
        _, flags = _dbg(hjd, env)
        for streams in (0, 3):
            d = hjd.Decoder(0, streams=streams, debug_flags=flags)
            b = d.prepare(jpgs)
            pc = b.path_counts()   # the paths under test were really laid out (and really switched off)
            assert (pc["sync_seq"] > 300) == (not env) and (pc["sync_small"] > (0 if not env else 300)) == (len(env) < 2), (env, pc)
            assert (pc["write_small"] > 300) == (len(env) < 2) and pc["sync_large"] >= 3, (env, pc)
            b.run()
            st = b.finish()
            for k_, (j, s_) in enumerate(zip(jpgs, st)):
                rc, orgb = want[j]
                if rc == 0:
                    assert s_ == 0 and np.array_equal(b.download(k_), orgb), (env, streams, len(j))
                else:
                    assert s_ != 0
            b.close()
            d.close()


def test_sibling_trajectories_change_nothing_but_the_time(hjd):
    """Small batches: up to eleven further decodes of every subsequence — the other block-in-MCU labels, at a block's start and one symbol into it (k_sync_sib).  Their checkpoints are
    hints for the refinement walks; the bytes, the statuses and the absence of host re-synchronisations must not depend on them."""
    jpgs = [T.synth_jpeg(3840, 2160, 1000, 85, 2, 2), T.synth_jpeg(1920, 1080, 501, 85, 1, 1), T.synth_jpeg(1000, 700, 4243, 90, 2, 1, 25, 9.0),
            T.encode_jpeg_ex(T.synth_rgb(1600, 1200, 5, 12.0), 80, 2, 2, 0, True, T.HUFF_FOUR_OPTIMISED)]
    want = [T.oracle_decode(j)[1] for j in jpgs]
    for sib in (0, 1, 3, 5, 8, 11):
        for bits in (0, 1024, 4096):
            d = hjd.Decoder(0, subseq_bits=bits, debug_flags=hjd.dbg_siblings(sib))
            b = d.prepare(jpgs)
            for _ in range(2):  # the second run re-uses the flags of the first (epochs)
                b.run()
                st = b.finish()
            assert st == [0] * len(jpgs) and b.resync_count() == 0, (sib, bits, st)
            for k, w in enumerate(want):
                assert np.array_equal(b.download(k), w), (sib, bits, k)
            b.close()
            d.close()


@pytest.mark.parametrize("bits", [64, 128, 256, 512, 1024, 4096, 16384])
def test_subsequence_sizes(hjd, bits, golden):
    """The subsequence length is a tuning knob only: output must not depend on it."""
    d = hjd.Decoder(0, subseq_bits=bits)
    fx = _ok_fixtures(golden)[:6]
    st, outs = d.decode([f[2] for f in fx])
    for (name, ent, data), s, o in zip(fx, st, outs):
        assert s == 0
        assert hashlib.sha256(o.tobytes()).hexdigest() == ent["rgb_sha256"], (name, bits)
    d.close()


@pytest.mark.parametrize("bits", [64, 128, 256])
def test_dense_records_at_small_subsequences(hjd, bits):
    """Uniform noise at q100, 4:2:0: nearly every coefficient of every block is non-zero, and with 64-bit subsequences
    every run of a few records is padded to a whole burst, so that a 96-block IDCT group spans close to 8000 record
    entries (round 2 capped the walk at 7168 and silently dropped the rest)."""
    rng = np.random.default_rng(77)
    d = hjd.Decoder(0, subseq_bits=bits)
    jpgs = []
    for w, h in ((1024, 16), (272, 48)):
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        jpgs.append(T.encode_jpeg(rgb, 100, 2, 2))
    st, outs = d.decode(jpgs)
    assert st == [0, 0]
    for j, o in zip(jpgs, outs):
        rc, orgb, _, _, _ = T.oracle_decode(j)
        assert rc == 0 and np.array_equal(o, orgb), bits
    d.close()


@pytest.mark.parametrize("env", [{"HJD_PHASE_WORDS": "2"}, {"HJD_PHASE_WORDS": "4", "HJD_WRITE_WINDOWS": "2"}, {"HJD_PHASE_WORDS": "8", "HJD_WRITE_WINDOWS": "8"},
                                 {"HJD_PHASE_WORDS": "32"}, {"HJD_SIBLINGS": "3"}, {"HJD_SYNC_CONC": "0", "HJD_STAGGER": "1"},
                                 {"HJD_SYNC_OVERLAP": "0"}, {"HJD_SYNC_OVERLAP": "1"}, {"HJD_NO_SEG_EMULATION": "1"}, {"HJD_SEG_CUT": "1"}])
def test_tuning_knobs_never_change_a_byte(hjd, golden, env):
    """Window size (= checkpoint spacing = what one k_write lane decodes), windows per k_write lane, sibling trajectories forced on,
    the front-end concurrency limit, staggered ranges, in-stream repair launches, none or a single warm-up
    subsequence per synchronisation workgroup, the emulation of dirty restart streams by one lane or with its segments cut short: tuning only.
    Environment variables are read when a decoder is created; the test switches among them are hjd_options.debug_flags (_dbg)."""
    fx = _ok_fixtures(golden, restart=("none", "clean", "dirty"))
    big = [T.synth_jpeg(1280, 720, 4242, 85, 2, 2), T.synth_jpeg(1000, 700, 4243, 90, 1, 1, 25, 9.0)]
    want = [T.oracle_decode(j)[1] for j in big]
    env, flags = _dbg(hjd, env)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        for bits, streams, launches in ((0, 0, 0), (512, 3, 3), (4096, 2, 2)):
            d = hjd.Decoder(0, subseq_bits=bits, streams=streams, sync_launches=launches, debug_flags=flags)
            st, outs = d.decode([f[2] for f in fx] + big)
            for (name, ent, data), s, o in zip(fx, st, outs):
                assert s == 0 and hashlib.sha256(o.tobytes()).hexdigest() == ent["rgb_sha256"], (name, env, bits)
            for o, w in zip(outs[len(fx):], want):
                assert np.array_equal(o, w), (env, bits)
            d.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_restart_fixtures(hjd, dec, golden):
    """Committed restart-marker fixtures: the output equals the compiled reference's — inside the clean-resync
    domain through the parallel path, and for the dirty one (the reference fires its resync early on tiny blocks,
    SURVEY App. D.3) through the sequential emulation kernel.  With dirty_mode = 1 the dirty one is refused."""
    d, man = golden
    fx = _ok_fixtures(golden, restart=("clean", "dirty"))
    assert len(fx) >= 5 and any(f[1]["restart"] == "dirty" for f in fx)
    jpgs = [f[2] for f in fx]
    b = dec.prepare(jpgs)
    b.run()
    st = b.finish()
    for k, ((name, ent, data), s) in enumerate(zip(fx, st)):
        assert s == 0, (name, s)
        assert hashlib.sha256(b.download(k).tobytes()).hexdigest() == ent["rgb_sha256"], name
    assert b.sequential_count() == sum(1 for f in fx if f[1]["restart"] == "dirty")
    b.close()
    strict = hjd.Decoder(0, dirty_mode=1)
    st, outs = strict.decode(jpgs)
    for (name, ent, data), s, o in zip(fx, st, outs):
        if ent["restart"] == "dirty":
            assert s == hjd.HJD_ERR_DIRTY_RESTART, (name, s)
        else:
            assert s == 0 and hashlib.sha256(o.tobytes()).hexdigest() == ent["rgb_sha256"], name
    strict.close()


@pytest.mark.parametrize("w,h,hf,vf,rst,sigma", [
    (640, 480, 2, 2, 40, 6.0),    # one MCU row per interval
    (640, 480, 2, 2, 1, 6.0),     # every MCU its own interval
    (333, 222, 1, 1, 7, 8.0),     # 4:4:4, ragged interval length
    (500, 300, 2, 1, 3, 6.0),     # 4:2:2
    (1920, 1080, 2, 2, 120, 6.0),
    (256, 256, 2, 2, 5, 0.5),     # smooth content: tiny trailing blocks are likely
    (256, 256, 1, 1, 1, 0.0),     # flat-ish 4:4:4, 1-MCU intervals: dirty in practice
    (128, 128, 2, 2, 0xFFFF, 6.0),  # DRI present, interval longer than the image: no RSTn at all
    (320, 240, 2, 2, -1, 6.0),    # DRI segment with Ri = 0: the reference keys its resync checks on the segment LENGTH
    (200, 120, 1, 1, -1, 0.5),    # (openjpg.cpp:441-446), so they are live although the scan holds no RSTn; smooth content:
    (64, 64, 2, 1, -1, 0.0),      # the last blocks are short and the reference's cursor reaches EOI early
])
def test_restart_synthetic_matches_oracle_or_is_refused(hjd, dec, w, h, hf, vf, rst, sigma):
    """Mixed subsampling + restart intervals (BASELINE config 5).  The oracle emulates the reference's
    resync rules exactly; whenever the reference decodes the stream — inside the clean-resync domain or not — the
    HIP path must be bit-exact, and where the reference cannot (it would hang / overrun) it must report an error."""
    if rst < 0:
        data = T.synth_jpeg(w, h, 77, 85, hf, vf, 0, sigma)
        sos = data.index(b"\xff\xda")
        data = data[:sos] + b"\xff\xdd\x00\x04\x00\x00" + data[sos:]
        rc, jd = hjd.parse_header(data)
        assert rc == 0 and jd.restart_interval > 0 and jd.restart_mcus == 0
    else:
        data = T.synth_jpeg(w, h, 77 + rst % 50, 85, hf, vf, rst, sigma)
    orc, orgb, _, _, ost = T.oracle_decode(data)
    st, outs = dec.decode([data])
    if orc == 0:
        assert st == [0], (st, ost.resyncs_dirty)
        assert np.array_equal(outs[0], orgb)
    else:
        assert st[0] == hjd.HJD_ERR_CORRUPT, (st, orc)


def test_segment_parallel_emulation_equals_the_one_lane_kernel(hjd):
    """Restart streams outside the clean-resync domain: the emulation runs parallel over the places where the
    reference can resynchronise (csrc/hjd_seq.hip: k_seg_*), one lane per (candidate, block-in-MCU label); the one-lane kernel
    stays as the fallback (HJD_DBG_NO_SEG_EMULATION forces it).  Both must give the oracle's bytes, or both the oracle's refusal."""
    specs = [(320, 240, 1, 1, 2, 0.5, 4105), (320, 240, 1, 1, 3, 2.0, 4110), (320, 240, 2, 1, 1, 4.0, 4119), (320, 240, 1, 2, 3, 4.0, 4159),
             (250, 187, 2, 1, 3, 0.5, 4205), (250, 187, 4, 1, 3, 0.5, 4253), (512, 384, 1, 1, 2, 0.5, 4265), (512, 384, 2, 1, 7, 0.5, 4289),
             (512, 384, 2, 2, 2, 4.0, 4299), (512, 384, 1, 2, 2, 2.0, 4314), (512, 384, 2, 1, 2, 0.5, 4281), (250, 187, 1, 1, 1, 0.5, 4181),
             (1920, 1080, 2, 1, 1, 7.0, 4103), (640, 480, 2, 2, 1, 6.0, 4100), (256, 256, 2, 2, 5, 0.5, 4102)]   # found by search: up to 1895 early resyncs, 12 between a DC code and its bits; two the reference cannot decode
    jpgs = [T.synth_jpeg(w, h, seed, 85, hf, vf, rst, sg) for (w, h, hf, vf, rst, sg, seed) in specs]
    # markers where no encoder puts them: an extra RSTn pair in the middle of an interval, a pair of fill bytes before a marker
    extra = bytearray(jpgs[13])
    sos = extra.index(b"\xff\xda")
    at = sos + 14 + (len(extra) - sos) // 2
    while extra[at - 1] == 0xFF or extra[at] == 0xFF:
        at += 1
    extra[at:at] = b"\xff\xd3"
    jpgs.append(bytes(extra))
    fill = bytearray(jpgs[14])
    k = fill.index(b"\xff\xd0", fill.index(b"\xff\xda"))
    fill[k:k] = b"\xff\xff"
    jpgs.append(bytes(fill))
    want = [T.oracle_decode(j) for j in jpgs]
    assert sum(1 for w_ in want if w_[0] == 0 and w_[4].resyncs_dirty) >= 10 and sum(1 for w_ in want if w_[0] != 0) >= 2   # the oracle took the dirty branches
    res = {}
    for knob in (None, "1", "cut"):
        # "cut": every chain soon meets a cut-off segment: tried in parallel, then left to the one-lane kernel
        flags = hjd.dbg_seg_cut(2) if knob == "cut" else (hjd.DBG_NO_SEG_EMULATION if knob else 0)
        d = hjd.Decoder(0, debug_flags=flags)
        b = d.prepare(jpgs)
        b.run()
        b.run()   # a second run of the same batch: every per-run list is rebuilt on the device
        st = b.finish()
        outs = [b.download(k) if s == 0 else None for k, s in enumerate(st)]
        res[knob] = (st, outs, b.sequential_count(), b.segment_count())
        b.close(); d.close()
    for knob, (st, outs, nseq, nseg) in res.items():
        for k, (s, o, (orc, orgb, _, _, _)) in enumerate(zip(st, outs, want)):
            if orc == 0:
                assert s == 0 and np.array_equal(o, orgb), (knob, k, s)
            else:
                assert s == hjd.HJD_ERR_CORRUPT, (knob, k, s, orc)
        assert nseq >= 12   # per finish, not per run
        assert (nseg == 0) if knob == "1" else (nseg >= 12 if knob is None else nseg < nseq), (knob, nseq, nseg)
    assert res[None][2] == res["1"][2] == res["cut"][2]


def test_mixed_restart_batch(hjd, dec):
    """Config 5 in one batch: 4:4:4 / 4:2:2 / 4:2:0, with and without DRI, different sizes."""
    specs = [(320, 240, 1, 1, 0), (320, 240, 1, 1, 10), (400, 200, 2, 1, 0), (400, 200, 2, 1, 25),
             (250, 187, 2, 2, 4), (512, 384, 2, 2, 0), (96, 80, 1, 2, 2)]
    jpgs = [T.synth_jpeg(w, h, 300 + k, 85, hf, vf, rst, 7.0) for k, (w, h, hf, vf, rst) in enumerate(specs)]
    st, outs = dec.decode(jpgs)
    for j, s, o in zip(jpgs, st, outs):
        orc, orgb, _, _, ost = T.oracle_decode(j)
        if orc == 0:
            assert s == 0 and np.array_equal(o, orgb)
        else:
            assert s == hjd.HJD_ERR_CORRUPT


def test_extreme_aspect_ratios(hjd, dec):
    """Ragged shapes: one MCU row / one MCU column, widths and heights far from MCU multiples."""
    for (w, h, hf, vf) in [(4099, 9, 2, 2), (7, 3001, 2, 2), (2500, 1, 1, 1), (1, 777, 2, 1), (8192, 16, 2, 2)]:
        data = T.synth_jpeg(w, h, w + h, 85, hf, vf)
        st, outs = dec.decode([data], hjd.OUT_BMP)
        assert st == [0], (w, h, st)
        orc, orgb, _, _, _ = T.oracle_decode(data)
        assert orc == 0 and outs[0].tobytes() == T.oracle_bmp(orgb), (w, h)


def test_uniform_frames(hjd):
    """Uniform images: every MCU is the same few bits, the stream is periodic, and a guessed decode may lock into a wrong cycle
    for good (white and black 4:2:0 frames with the standard tables do: all subsequences start at one and the same unlucky bit
    phase).  The bytes must be the oracle's whatever the subsequence size; with one synchronisation workgroup per image the run of
    identical subsequences is resolved by moving the first ones' records along (sync_body), not by one lane walking the whole
    scan: no host re-synchronisation and a handful of refinement rounds (a few more where the decoder's state only comes round
    after several subsequences)."""
    frames = []
    for val in (255, 0, 200, 17):
        for hf, vf in ((2, 2), (1, 1), (2, 1)):
            frames.append(T.encode_jpeg(np.full((1080, 1920, 3), val, np.uint8), 85, hf, vf))
    for mode in (T.HUFF_FOUR_OPTIMISED, T.HUFF_SIX_OPTIMISED):  # one- and two-bit codes: MCUs of 6 or 12 bits, periods of three words
        for hf, vf in ((2, 2), (1, 1), (4, 1)):
            frames.append(T.encode_jpeg_ex(np.full((1080, 1920, 3), 255, np.uint8), 85, hf, vf, huff_mode=mode))
    half = T.synth_rgb(1920, 1080, 77, 6.0)
    half[300:900] = 255  # a uniform band between textured parts
    frames.append(T.encode_jpeg(half, 85, 2, 2))
    want = [T.oracle_decode(j)[1] for j in frames]
    for bits in (0, 256, 4096):
        d = hjd.Decoder(0, subseq_bits=bits)
        b = d.prepare(frames)
        b.run()
        st = b.finish()
        assert st == [0] * len(frames), (bits, st)
        for k in range(len(frames)):
            assert np.array_equal(b.download(k), want[k]), (bits, k)
        assert b.resync_count() == 0, bits   # also at 256 bits, where a 1080p scan is several synchronisation workgroups: a workgroup whose warm-up
                                             # subsequences are uniform waits for the one before it (round 5) instead of leaving the border to the host
        if bits != 256:
            assert max(b.sync_stats(k)["iters_max"] for k in range(len(frames))) <= 16, [b.sync_stats(k) for k in range(len(frames))]
        b.close()
        d.close()


def test_flat_content_under_optimised_tables(hjd):
    """Flat patches and bands under Huffman tables optimised for them (one- and two-bit codes): every bit is a
    code boundary and a guessed decode never falls into step with the true one — round 3 walked such a 4K frame with single lanes
    (24 / 13 / 5 / 9 ms of k_sync for the four frames below, 16-64 refinement rounds).  With the sibling trajectories of all 2 nb
    entering classes a slot entered at its first bit is adopted without decoding, and chain rounds run the truth down the
    workgroup (sync_body: entry_adopt): bytes equal the oracle's, no host re-synchronisation, a bounded number of refinement rounds and decodes."""
    W, H = 3840, 2160
    rng = np.random.default_rng(3)
    patches = np.zeros((H, W, 3), np.uint8)
    for y in range(0, H, H // 6):
        for x in range(0, W, W // 8):
            patches[y:y + H // 6, x:x + W // 8] = rng.integers(0, 256, 3)
    bands = np.zeros((H, W, 3), np.uint8)
    for k, y in enumerate(range(0, H, 120)):
        bands[y:y + 120] = (k * 37) % 256
    d = hjd.Decoder(0)
    for name, rgb in (("patches", patches), ("bands", bands)):
        for hf, vf in ((2, 2), (1, 1)):
            j = T.encode_jpeg_ex(rgb, 85, hf, vf, huff_mode=T.HUFF_FOUR_OPTIMISED)
            orc, orgb, _, _, _ = T.oracle_decode(j)
            assert orc == 0
            b = d.prepare([j])   # one image: the launch is small enough for all 2 nb - 1 siblings
            b.run()
            assert b.finish() == [0] and b.resync_count() == 0, (name, hf, vf)
            assert np.array_equal(b.download(0), orgb), (name, hf, vf)
            # (no wall-clock bound in a parity test: the work is bounded instead — refinement rounds against the cap of 96, subsequence
            # decodes against the scan's subsequences; times: profiles/r05_partly_uniform.txt)
            stt = b.sync_stats(0)
            assert stt["iters_max"] <= 64 and stt["decodes"] <= 3 * stt["nsub"], (name, hf, vf, stt)
            b.close()
    # An object on a black background, 4:4:4, tables optimised for the image: the uniform region after the object spans
    # a synchronisation workgroup's border, a guessed decode of a periodic stream locks into a wrong cycle and the warm-up
    # subsequences cannot settle it — round 4 needed a host re-synchronisation for every run of this image.  The workgroup now waits
    # for the one before it (sync_body: "a uniform region across a workgroup border"): no host round trip.
    tex = T.synth_rgb(W, H, 11, 8.0)
    obj = np.zeros((H, W, 3), np.uint8)
    obj[H // 3:2 * H // 3, W // 8:W // 2] = tex[H // 3:2 * H // 3, W // 8:W // 2]
    big_white = np.full((H, W, 3), 255, np.uint8)   # one uniform scan over several synchronisation workgroups at 256-bit subsequences
    for name, rgb, hf, vf, mode, bits in (("object on black", obj, 1, 1, T.HUFF_FOUR_OPTIMISED, 0), ("object on black", obj, 2, 2, T.HUFF_FOUR_OPTIMISED, 0),
                                          ("object on black", obj, 1, 1, T.HUFF_STD, 0), ("white", big_white, 2, 2, T.HUFF_STD, 256),
                                          ("white", big_white, 1, 1, T.HUFF_FOUR_OPTIMISED, 512)):
        j = T.encode_jpeg_ex(rgb, 85, hf, vf, huff_mode=mode)
        orc, orgb, _, _, _ = T.oracle_decode(j)
        assert orc == 0
        dd = hjd.Decoder(0, subseq_bits=bits)
        b = dd.prepare([j])
        for _ in range(2):
            b.run()
            assert b.finish() == [0]
        assert b.resync_count() == 0, (name, hf, vf, bits, b.sync_stats(0))
        assert np.array_equal(b.download(0), orgb), (name, hf, vf, bits)
        assert b.sync_stats(0)["iters_max"] <= 64
        b.close()
        dd.close()
    d.close()


def test_bit_flipped_scans_never_crash_and_never_lie(hjd, dec):
    """Corrupt entropy data: the GPU path must terminate, and whenever it reports success its bytes
    must be the oracle's (which restates what the reference computes on that exact stream); whenever the
    oracle says the reference would hang / overrun / lose a DC code, the GPU must report an error."""
    rng = np.random.default_rng(2024)
    base = [T.synth_jpeg(160, 120, 5, 85, 2, 2), T.synth_jpeg(120, 88, 6, 60, 1, 1), T.synth_jpeg(200, 64, 7, 90, 2, 1, 4, 7.0)]
    jpgs = []
    for k in range(45):
        b = bytearray(base[k % 3])
        sos = b.index(b"\xff\xda")
        for _ in range(1 + k % 3):
            pos = int(rng.integers(sos + 14, len(b) - 2))
            b[pos] ^= 1 << int(rng.integers(0, 8))
        jpgs.append(bytes(b))
    st, outs = dec.decode(jpgs)
    agree = 0
    for j, s, o in zip(jpgs, st, outs):
        orc, orgb, _, _, ost = T.oracle_decode(j)
        if s == 0:
            assert orc == 0, "GPU decoded a stream the reference cannot"
            assert np.array_equal(o, orgb)
            agree += 1
        if orc in (T.HJO_ERR_REF_WOULD_HANG, T.HJO_ERR_REF_OVERRUN, T.HJO_ERR_DC_CODE_MISSING):
            assert s != 0
    assert agree >= 5  # a good share of single-bit flips stay decodable (and must then match exactly)


def test_config3_batch_of_1024_1080p(hjd, dec):
    """BASELINE config 3 at its stated size: ONE batch of 1024 1920x1080 4:2:0 q85 JPEGs on one MI355X (about
    12 GB of HBM).  Six distinct sources: their outputs equal the oracle's and hash to what the compiled
    reference wrote (tests/golden/large_manifest.json); every replica is byte-identical to its source's output
    (determinism across the batch); a second run reproduces the first (idempotence)."""
    import zlib
    N = 1024
    src = [T.synth_jpeg(1920, 1080, 500 + k, 85, 2, 2) for k in range(6)]
    ref = [T.oracle_decode(j)[1] for j in src]
    ents = [T.large_golden(1920, 1080, 500 + k, 85, 2, 2, j) for k, j in enumerate(src)]
    jpgs = [src[i % 6] for i in range(N)]
    b = dec.prepare(jpgs)
    b.run()
    assert b.finish() == [0] * N
    first = []
    for i in range(N):
        o = b.download(i)
        if i < 6:
            assert np.array_equal(o, ref[i])
            if ents[i] is not None:
                assert hashlib.sha256(o.tobytes()).hexdigest() == ents[i]["rgb_sha256"]
        first.append(zlib.crc32(o.tobytes()))
    assert all(first[i] == first[i % 6] for i in range(N))
    b.run()
    assert b.finish() == [0] * N
    assert [zlib.crc32(b.download(i).tobytes()) for i in range(0, N, 37)] == first[0:N:37]
    assert b.resync_count() == 0
    b.close()


def test_large_image(hjd, dec):
    """64 Mpixel image (8192x8192, 4:2:0): 1.5 M blocks, ~90 k subsequences, ~16 M records in one image."""
    data = T.synth_jpeg(8192, 8192, 4242, 85, 2, 2)
    st, outs = dec.decode([data])
    assert st == [0]
    orc, orgb, _, _, _ = T.oracle_decode(data)
    assert orc == 0 and np.array_equal(outs[0], orgb)


def test_reference_main_binary_runs_on_the_gpu(hjd, golden, tmp_path):
    """The drop-in itself: the reference's OWN main.cpp, compiled in the build container and linked against libhjd.so in place of
    openjpg.cpp / loadjpg.cpp (oracle/Makefile: _ref/refmain_hjd), run here.  main.cpp:28-36 converts ../../../../data/Lenna.jpg
    into ../../../../data/out.bmp: the file it writes must be the BMP the compiled reference wrote (manifest hash; the two pad
    bytes per row the reference leaves to the stack are masked on both sides)."""
    import shutil
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "refmain_hjd")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/refmain_hjd is built in the build container only (make -C oracle ref)")
    d, man = golden
    data = tmp_path / "data"
    cwd = tmp_path / "a" / "b" / "c" / "d"
    data.mkdir()
    cwd.mkdir(parents=True)
    shutil.copy(os.path.join(d, "Lenna.jpg"), data / "Lenna.jpg")
    r = subprocess.run([exe], cwd=str(cwd), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    bmp = (data / "out.bmp").read_bytes()
    ent = man["Lenna.jpg"]
    assert len(bmp) == ent["bmp_bytes"]
    assert hashlib.sha256(bmp).hexdigest() in (ent["bmp_sha256"], ent.get("bmp_sha256_masked"))


def _mask_bmp_pad(bmp, w, h):
    """BMP row-pad bytes 2 and 3 zeroed: the reference's WriteBMP24 writes them from the stack (openjpg.cpp:563-567;
    tools/make_golden.py: bmp_sha256_masked)."""
    pad = (4 - (3 * w) % 4) % 4
    body = np.frombuffer(bmp, np.uint8)[54:].reshape(h, 3 * w + pad).copy()
    if pad > 1:
        body[:, 3 * w + 1:] = 0
    return bmp[:54] + body.tobytes()


@pytest.mark.parametrize("name", ["Lenna.jpg", "syn_256x192_420_q85_rst16.jpg", "syn_250x187_444_q85_rst4.jpg", "syn_250x187_420_q85_4opt_rst5.jpg",
                                  "flat_64x64_444_rst1_dirty.jpg", "syn_250x187_422_q85_3dqt_6huff.jpg"])
def test_reference_host_binary_runs_on_the_gpu(hjd, golden, tmp_path, name):
    """BASELINE north_star's drop-in at the HW/SW cut, executed: the reference's main.cpp AND openjpg.cpp — its file read, its
    JpegParseHeader (openjpg.cpp:371-496), its WriteBMP24 (:504-570) — compiled where they lie, unchanged, with JpegDecodeHW
    (loadjpg.h:180, called at openjpg.cpp:655) defined by hls-jpeg-decoder_amd/host/reference_decodehw.cpp through include/hjd_reference_shim.h on top of
    libhjd.so; loadjpg.cpp is not linked (oracle/Makefile: _ref/refhost_hjd).  main.cpp:28-36 converts ../../../../data/Lenna.jpg
    (whatever file sits there) into ../../../../data/out.bmp: the BMP the reference's own writer wrote must be the one the wholly
    compiled reference wrote.  The restart fixtures reach the library WITHOUT Ri (openjpg.cpp:441-446 keeps the segment length):
    clean ones through the parallel path, the dirty one through the emulation."""
    import shutil
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "refhost_hjd")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/refhost_hjd is built in the build container only (make -C oracle ref)")
    d, man = golden
    data = tmp_path / "data"
    cwd = tmp_path / "a" / "b" / "c" / "d"
    data.mkdir()
    cwd.mkdir(parents=True)
    shutil.copy(os.path.join(d, name), data / "Lenna.jpg")
    r = subprocess.run([exe], cwd=str(cwd), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "hjd:" not in r.stderr, r.stderr[-2000:]   # the library reports refusals there
    bmp = (data / "out.bmp").read_bytes()
    ent = man[name]
    assert len(bmp) == ent["bmp_bytes"]
    assert hashlib.sha256(_mask_bmp_pad(bmp, ent["width"], ent["height"])).hexdigest() == ent["bmp_sha256_masked"], name


@pytest.mark.parametrize("case", [(1920, 1080, 500, 85, 2, 2, 0, 6.0), (1920, 1080, 700, 85, 2, 2, 120, 6.0), (1920, 1080, 701, 85, 1, 1, 16, 6.0),
                                  (3840, 2160, 1003, 85, 2, 2, 0, 6.0)])
def test_reference_host_binary_big_capacity(hjd, tmp_path, case):
    """The same drop-in at the capacity the full-size inputs need (the three #defines of loadjpg.h:55-57 enlarged, as for
    oracle/_ref/ref_big; argv driver instead of main.cpp's fixed paths: _ref/refhost_hjd_big): one 1080p image, one 4K bench input,
    1080p with one MCU row per restart interval (clean) and 1080p 4:4:4 with 16-MCU intervals (the reference leaves the clean-resync
    domain four times on it) — the BMP its WriteBMP24 wrote hashes to what the wholly compiled reference wrote
    (tests/golden/large_manifest.json)."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "refhost_hjd_big")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/refhost_hjd_big is built in the build container only (make -C oracle ref)")
    w, h, seed, q, hf, vf, rst, sigma = case
    j = T.synth_jpeg(w, h, seed, q, hf, vf, rst, sigma)
    ent = T.large_golden(w, h, seed, q, hf, vf, j, rst)
    if ent is None:
        pytest.skip("regenerated input differs from the one the reference was run on (other libm)")
    jp, bp = tmp_path / "in.jpg", tmp_path / "out.bmp"
    jp.write_bytes(j)
    r = subprocess.run("ulimit -s unlimited; exec %s %s %s > /dev/null" % (exe, jp, bp), shell=True, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "hjd:" not in r.stderr, r.stderr[-2000:]
    assert hashlib.sha256(bp.read_bytes()).hexdigest() == ent["bmp_sha256"], case


def test_decode_hw_without_the_restart_interval(hjd, dec, golden):
    """The HW/SW cut as an unmodified reference host drives it: restart_interval > 0 (the DRI segment's length, openjpg.cpp:441-446)
    and restart_mcus = 0 — Ri never reaches JpegDecodeHW.  The library then takes the markers where the scan holds them (host count
    sizes the slabs; the final pass asks only what the reference's check asks, loadjpg.cpp:535-550): clean streams stay on the parallel
    path (no emulation, no host re-synchronisation), dirty ones go through the emulation; every output equals the oracle's / the
    compiled reference's."""
    d, man = golden
    names = ["syn_256x192_420_q85_rst16.jpg", "syn_250x187_444_q85_rst4.jpg", "syn_250x187_422_q85_rst1.jpg", "syn_250x187_420_q85_4opt_rst5.jpg",
             "flat_64x64_444_rst1_dirty.jpg"]
    jpgs = [open(os.path.join(d, n), "rb").read() for n in names]
    jpgs += [T.synth_jpeg(1920, 1080, 700, 85, 2, 2, 120, 6.0), T.synth_jpeg(1920, 1080, 701, 85, 1, 1, 16, 6.0),
             T.synth_jpeg(640, 480, 77 + 1, 85, 2, 2, 1, 6.0), T.synth_jpeg(128, 128, 5, 85, 2, 2, 0xFFFF, 6.0)]
    extra = bytearray(jpgs[5])   # an RSTn pair where no encoder puts one: in the middle of an interval (the reference resynchronises there)
    sos = extra.index(b"\xff\xda")
    at = sos + 14 + (len(extra) - sos) // 2
    while extra[at - 1] == 0xFF or extra[at] == 0xFF:
        at += 1
    extra[at:at] = b"\xff\xd3"
    jpgs.append(bytes(extra))
    want = [T.oracle_decode(j) for j in jpgs]
    jds = []
    for j in jpgs:
        st, jd = hjd.parse_header(j)
        assert st == 0 and jd.restart_interval > 0 and jd.restart_mcus > 0
        jd.restart_mcus = 0
        jds.append(jd)
    L = hjd.lib()
    # (a) the single-image entry point, as the shim calls it
    for k, (j, jd, (orc, orgb, _, _, ost)) in enumerate(zip(jpgs, jds, want)):
        out = np.zeros((jd.height, jd.width, 3), np.uint8)
        rc = L.hjd_jpeg_decode_hw(ctypes.byref(jd), jd.height, jd.width, jd.hfactor, jd.vfactor, out.ctypes.data)
        if orc == 0:
            assert rc == 0 and np.array_equal(out, orgb), (k, rc)
        else:
            assert rc == hjd.HJD_ERR_CORRUPT, (k, rc, orc)
    for n, o in zip(names, range(len(names))):
        assert hashlib.sha256(want[o][1].tobytes()).hexdigest() == man[n]["rgb_sha256"]
    # (b) the batch-level twin: which path every image took
    clean = [k for k, w_ in enumerate(want) if w_[0] == 0 and w_[4].resyncs_dirty == 0]
    dirty = [k for k, w_ in enumerate(want) if w_[0] == 0 and w_[4].resyncs_dirty != 0]
    assert len(clean) >= 6 and len(dirty) >= 2, (clean, dirty)
    for subset, expect_seq in ((clean, 0), (dirty, len(dirty)), (list(range(len(jpgs))), None)):
        b = dec.prepare_parsed([jds[k] for k in subset], keep_alive=jpgs)
        b.run()
        st = b.finish()
        for i, k in enumerate(subset):
            if want[k][0] == 0:
                assert st[i] == 0 and np.array_equal(b.download(i), want[k][1]), (k, st[i])
            else:
                assert st[i] == hjd.HJD_ERR_CORRUPT, (k, st[i])
        assert b.resync_count() == 0
        if expect_seq is not None:
            assert b.sequential_count() == expect_seq, (subset, b.sequential_count())
        if expect_seq:
            assert b.segment_count() == expect_seq   # parallel over the resynchronisation points, not one lane per image
        b.close()


def test_reference_shaped_entry_points(hjd, golden):
    """The three single-image entry points that mirror the reference's API surface:
    hjd_jpeg_decode_hw (JpegDecodeHW, loadjpg.h:180 — parsed tables in, RGB out, argument order height, width),
    hjd_decode_jpg_file_data (DecodeJpgFileData, openjpg.h:25-27) and hjd_convert_jpg_file (ConvertJpgFile)."""
    import ctypes
    d, man = golden
    L = hjd.lib()
    for name in ("Lenna.jpg", "syn_250x187_422_q85.jpg", "pil_250x187_420_q85_opt.jpg", "syn_256x192_420_q85_rst16.jpg"):
        data = open(os.path.join(d, name), "rb").read()
        orc, orgb, _, _, _ = T.oracle_decode(data)
        st, jd = hjd.parse_header(data)
        assert st == 0
        out = np.zeros((jd.height, jd.width, 3), np.uint8)
        rc = L.hjd_jpeg_decode_hw(ctypes.byref(jd), jd.height, jd.width, jd.hfactor, jd.vfactor, out.ctypes.data)
        assert rc == 0, (name, rc)
        assert np.array_equal(out, orgb), name
        # memory in, callee-allocated RGB out
        rgb, w, h = ctypes.c_void_p(), ctypes.c_uint(), ctypes.c_uint()
        assert L.hjd_decode_jpg_file_data(data, len(data), ctypes.byref(rgb), ctypes.byref(w), ctypes.byref(h)) == 1
        got = np.ctypeslib.as_array(ctypes.cast(rgb, ctypes.POINTER(ctypes.c_uint8)), shape=(h.value, w.value, 3)).copy()
        L.hjd_free(rgb)
        assert np.array_equal(got, orgb), name
    # hFactor = 0 is a SIGFPE in the reference (loadjpg.cpp:1153); here it is a status
    assert L.hjd_jpeg_decode_hw(ctypes.byref(jd), jd.height, jd.width, 0, 1, out.ctypes.data) == hjd.HJD_ERR_UNSUPPORTED
    rgb, w, h = ctypes.c_void_p(), ctypes.c_uint(), ctypes.c_uint()
    assert L.hjd_decode_jpg_file_data(b"garbage!", 8, ctypes.byref(rgb), ctypes.byref(w), ctypes.byref(h)) == 0
    assert L.hjd_last_status() == hjd.HJD_ERR_NOT_JPEG


@pytest.mark.gpu
def test_four_caller_streams_mixed_batches(hjd):
    """Four batches of one decoder on four caller streams, enqueued step after step without a host synchronisation in between
    (BASELINE config 5's shape in the small): large and small images, restart streams inside and outside the clean-resync domain.
    With three or more caller streams in flight the library keeps the small images' kernels on the run's own stream; the
    emulation of the dirty restart streams is enqueued by the FIRST finish of a batch's contents and by the runs after it
    (hjd_api.cpp: emulation_enqueue) — the outputs of a batch that is run once, of one that is run again after its finish, and of
    one finished after several runs must all equal the oracle's, and the emulation counters must say which images took it."""
    import torch
    torch.zeros(1, device="cuda:0")
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(3)]
    d = hjd.Decoder(0)
    specs = [(1000, 700, 2, 2, 0, 7.0, 4100), (250, 187, 2, 1, 0, 7.0, 4101), (640, 480, 1, 1, 40, 7.0, 4102), (64, 48, 2, 2, 0, 7.0, 4103),
             (512, 384, 2, 1, 7, 0.5, 4289), (640, 480, 2, 2, 1, 6.0, 4100), (1280, 720, 2, 2, 80, 7.0, 4106), (96, 80, 4, 1, 0, 7.0, 4107),
             (320, 240, 1, 1, 2, 0.5, 4105)]   # three of them dirty (test_segment_parallel_emulation_equals_the_one_lane_kernel's)
    jpgs = [T.synth_jpeg(w, h, seed, 85, hf, vf, rst, sg) for (w, h, hf, vf, rst, sg, seed) in specs]
    want = [T.oracle_decode(j) for j in jpgs]
    batches = [d.prepare([jpgs[(k + q) % len(jpgs)] for q in range(6)]) for k in range(4)]
    order = [[(k + q) % len(jpgs) for q in range(6)] for k in range(4)]

    def check(b, idx, st):
        for i, g in enumerate(idx):
            orc, orgb = want[g][0], want[g][1]
            if orc == 0:
                assert st[i] == 0, (specs[g], st[i])
                assert np.array_equal(b.download(i), orgb), specs[g]
            else:
                assert st[i] != 0, specs[g]

    # run once, finish: the first finish enqueues the emulation for the contents that need it
    for b, ts in zip(batches, streams):
        b.run(ts.cuda_stream)
    for b, ts, idx in zip(batches, streams, order):
        check(b, idx, b.finish(ts.cuda_stream))
    dirty = sum(b.sequential_count() for b in batches)
    # three steps back to back on the four streams, then the finishes: these runs carry the emulation themselves
    for _ in range(3):
        for b, ts in zip(batches, streams):
            b.run(ts.cuda_stream)
    for b, ts, idx in zip(batches, streams, order):
        check(b, idx, b.finish(ts.cuda_stream))
    # the images on which the oracle took the dirty branches and decoded: emulated here, once per batch that holds them and finish
    n_dirty = sum(1 for idx in order for g in idx if want[g][0] == 0 and want[g][4].resyncs_dirty)
    assert n_dirty >= 4 and dirty == n_dirty, (dirty, n_dirty)
    assert sum(b.sequential_count() for b in batches) == 2 * n_dirty   # ... and once more for the second finish
    assert all(b.resync_count() == 0 for b in batches)
    for b in batches:
        b.close()
    d.close()


@pytest.mark.parametrize("streams", [2, 3, 7])
def test_internal_streams(hjd, golden, streams):
    """hjd_options.streams: the batch is split into contiguous image ranges on internal HIP streams,
    forked from and joined to the caller's stream.  Same bytes, any split (including more streams than
    images can fill evenly, failed images in between, restart and non-restart images mixed)."""
    d = hjd.Decoder(0, streams=streams)
    fx = _ok_fixtures(golden, restart=("none", "clean"))
    jpgs = [f[2] for f in fx] + [b"junk", T.synth_jpeg(1000, 800, 77, 85, 2, 2), T.synth_jpeg(640, 480, 78, 85, 1, 1, 9, 7.0)]
    for rep in range(2):
        st, outs = d.decode(jpgs)
        for j, s, o in zip(jpgs, st, outs):
            if j == b"junk":
                assert s == hjd.HJD_ERR_NOT_JPEG
                continue
            orc, orgb, _, _, ost = T.oracle_decode(j)
            assert orc == 0 and s == 0 and np.array_equal(o, orgb)
    # also on a non-default caller stream
    import torch
    side = torch.cuda.Stream()
    b = d.prepare(jpgs[:5])
    b.run(side.cuda_stream)
    assert b.finish(side.cuda_stream) == [0] * 5
    for i in range(5):
        assert np.array_equal(b.download(i), T.oracle_decode(jpgs[i])[1])
    b.close()
    d.close()


def test_colour_stage_exhaustive(hjd, dec):
    """ConvertYCrCbtoRGB + Clamp (loadjpg.cpp:867-880, :83-91) over ALL 2^24 (Y, Cb, Cr) triples, computed
    on the device by the functions the decode kernel uses (hjd_test_color_table), against the oracle's formula.  The
    numpy restatement used for the bulk comparison is itself pinned to the oracle's C function on 20 000 samples."""
    import ctypes
    got = dec.color_table()
    f = np.float32
    cb = np.arange(256, dtype=np.int32)[:, None]
    cr = np.arange(256, dtype=np.int32)[None, :]
    fcb, fcr = (cb - 128).astype(f), (cr - 128).astype(f)
    a, b, c, d = f(1.402) * fcr, f(0.34414) * fcb, f(0.71414) * fcr, f(1.772) * fcb

    def expect(y):
        fy = f(y)
        r = np.broadcast_to(fy + a, (256, 256))
        g = (fy - b) - c
        bl = np.broadcast_to(fy + d, (256, 256))
        return np.stack([np.clip(v.astype(np.int32), 0, 255).astype(np.uint8) for v in (r, g, bl)], axis=-1)

    o = T.oracle_lib()
    o.hjo_ycc_to_rgb.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
    rng = np.random.default_rng(1)
    px = np.zeros(3, np.uint8)
    exp_cache = {}
    for y, u, v in rng.integers(0, 256, (20000, 3)):
        o.hjo_ycc_to_rgb(int(y), int(u), int(v), px.ctypes.data, px.ctypes.data + 1, px.ctypes.data + 2)
        if int(y) not in exp_cache:
            exp_cache[int(y)] = expect(int(y))
        assert tuple(exp_cache[int(y)][u, v]) == tuple(px), (y, u, v)
    bad = 0
    for y in range(256):
        bad += int((got[y] != (exp_cache.get(y) if y in exp_cache else expect(y))).any(axis=-1).sum())
    assert bad == 0, "%d of 16777216 triples differ" % bad


def test_overflow_stress_blocks_through_the_back_end(hjd, dec):
    """chosen coefficient blocks -> tiles on the device.  Blocks with magnitudes up to 32767 and
    quantisers up to 255 (products that wrap int16, sums that saturate Clamp), dense and sparse, fed to the
    back end by coding exactly those coefficients into a scan (tools/jpeg_enc.c, explicit-coefficient mode)."""
    rng = np.random.default_rng(8)
    w, h = 256, 128                      # 4:4:4 -> 512 MCUs x 3 blocks
    nb = (w // 8) * (h // 8) * 3
    co = np.zeros((nb, 64), np.int16)
    for b in range(nb):
        k = int(rng.integers(0, 64))
        idx = rng.choice(np.arange(1, 64), min(k, 63), replace=False)
        scale = [8, 200, 2047, 32767][b % 4]
        co[b, idx] = rng.integers(-scale, scale + 1, len(idx))
    co[:, 0] = rng.integers(-30000, 30001, nb)
    qt = rng.integers(1, 256, (3, 64)).astype(np.uint8)
    j = T.encode_jpeg_ex(None, 85, 1, 1, 0, True, T.HUFF_SIX_OPTIMISED, qtab=qt, coefs=co, size=(w, h))
    orc, orgb, ocoef, _, _ = T.oracle_decode(j, want_coefs=True)
    assert orc == 0 and np.array_equal(ocoef, co)          # the scan really carries the chosen blocks
    st, outs = dec.decode([j])
    assert st == [0] and np.array_equal(outs[0], orgb)
    b = dec.prepare([j])
    b.run()
    assert b.finish() == [0]
    assert np.array_equal(b.download_coefs(0, nb), co)
    b.close()


def test_pipelined_host_path_is_byte_identical(hjd, golden):
    """hjd_decode_batch_host (header walk / H2D / kernels / D2H of successive chunks overlapped on three streams, ring of
    re-used batch objects): same bytes as the resident-batch path, for pinned and for pageable destinations, RGB and
    BMP, with failing images in the middle, over several chunks, and again on the re-used ring."""
    d, man = golden
    fx = _ok_fixtures(golden, restart=("none", "clean"))
    big = [T.synth_jpeg(1920, 1080, 600 + k, 85, 2, 2) for k in range(4)] + [T.synth_jpeg(3840, 2160, 1000, 85, 2, 2)]
    jpgs = [f[2] for f in fx] + [b"junk"] + big * 6 + [open(os.path.join(d, "pil_gray.jpg"), "rb").read()] + big * 3
    dd = hjd.Decoder(0)
    want = {}
    for fmt in (hjd.OUT_RGB, hjd.OUT_BMP):
        st0, ref = dd.decode(jpgs, fmt)
        for pinned in (True, False, True):
            st, outs = dd.decode_host(jpgs, fmt, pinned=pinned)
            assert st == st0
            for j, s, o, r in zip(jpgs, st, outs, ref):
                if s == 0:
                    assert np.array_equal(np.asarray(o), r.reshape(-1)), (fmt, pinned, len(j))
    assert st0[len(fx)] == hjd.HJD_ERR_NOT_JPEG and sum(1 for s in st0 if s) == 2
    dd.close()


def test_two_ranks_share_one_batch(hjd, tmp_path):
    """The N > 1 code path on the hardware at hand: two ranks (own processes, gloo on 127.0.0.1, both on device 0)
    decode the two shards shard.partition() gives them of ONE 64-image mixed batch.  Union = every image exactly
    once, every image bit-exact against the oracle."""
    import json
    import socket
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import rank_worker
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, rank_worker.__file__, "0", str(tmp_path / ("r%d.json" % r))], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = {}
    for r in range(world):
        part = json.load(open(tmp_path / ("r%d.json" % r)))
        assert not (set(part) & set(got)), "an image was decoded by both ranks"
        got.update(part)
        assert 24 <= len(part) <= 40  # balanced by compressed size, not by count
    jpgs = rank_worker.mixed_batch()
    assert sorted(int(k) for k in got) == list(range(len(jpgs)))
    for g, j in enumerate(jpgs):
        orc, orgb, _, _, ost = T.oracle_decode(j)
        if orc == 0:
            assert got[str(g)] == hashlib.sha256(orgb.tobytes()).hexdigest(), g
        else:
            assert got[str(g)].startswith("status"), g


@pytest.mark.parametrize("world", [2, 4])
def test_bench_launches_its_own_ranks(tmp_path, world):
    """`python bench.py --gpus N` without a launcher: the parent spawns the ranks itself; here all on device 0
    over gloo (rehearsal knobs), with BASELINE config 4's image shape (1080p 4:2:0 q85).  N = 4 is the largest world size this
    pool allows on one card (at most 6 processes may use it, this test process being one); the 8-rank plumbing is covered
    without a GPU by tests/test_shard_gloo.py::test_eight_rank_rehearsal_of_config4_plumbing.  The line must say n_gpus N,
    count every rank's pixels, and every rank checks its own outputs against reference hashes / the oracle (bench.py exits
    non-zero on a mismatch, and the parent on any rank's failure)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HJD_BENCH_BACKEND="gloo", HJD_BENCH_FORCE_DEVICE="0")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
                        "--batch", "8", "--width", "1920", "--height", "1080", "--distinct", "2", "--streams", "2",
                        "--no-cpu-baseline", "--no-end-to-end"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == world and line["config"]["images_per_gpu"] == 8 and line["parity"] == "ok"
    assert [p["rank"] for p in line["per_rank"]] == list(range(world)) and all(0 < p["ms_per_step"] <= line["ms_per_step"] + 1e-3 for p in line["per_rank"])
    px = 8 * world * 1920 * 1080
    assert abs(line["value"] * 1e6 * line["ms_per_step"] * 1e-3 - px) / px < 0.02


def test_bench_single_rank_over_rccl():
    """The RCCL branch of the multi-rank plumbing, executed on the one GPU a builder's box has: a process group of ONE rank over
    backend "nccl" (= RCCL; HJD_BENCH_GROUP_AT_ONE=1), i.e. the eager communicator on the rank's device, the barriers around the
    timed region, the all-reduce of ones, the all-gather of the per-rank times and the MAX / SUM agreement all run through RCCL
    with tensors on the device — exactly the calls N > 1 ranks make.  The line must say backend nccl and outputs must be right."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HJD_BENCH_GROUP_AT_ONE="1", HJD_BENCH_BACKEND="nccl", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--batch", "8", "--width", "1920", "--height", "1080", "--distinct", "2", "--streams", "2",
                        "--no-cpu-baseline", "--no-end-to-end"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    if line["backend"] != "nccl" and "RCCL init failed" in r.stderr:  # a box on which RCCL cannot come up at all (shard.init_group fell back to gloo)
        pytest.skip("RCCL could not be brought up on this box: " + r.stderr[-300:])
    assert line["backend"] == "nccl", (line["backend"], r.stderr[-2000:])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["parity"] == "ok"
    assert [p["rank"] for p in line["per_rank"]] == [0] and abs(line["per_rank"][0]["ms_per_step"] - line["ms_per_step"]) < 1e-2


def test_two_host_threads_share_one_decoder(hjd, dec, golden):
    """SURVEY §8b asks for a thread-safe ABI: two host threads drive ONE decoder, each with batches of its own on a stream
    of its own (prepare / run / finish / download interleave freely; ctypes releases the GIL inside the calls), while a third
    hammers the single-image entry point, which serialises on the library's one cached workspace.  Everything bit-exact."""
    import threading
    import torch
    fx = _ok_fixtures(golden)
    sets = [fx[0::2], fx[1::2]]
    errors = []

    def worker(k):
        try:
            stream = torch.cuda.Stream()
            for rep in range(6):
                mine = sets[k][rep % 3:] + sets[k][:rep % 3]
                b = dec.prepare([f[2] for f in mine])
                b.run(stream.cuda_stream)
                st = b.finish(stream.cuda_stream)
                for i, (name, ent, data) in enumerate(mine):
                    if st[i] != 0 or hashlib.sha256(b.download(i).tobytes()).hexdigest() != ent["rgb_sha256"]:
                        errors.append((k, rep, name, st[i]))
                b.close()
        except Exception as e:  # noqa
            errors.append((k, repr(e)))

    def single(k):
        try:
            name, ent, data = fx[k]
            for rep in range(4):
                rc, jd = hjd.parse_header(data)
                out = np.empty(jd.width * jd.height * 3, np.uint8)
                rc = hjd.lib().hjd_jpeg_decode_hw(ctypes.byref(jd), jd.height, jd.width, jd.hfactor, jd.vfactor, out.ctypes.data)
                if rc != 0 or hashlib.sha256(out.tobytes()).hexdigest() != ent["rgb_sha256"]:
                    errors.append(("single", name, rc))
        except Exception as e:  # noqa
            errors.append(("single", repr(e)))

    th = [threading.Thread(target=worker, args=(0,)), threading.Thread(target=worker, args=(1,)),
          threading.Thread(target=single, args=(0,)), threading.Thread(target=single, args=(3,))]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
        assert not t.is_alive()
    assert not errors, errors[:5]


def test_empty_and_all_bad_batches(hjd, dec):
    """Ragged inputs at the batch level: no images, and only images that fail to parse."""
    st, outs = dec.decode([])
    assert st == [] and outs == []
    st, outs = dec.decode([b"", b"\xff\xd8", b"not a jpeg"])
    assert all(s != 0 for s in st) and outs == [None, None, None]


def test_statuses(hjd, dec, golden):
    d, man = golden
    bad = [open(os.path.join(d, n), "rb").read() for n in ("pil_progressive.jpg", "pil_gray.jpg")]
    lenna = open(os.path.join(d, "Lenna.jpg"), "rb").read()
    trunc = lenna[:50000]          # scan cut short
    notjpg = b"hello world, not a jpeg"
    st, outs = dec.decode(bad + [trunc, notjpg, lenna])
    assert st[0] == hjd.HJD_ERR_UNSUPPORTED and st[1] == hjd.HJD_ERR_UNSUPPORTED
    assert st[2] == hjd.HJD_ERR_CORRUPT
    assert st[3] == hjd.HJD_ERR_NOT_JPEG
    assert st[4] == 0 and outs[4] is not None


# ---- conformant mode (opt-in, outside the parity contract) --------------------------------------

UNZZ = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
        35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def _conformant_model(hjd, dec, data, ri):
    """float64 model of what a conforming baseline decoder with nearest-neighbour chroma computes, from the
    coefficients the (default-mode, reference-exact) entropy stage delivers: DC predictors restarted at every
    interval, every component dequantised with its own table, exact IDCT, rounding, JFIF colour constants."""
    st, jd = hjd.parse_header(data)
    assert st == 0
    w, h, hf, vf = jd.width, jd.height, jd.hfactor, jd.vfactor
    mx, my = -(-w // (8 * hf)), -(-h // (8 * vf))
    nby = hf * vf
    nb = nby + 2
    b = dec.prepare([data])
    b.run()
    assert b.finish() == [0]
    co = b.download_coefs(0, mx * my * nb).astype(np.int64).reshape(mx * my, nb, 64)
    b.close()
    if ri:  # the reference carries the predictors across intervals; undo that per component
        comp_last = {0: nby - 1, 1: nby, 2: nby + 1}
        dc = co[:, :, 0].copy()
        for m in range(mx * my):
            s0 = m - m % ri
            if s0:
                for k in range(nb):
                    c = 0 if k < nby else k - nby + 1
                    co[m, k, 0] = ((dc[m, k] - dc[s0 - 1, comp_last[c]] + 32768) % 65536) - 32768
    qt = [np.array(list(jd.qtable[c]), np.float64) for c in (1, 2, 3)]
    x = np.arange(8)
    cs = np.cos((2 * x[:, None] + 1) * x[None, :] * np.pi / 16)       # [x][u]
    cu = np.where(x == 0, np.sqrt(0.5), 1.0)
    planes = [np.zeros((my * 8 * vf, mx * 8 * hf)), np.zeros((my * 8, mx * 8)), np.zeros((my * 8, mx * 8))]
    for m in range(mx * my):
        r, c0 = divmod(m, mx)
        for k in range(nb):
            c = 0 if k < nby else k - nby + 1
            F = np.zeros(64)
            F[UNZZ] = co[m, k] * qt[c]                                   # natural order v*8+u
            F = F.reshape(8, 8)                                          # [v][u]
            t = 0.25 * np.einsum("yv,v,vu,u,xu->yx", cs, cu, F, cu, cs)
            px = np.clip(np.floor(t + 128.5), 0, 255)
            if c == 0:
                by, bx = divmod(k, hf)
                planes[0][r * 8 * vf + by * 8:r * 8 * vf + by * 8 + 8, c0 * 8 * hf + bx * 8:c0 * 8 * hf + bx * 8 + 8] = px
            else:
                planes[c][r * 8:r * 8 + 8, c0 * 8:c0 * 8 + 8] = px
    Y = planes[0][:h, :w]
    Cb = np.repeat(np.repeat(planes[1], vf, 0), hf, 1)[:h, :w] - 128.0
    Cr = np.repeat(np.repeat(planes[2], vf, 0), hf, 1)[:h, :w] - 128.0
    rgb = np.stack([Y + 1.402 * Cr, Y - 0.344136 * Cb - 0.714136 * Cr, Y + 1.772 * Cb], -1)
    return np.clip(np.floor(rgb + 0.5), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("w,h,hf,vf,ri,three", [(250, 187, 2, 2, 0, False), (320, 240, 2, 2, 5, True), (200, 120, 1, 1, 7, True),
                                                   (333, 100, 2, 1, 1, False)])
def test_conformant_mode_matches_a_float64_model(hjd, dec, w, h, hf, vf, ri, three):
    """hjd_options.conformant (outside the parity contract, never the default): DC predictors reset at
    RSTn, each component's own quantiser, true-cosine rounding IDCT, JFIF colour with rounding.  Within +-1 of a float64
    model (fp32 vs fp64 ties); and the DEFAULT decoder's bytes for the same input still equal the oracle's."""
    data = T.encode_jpeg_ex(T.synth_rgb(w, h, 900 + ri, 6.0), 85, hf, vf, ri, three, T.HUFF_STD)
    conf = hjd.Decoder(0, conformant=1)
    st, outs = conf.decode([data])
    conf.close()
    assert st == [0]
    model = _conformant_model(hjd, dec, data, ri)
    diff = np.abs(outs[0].astype(np.int32) - model.astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.02, (int(diff.max()), float((diff > 0).mean()))
    st, outs = dec.decode([data])
    orc, orgb, _, _, _ = T.oracle_decode(data)
    assert st == [0] and orc == 0 and np.array_equal(outs[0], orgb)


def test_conformant_mode_agrees_with_libjpeg_on_444_restart_streams(hjd):
    """The same image through Pillow (libjpeg): no chroma upsampling in 4:4:4, so only IDCT / colour rounding differ
    (a few levels at most), while the reference-exact default mode is off by tens of levels on a DRI stream because
    the reference never resets its DC predictors."""
    PIL = pytest.importorskip("PIL.Image")
    import io
    data = T.synth_jpeg(320, 200, 77, 90, 1, 1, 6, 6.0)
    lib = np.asarray(PIL.open(io.BytesIO(data)).convert("RGB")).astype(np.int32)
    conf = hjd.Decoder(0, conformant=1)
    st, outs = conf.decode([data])
    conf.close()
    assert st == [0]
    d = np.abs(outs[0].astype(np.int32) - lib)
    assert d.max() <= 4 and d.mean() < 0.7, (int(d.max()), float(d.mean()))
    ref = hjd.Decoder(0)
    st, outs = ref.decode([data])
    ref.close()
    assert st == [0] and np.abs(outs[0].astype(np.int32) - lib).mean() > 5.0


def test_conformant_mode_takes_16_bit_dc_codes(hjd, dec, golden):
    """A DC Huffman table with a 16-bit code: the reference's search stops at 15 bits (loadjpg.cpp:562), so the default
    mode refuses the file; the conformant mode decodes it.  The fixture's DC table is edited in place: the code of its
    last symbol (never used by this image) moves from 9 to 16 bits, which leaves every other code unchanged."""
    d, man = golden
    data = bytearray(open(os.path.join(d, "syn_250x187_420_q85.jpg"), "rb").read())
    k = data.index(b"\xff\xc4")
    assert data[k + 4] == 0x00 and data[k + 5 + 8] == 1 and data[k + 5 + 15] == 0   # DC table 0: one 9-bit code, no 16-bit code
    data[k + 5 + 8] = 0
    data[k + 5 + 15] = 1
    data = bytes(data)
    st, outs = dec.decode([data])
    assert st == [hjd.HJD_ERR_UNSUPPORTED]
    conf = hjd.Decoder(0, conformant=1)
    st, outs = conf.decode([data, open(os.path.join(d, "syn_250x187_420_q85.jpg"), "rb").read()])
    conf.close()
    assert st == [0, 0] and np.array_equal(outs[0], outs[1])


def test_byte_stuffing_dense_streams(hjd, dec):
    """Noise at high quality: an 0xFF (hence a stuffed 0x00, or with DRI a marker) in about every tenth 16-byte lane of the unstuff
    kernels; FF as the last byte of a lane / of a 4 KiB chunk, stuffed zeros as the first.  The unstuff kernels treat lanes without
    an FF wholesale and work the others off a list: every decision of that split is exercised here, in one batch whose images
    start at unrelated chunk phases, and checked against the oracle."""
    rng = np.random.default_rng(5)
    jpgs = []
    for k, (w, h, hf, vf, rst) in enumerate([(257, 131, 1, 1, 0), (640, 360, 2, 2, 0), (333, 222, 2, 1, 0), (512, 96, 1, 1, 3),
                                            (640, 368, 2, 2, 40), (97, 411, 1, 2, 0), (48, 48, 1, 1, 0), (1024, 512, 2, 2, 0)]):
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        jpgs.append(T.encode_jpeg(rgb, 98 if k % 2 == 0 else 92, hf, vf, rst))
    assert sum(j.count(b"\xff\x00") for j in jpgs) > 4000
    # stuffed pairs at every phase of a 16-byte lane, including split across two lanes (FF last, 00 first)
    phases = set()
    for j in jpgs:
        sos = j.index(b"\xff\xda")
        scan0 = sos + 2 + int.from_bytes(j[sos + 2:sos + 4], "big")
        q = j.find(b"\xff\x00", scan0)
        while q >= 0:
            phases.add((q - scan0) % 16)
            q = j.find(b"\xff\x00", q + 2)
    assert phases == set(range(16))
    st, outs = dec.decode(jpgs)
    assert st == [0] * len(jpgs), st
    for k, j in enumerate(jpgs):
        orc, orgb, _, _, _ = T.oracle_decode(j)
        assert orc == 0 and np.array_equal(outs[k], orgb), k


def test_random_corpus(hjd, dec):
    """Property test over the parameter space the parser admits: 240 seeded random images (1..400 pixels a side, every
    sampling with hf*vf <= 4, quality 15..100, smooth to noisy content, no / short / long restart intervals, two or three
    DQTs, standard / permuted / optimised Huffman tables), decoded in three batches (ragged mixes of table sets and sizes).
    Wherever the oracle — the reference's algorithm — produces an image the HIP path must produce the same bytes; where
    the reference would not terminate or overrun, it must report an error."""
    rng = np.random.default_rng(20251004)
    samplings = [(1, 1), (2, 1), (1, 2), (2, 2), (4, 1), (1, 4)]
    jpgs = []
    for k in range(240):
        w, h = int(rng.integers(1, 401)), int(rng.integers(1, 401))
        hf, vf = samplings[int(rng.integers(0, len(samplings)))]
        q = int(rng.integers(15, 101))
        sigma = float(rng.choice([0.0, 0.5, 3.0, 8.0, 30.0]))
        nm = -(-w // (8 * hf)) * -(-h // (8 * vf))
        rst = int(rng.choice([0, 0, 1, int(rng.integers(1, nm + 2)), -(-w // (8 * hf))]))
        rgb = T.synth_rgb(w, h, 9000 + k, sigma)
        jpgs.append(T.encode_jpeg_ex(rgb, q, hf, vf, rst, three_dqt=bool(rng.integers(0, 2)),
                                     huff_mode=int(rng.choice([T.HUFF_STD, T.HUFF_SIX_PERMUTED, T.HUFF_SIX_OPTIMISED, T.HUFF_FOUR_OPTIMISED]))))
    ok = bad = 0
    for lo in range(0, len(jpgs), 80):
        part = jpgs[lo:lo + 80]
        st, outs = dec.decode(part)
        for k, j in enumerate(part):
            orc, orgb, _, _, _ = T.oracle_decode(j)
            if orc == 0:
                assert st[k] == 0 and np.array_equal(outs[k], orgb), (lo + k, st[k])
                ok += 1
            else:
                assert st[k] != 0, (lo + k, orc)
                bad += 1
    assert ok >= 180 and bad >= 10, (ok, bad)
