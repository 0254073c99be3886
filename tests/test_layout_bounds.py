"""Every slab a kernel touches is as large as the kernel's own indexing needs — checked on the CPU, for ragged
sizes, from the layout hjd_batch_prepare would produce (hjd_layout_query, a dry run of the same code).

Background: a development build of round 1 aborted inside hipStreamSynchronize (a GPU memory
fault) while the transposed bit stream was being introduced: the staging loops of the entropy kernels read WHOLE
64-row groups for every wave of a workgroup, while the slab had been sized from the number of subsequences alone
(DESIGN.md §8).  This test restates, kernel by kernel, the largest element each access pattern can reach and
compares it with what is allocated.  The formulas below follow csrc/*.hip; a change of indexing there without a
matching change of the allocation fails here, without a GPU."""
import ctypes
import importlib
import itertools
import os

import pytest

import hjdtools as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hjd():
    return importlib.import_module("hls-jpeg-decoder_amd")


@pytest.fixture(scope="module")
def header(hjd):
    j = T.synth_jpeg(64, 48, 1, 85, 2, 2)
    st, jd = hjd.parse_header(j)
    assert st == 0
    return j, jd


def query(hjd, jd, w, h, hf, vf, scan_len, S, overlap, rst_mcus=0, fmt=0):
    q = hjd.JpegData()
    ctypes.memmove(ctypes.byref(q), ctypes.byref(jd), ctypes.sizeof(q))
    q.width, q.height, q.hfactor, q.vfactor, q.stream_len = w, h, hf, vf, scan_len
    q.restart_interval, q.restart_mcus = (4, rst_mcus) if rst_mcus else (0, 0)
    o = hjd.LayoutInfo()
    rc = hjd.lib().hjd_layout_query(ctypes.byref(q), S, overlap, fmt, ctypes.byref(o))
    assert rc == 0, (rc, w, h, S)
    return o


def ustream_word_index(i, k, wps):
    """word k of subsequence i in the transposed stream (hjd_entropy.hip::row_base, hjd_unstuff.hip)"""
    return ((i >> 6) * (wps + 2) + k) * 64 + (i & 63)


SHAPES = [(1, 1, 1, 1), (17, 9, 2, 2), (4099, 9, 2, 2), (8192, 16, 2, 2), (7, 3001, 2, 1), (250, 187, 1, 2), (512, 512, 4, 1),
          (3840, 2160, 2, 2), (1920, 1080, 2, 2), (65535, 8, 1, 1)]


@pytest.mark.parametrize("S", [64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384])
def test_no_kernel_can_leave_its_slab(hjd, header, S):
    _, jd = header
    for (w, h, hf, vf), bpp, overlap, rst, fmt in itertools.product(SHAPES, (0.05, 1.35, 6.0), (0, -1, 3), (0, 5), (0, 1)):
        mcus = -(-w // (8 * hf)) * -(-h // (8 * vf))
        scan_len = max(1, int(w * h * bpp / 8)) + (2 * (mcus // rst) if rst else 0)
        o = query(hjd, jd, w, h, hf, vf, scan_len, S, overlap, rst, fmt)
        wps = S // 32
        T_, WW = o.sync_wg, o.write_wg
        nsub = o.nsub_cap                      # the kernels use ceil(ustream_len*8/S) <= this (unstuffing only removes bytes)
        assert nsub == -(-scan_len * 8 // S)
        # --- raw: k_unstuff_* read 16 bytes per lane of every chunk, lane 255 one byte more (next = base[16])
        assert o.nchunks == -(-scan_len // o.unstuff_chunk)
        assert o.nchunks * o.unstuff_chunk + 16 <= o.raw_bytes
        # --- ustream: k_unstuff_write stores word (i, k), k < wps, and repeats words 0,1 of row i as wps, wps+1 of row i-1;
        #     k_sync / k_write stage words [ph*WPP, ph*WPP + WPP + 2) of rows i < nsub: the largest is word wps+1 of row nsub-1
        top = ustream_word_index(nsub - 1, wps + 1, wps)
        assert (top + 1) * 4 <= o.ustream_bytes, ("ustream", w, h, S)
        assert o.phase_words <= wps and wps % o.phase_words == 0
        # --- per-subsequence arrays: indices sub0 + i, i < nsub; k_write and k_repair also read entry g - 1 >= sub0
        assert nsub <= o.substate_entries
        # --- workgroup counts cover every subsequence under each kernel's tiling
        K = o.sync_overlap
        Q = T_ - K
        seq = o.nsyncwg == 0                             # k_sync_seq: a lane decodes the whole scan (thumbnails without restart markers)
        assert K < T_ and (o.nsyncwg * Q >= nsub or (seq and scan_len * 8 <= 65536 and not rst))   # k_sync: workgroup w owns [w*Q, (w+1)*Q)
        assert o.nrepwg * T_ >= nsub                     # k_repair
        nph = wps // o.phase_words
        assert o.nwritewg * WW >= nsub * nph             # k_write: one lane per staging window
        assert o.nwritewg <= o.wgsum_entries             # k_repair_totals writes wgsum[wg * (T * nph / WW) + h] only below nwritewg (guarded)
        assert WW % nph == 0 and T_ % (WW // nph) == 0   # a final-pass workgroup covers whole subsequences, a repair workgroup whole final-pass workgroups
        # --- LDS of the entropy kernels: tables, state arrays, then one window of phase_words + 2 words per lane
        assert o.lds_state_word_sync >= o.lds_state_word and o.lds_bytes_sync >= (o.lds_state_word_sync + 5 * T_ + T_ // 4 + 32 + (T_ // 64) * (o.phase_words + 2) * 64) * 4 and (T_ // 64) * (o.phase_words + 2) * 64 >= 2 * T_  # the work list lives in the staging windows
        assert o.lds_bytes_write >= (o.lds_state_word + 64 + 8 * WW + (wps + 2) * (WW // nph + 1)) * 4   # k_write: scratch + the lanes' record bursts (HJD_REC_RUN words each) + the workgroup's rows, all of their words
        assert o.lds_bytes_sync <= 160 * 1024 and o.lds_bytes_write <= 160 * 1024
        # --- checkpoints: canonical entry (ph, i) of the (only) image at ph * nsub_cap + i; k_sync workgroup w's scratch
        #     entries at nsub_cap * nph + (w * nph + ph) * K + slot, slot < K
        assert (nph - 1) * nsub + nsub - 1 < nsub * nph <= o.ckpt_entries
        small = K < 64 and nsub <= 64 - K    # one small synchronisation workgroup (k_sync_small): owns from subsequence 0, no warm-up slots, no scratch
        if small:
            assert o.nsyncwg == (0 if seq else 1)
        else:
            assert nsub * nph + ((o.nsyncwg - 1) * nph + nph - 1) * K + max(K, 1) - 1 < o.ckpt_entries
        # --- blocks: k_write stores blk[b] for b <= nblocks (closing entry); the IDCT kernel loads blk[b0 + tid], tid <= KB,
        #     and dca[b0 + tid], tid < KB, for the last group of the last MCU row: b0 + KB == nblocks
        assert o.nblocks == mcus * o.nb and o.nb == hf * vf + 2
        assert o.nblocks + 1 <= o.blk_entries and o.nblocks <= o.dca_entries
        # --- records: every store and every 16-byte load is guarded by rec_cap; the slab holds one burst more
        assert o.rec_cap % o.rec_run == 0 and o.rec_cap + o.rec_run <= o.rec_entries
        assert o.rec_cap >= min(o.nblocks * 63, scan_len * 8 // 2) + o.rec_run * nsub * nph   # a record costs at least two scan bits; every window's run is padded
        # --- DC chunks cover every MCU; dcsum holds three sums per chunk
        assert o.ndcchunk * o.dc_chunk >= mcus and 3 * o.ndcchunk <= o.dcsum_entries
        # --- IDCT groups: groups_per_row * mcuy workgroups of <= group_mcus MCUs
        assert o.groups_per_row * o.group_mcus >= o.mcux and o.ngroups == o.groups_per_row * o.mcuy
        # --- output: the last byte written is pixel (W-1, H-1) (RGB) or the end of the top row incl. padding (BMP)
        if fmt == 0:
            assert o.out_stride == 3 * w and o.out_bytes >= 3 * w * h
        else:
            assert o.out_stride == (3 * w + 3) // 4 * 4 and o.out_bytes >= 54 + o.out_stride * h
        # --- restart markers: one entry per marker a well-formed scan holds, and the kernels clamp to mark_cap
        if rst:
            assert o.mark_cap >= -(-mcus // rst) - 1
            # --- segment-parallel emulation (hjd_seq.hip): k_seg_list writes pos[at] only for at < seg_cap and gives the image up
            #     beyond; nodes j * nb + m, j <= candidates <= seg_cap; per node 2 x 16 + 2 x 4 bytes, per candidate 4, per image
            #     with DRI (one here) a count, a chain length and two plan entries, one more plan entry each, two counters
            n_rst = -(-mcus // rst) - 1
            assert o.seg_cap >= n_rst + 1 and o.seg_cap + 1 < (1 << 25)           # the successor field holds candidate + 1 in 25 bits
            assert o.seg_nodes == (o.seg_cap + 1) * o.nb
            assert o.seg_slab_bytes >= 40 * o.seg_nodes + 4 * (o.seg_cap + 2 + 2 * 2 + 2)
            tile = 512 if o.nb <= 6 else 256                                       # k_seg_chain: candidates per LDS tile
            assert tile * o.nb <= 3072 and tile <= 512                            # s_lnk[9][3072], s_d[512]
        else:
            assert o.seg_cap == 0 and o.seg_nodes == 0 and o.seg_slab_bytes == 0


def test_xcd_aware_chunk_numbering_is_a_bijection():
    """k_unstuff_write numbers its chunks so that XCD x (workgroup b runs on XCD b mod 8) takes a CONTIGUOUS run of them:
    chunk = x * (n // 8) + min(x, n % 8) + b // 8 (csrc/hjd_unstuff.hip).  Every chunk must be taken exactly once for every
    grid size — a chunk left out would leave a hole in the unstuffed stream, one taken twice is a race — and the runs must be
    contiguous per XCD, which is the point of the numbering.  (The kernel's own arithmetic is exercised by every GPU test; this
    restates it so that a change there is made on purpose.)"""
    src = open(os.path.join(ROOT, "hls-jpeg-decoder_amd", "csrc", "hjd_unstuff.hip")).read()
    assert "c = x * q + min(x, r) + j + P.off_chunk;" in src and "x = blockIdx.x & 7u, j = blockIdx.x >> 3, q = n >> 3, r = n & 7u" in src
    for n in list(range(1, 300)) + [1023, 1024, 1025, 5199, 5200, 5207, 83200, 83201]:
        q, r = n >> 3, n & 7
        taken = [None] * n
        for b in range(n):
            x, j = b & 7, b >> 3
            c = x * q + min(x, r) + j
            assert 0 <= c < n and taken[c] is None, (n, b, c)
            taken[c] = x
        # contiguous per XCD, XCDs in ascending order
        assert taken == sorted(taken), n
