"""ctypes view of the C ABI in include/hjd.h (libhjd.so: HIP kernels for gfx950 + host runtime).

The package directory name carries a hyphen (it mirrors the reference repo's name), so import it
with importlib:  hjd = importlib.import_module("hls-jpeg-decoder_amd").

Everything here is plumbing; the decode runs in the HIP kernels.  If libhjd.so is missing or no
HIP device is usable the calls fail loudly — there is no CPU fallback.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HJD_LIB") or os.path.join(_HERE, "lib", "libhjd.so")  # HJD_LIB: an experiment build of the same library

HJD_OK, HJD_ERR_NOT_JPEG, HJD_ERR_UNSUPPORTED, HJD_ERR_CORRUPT, HJD_ERR_DIRTY_RESTART, HJD_ERR_NO_DEVICE, \
    HJD_ERR_ARG, HJD_ERR_IO, HJD_ERR_INTERNAL = range(9)
OUT_RGB, OUT_BMP = 0, 1
MAX_STAGES = 16

# every symbol include/hjd.h declares
EXPORTS = [
    "hjd_status_string", "hjd_last_status", "hjd_convert_jpg_file", "hjd_decode_jpg_file_data", "hjd_free",
    "hjd_parse_header", "hjd_parse_header_ex", "hjd_jpeg_decode_hw", "hjd_decoder_create", "hjd_decoder_destroy", "hjd_device_count",
    "hjd_batch_prepare", "hjd_batch_prepare_parsed", "hjd_batch_destroy", "hjd_batch_run", "hjd_batch_finish", "hjd_batch_image_info",
    "hjd_batch_out_device_ptr", "hjd_batch_out_bytes", "hjd_batch_download", "hjd_batch_totals",
    "hjd_batch_run_timed", "hjd_batch_sync_stats", "hjd_batch_resync_count", "hjd_batch_path_counts", "hjd_batch_sequential_count", "hjd_batch_segment_count", "hjd_batch_enable_timing", "hjd_batch_collect_timing", "hjd_decode_batch_host", "hjd_bmp_bytes", "hjd_rgb_bytes", "hjd_batch_download_coefs", "hjd_test_color_table", "hjd_batch_create", "hjd_batch_reset", "hjd_host_alloc", "hjd_host_free", "hjd_layout_query", "hjd_debug_sync_profile",
]


class HuffSpec(ctypes.Structure):
    _fields_ = [("defined", ctypes.c_uint8), ("bits", ctypes.c_uint8 * 17), ("vals", ctypes.c_uint8 * 256)]


class JpegData(ctypes.Structure):
    _fields_ = [("qtable", (ctypes.c_uint8 * 64) * 4), ("dc_table_index", ctypes.c_uint8 * 4),
                ("ac_table_index", ctypes.c_uint8 * 4), ("htdc", HuffSpec * 4), ("htac", HuffSpec * 4),
                ("stream", ctypes.c_void_p), ("stream_len", ctypes.c_size_t), ("restart_interval", ctypes.c_int),
                ("restart_mcus", ctypes.c_uint), ("width", ctypes.c_uint), ("height", ctypes.c_uint),
                ("hfactor", ctypes.c_uint8), ("vfactor", ctypes.c_uint8), ("header_bytes", ctypes.c_size_t)]


class LayoutInfo(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in (
        "subseq_bits", "sync_overlap", "phase_words", "sync_wg", "write_wg", "group_mcus", "unstuff_chunk", "dc_chunk", "rec_run",
        "nblocks", "mcux", "mcuy", "nb", "nchunks", "nsub_cap", "nsyncwg", "nrepwg", "nwritewg", "ndcchunk", "ngroups",
        "groups_per_row", "mark_cap", "rec_cap", "out_stride", "lds_bytes_sync", "lds_bytes_write", "lds_state_word", "lds_state_word_sync")] + \
        [(n, ctypes.c_uint64) for n in ("raw_bytes", "ustream_bytes", "blk_entries", "dca_entries", "rec_entries", "out_bytes",
                                        "substate_entries", "wgsum_entries", "dcsum_entries", "ckpt_entries",
                                        "seg_cap", "seg_nodes", "seg_slab_bytes")]


class Options(ctypes.Structure):
    _fields_ = [("subseq_bits", ctypes.c_int), ("sync_launches", ctypes.c_int), ("streams", ctypes.c_int),
                ("sync_overlap", ctypes.c_int), ("dirty_mode", ctypes.c_int), ("conformant", ctypes.c_int), ("debug_flags", ctypes.c_int), ("reserved", ctypes.c_int)]


# hjd_options.debug_flags (include/hjd.h: HJD_DBG_*)
DBG_NO_SMALL_SYNC, DBG_NO_SEQ_SYNC, DBG_NO_SEG_EMULATION = 1, 2, 4


def dbg_siblings(n):
    return ((n + 1) & 15) << 8


def dbg_seg_cut(n):
    return (n & 0x7FFF) << 16


_lib = None


def lib():
    """Load libhjd.so (build it with `make lib` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        # build on demand (hipcc cross-compiles for gfx950 without a GPU); never a substitute implementation
        import subprocess
        root = os.path.dirname(_HERE)
        r = subprocess.run(["make", "-s", "lib"], cwd=root, capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(LIB_PATH):
            raise RuntimeError("libhjd.so not built (%s) and `make lib` failed: %s\nthere is no CPU fallback"
                               % (LIB_PATH, r.stderr[-2000:]))
    L = ctypes.CDLL(LIB_PATH)
    vp, ip, sz = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_size_t
    L.hjd_status_string.restype = ctypes.c_char_p
    L.hjd_status_string.argtypes = [ctypes.c_int]
    L.hjd_last_status.restype = ctypes.c_int
    L.hjd_convert_jpg_file.restype = ctypes.c_int
    L.hjd_convert_jpg_file.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    L.hjd_decode_jpg_file_data.restype = ctypes.c_int
    L.hjd_decode_jpg_file_data.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(vp),
                                           ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)]
    L.hjd_free.restype = None
    L.hjd_free.argtypes = [vp]
    L.hjd_parse_header.restype = ctypes.c_int
    L.hjd_parse_header.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(JpegData)]
    L.hjd_parse_header_ex.restype = ctypes.c_int
    L.hjd_parse_header_ex.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(JpegData), ctypes.c_int]
    L.hjd_jpeg_decode_hw.restype = ctypes.c_int
    L.hjd_jpeg_decode_hw.argtypes = [ctypes.POINTER(JpegData), ctypes.c_uint, ctypes.c_uint, ctypes.c_ubyte,
                                     ctypes.c_ubyte, vp]
    L.hjd_decoder_create.restype = vp
    L.hjd_decoder_create.argtypes = [ctypes.c_int, ctypes.POINTER(Options)]
    L.hjd_decoder_destroy.restype = None
    L.hjd_decoder_destroy.argtypes = [vp]
    L.hjd_device_count.restype = ctypes.c_int
    L.hjd_batch_prepare.restype = vp
    L.hjd_batch_prepare.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(sz),
                                    ctypes.c_int, ip]
    L.hjd_batch_prepare_parsed.restype = vp
    L.hjd_batch_prepare_parsed.argtypes = [vp, ctypes.c_int, ctypes.POINTER(JpegData), ctypes.c_int, ip]
    L.hjd_batch_destroy.restype = None
    L.hjd_batch_destroy.argtypes = [vp]
    L.hjd_batch_run.restype = ctypes.c_int
    L.hjd_batch_run.argtypes = [vp, vp]
    L.hjd_batch_finish.restype = ctypes.c_int
    L.hjd_batch_finish.argtypes = [vp, vp, ip]
    L.hjd_batch_image_info.restype = ctypes.c_int
    L.hjd_batch_image_info.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint),
                                       ctypes.POINTER(sz), ctypes.POINTER(sz)]
    L.hjd_batch_out_device_ptr.restype = vp
    L.hjd_batch_out_device_ptr.argtypes = [vp]
    L.hjd_batch_out_bytes.restype = sz
    L.hjd_batch_out_bytes.argtypes = [vp]
    L.hjd_batch_download.restype = ctypes.c_int
    L.hjd_batch_download.argtypes = [vp, ctypes.c_int, vp, sz]
    L.hjd_batch_download_coefs.restype = ctypes.c_int
    L.hjd_batch_download_coefs.argtypes = [vp, ctypes.c_int, vp, sz]
    L.hjd_batch_totals.restype = ctypes.c_int
    L.hjd_batch_totals.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                                   ctypes.POINTER(ctypes.c_uint64)]
    L.hjd_batch_run_timed.restype = ctypes.c_int
    L.hjd_batch_run_timed.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_char_p),
                                      ctypes.c_int]
    L.hjd_batch_enable_timing.restype = ctypes.c_int
    L.hjd_batch_enable_timing.argtypes = [vp, ctypes.c_int]
    L.hjd_batch_collect_timing.restype = ctypes.c_int
    L.hjd_batch_collect_timing.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_char_p),
                                           ctypes.c_int, ip]
    L.hjd_batch_sync_stats.restype = ctypes.c_int
    L.hjd_batch_sync_stats.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]
    L.hjd_batch_path_counts.restype = ctypes.c_int
    L.hjd_batch_path_counts.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32)]
    L.hjd_batch_resync_count.restype = ctypes.c_int
    L.hjd_batch_resync_count.argtypes = [vp]
    L.hjd_batch_sequential_count.restype = ctypes.c_int
    L.hjd_batch_sequential_count.argtypes = [vp]
    L.hjd_batch_segment_count.restype = ctypes.c_int
    L.hjd_batch_segment_count.argtypes = [vp]
    L.hjd_decode_batch_host.restype = ctypes.c_int
    L.hjd_decode_batch_host.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(sz), ctypes.c_int,
                                        ctypes.POINTER(vp), ctypes.POINTER(sz), ip]
    L.hjd_batch_create.restype = vp
    L.hjd_batch_create.argtypes = [vp]
    L.hjd_batch_reset.restype = ctypes.c_int
    L.hjd_batch_reset.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(sz), ctypes.c_int, ip, vp]
    L.hjd_layout_query.restype = ctypes.c_int
    L.hjd_layout_query.argtypes = [ctypes.POINTER(JpegData), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(LayoutInfo)]
    L.hjd_host_alloc.restype = vp
    L.hjd_host_alloc.argtypes = [sz]
    L.hjd_host_free.restype = None
    L.hjd_host_free.argtypes = [vp]
    L.hjd_test_color_table.restype = ctypes.c_int
    L.hjd_test_color_table.argtypes = [vp, vp, sz]
    L.hjd_bmp_bytes.restype = sz
    L.hjd_bmp_bytes.argtypes = [ctypes.c_uint, ctypes.c_uint]
    L.hjd_rgb_bytes.restype = sz
    L.hjd_rgb_bytes.argtypes = [ctypes.c_uint, ctypes.c_uint]
    _lib = L
    return L


def status_string(s):
    return lib().hjd_status_string(int(s)).decode()


def parse_header(jpg):
    """hjd_parse_header -> (status, JpegData).  JpegData.stream points into `jpg` (keep it alive)."""
    jd = JpegData()
    return lib().hjd_parse_header(jpg, len(jpg), ctypes.byref(jd)), jd


class Decoder:
    """hjd_decoder: one per (process, device)."""

    def __init__(self, device=0, subseq_bits=0, sync_launches=0, streams=0, sync_overlap=0, dirty_mode=0, conformant=0, debug_flags=0):
        opt = Options(subseq_bits, sync_launches, streams, sync_overlap, dirty_mode, conformant, debug_flags, 0)
        self._h = lib().hjd_decoder_create(device, ctypes.byref(opt))
        if not self._h:
            raise RuntimeError("hjd_decoder_create failed: %s" % status_string(lib().hjd_last_status()))

    def close(self):
        if getattr(self, "_h", None):
            lib().hjd_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    def color_table(self):
        """Test hook: (256, 256, 256, 3) uint8 array, [Y, Cb, Cr] -> R, G, B as the decode kernel computes them."""
        buf = np.empty(3 << 24, np.uint8)
        rc = lib().hjd_test_color_table(self._h, buf.ctypes.data, buf.size)
        if rc:
            raise RuntimeError("hjd_test_color_table: %s" % status_string(rc))
        return buf.reshape(256, 256, 256, 3)

    def prepare(self, jpgs, out_format=OUT_RGB):
        return Batch(self, jpgs, out_format)

    def prepare_parsed(self, jdatas, keep_alive=(), out_format=OUT_RGB):
        """hjd_batch_prepare_parsed: JpegData structs the caller filled (their stream pointers must stay valid: keep_alive)."""
        return Batch(self, keep_alive, out_format, parsed=list(jdatas))

    def decode_host(self, jpgs, out_format=OUT_RGB, pinned=True, outputs=None):
        """hjd_decode_batch_host: host JPEG bytes -> host RGB/BMP bytes through the three-stream pipeline.
        Returns (statuses, outputs): numpy arrays in pinned memory (hjd_host_alloc; freed with the arrays) or in
        ordinary memory (pinned=False: staged).  `outputs` re-uses the arrays of an earlier call."""
        n = len(jpgs)
        L = lib()
        if outputs is None:
            outputs = []
            for j in jpgs:
                st, jd = parse_header(j)
                nb = 0 if st else (L.hjd_rgb_bytes(jd.width, jd.height) if out_format == OUT_RGB else L.hjd_bmp_bytes(jd.width, jd.height))
                outputs.append(PinnedArray(nb) if pinned else np.empty(max(nb, 1), np.uint8))
        arr = (ctypes.c_char_p * n)(*jpgs)
        lens = (ctypes.c_size_t * n)(*[len(j) for j in jpgs])
        outs = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outputs])
        caps = (ctypes.c_size_t * n)(*[o.size for o in outputs])
        st = (ctypes.c_int * n)()
        rc = L.hjd_decode_batch_host(self._h, n, arr, lens, out_format, outs, caps, st)
        if rc:
            raise RuntimeError("hjd_decode_batch_host: %s" % status_string(rc))
        return list(st), outputs

    def decode_host_pipelined_bench(self, jpgs, w, h, reps=3):
        """End-to-end figure for bench.py: host bytes in -> pinned host RGB out, whole batch, best of `reps` after
        one warm-up call (which grows the ring of re-usable batches and the pinned output buffers)."""
        import time
        st, outs = self.decode_host(jpgs, OUT_RGB, pinned=True)
        assert all(s == HJD_OK for s in st), st
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            st, outs = self.decode_host(jpgs, OUT_RGB, outputs=outs)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        assert all(s == HJD_OK for s in st), st
        return {"value": round(len(jpgs) * w * h / best / 1e6, 1), "unit": "Mpixels/s", "ms": round(best * 1e3, 3),
                "includes": "header walk + staging copy + H2D + all kernels + D2H into pinned host memory, %d images, "
                            "three-stream pipeline (hjd_decode_batch_host)" % len(jpgs),
                "excludes": "allocation of the re-usable workspace (first call)"}

    def decode(self, jpgs, out_format=OUT_RGB):
        """Decode a list of JPEG byte strings.  Returns (statuses, outputs): outputs[i] is a numpy array
        (h, w, 3) for RGB, a bytes-like uint8 array for BMP, or None when statuses[i] != 0."""
        b = self.prepare(jpgs, out_format)
        try:
            b.run()
            st = b.finish()
            outs = [b.download(i) if st[i] == HJD_OK else None for i in range(len(jpgs))]
            return st, outs
        finally:
            b.close()


class PinnedArray(np.ndarray):
    """uint8 array in pinned host memory (hjd_host_alloc), released with the array."""

    def __new__(cls, nbytes):
        import weakref
        ptr = lib().hjd_host_alloc(max(int(nbytes), 1))
        if not ptr:
            raise MemoryError("hjd_host_alloc(%d) failed" % nbytes)
        buf = (ctypes.c_uint8 * max(int(nbytes), 1)).from_address(ptr)
        a = np.frombuffer(buf, np.uint8, count=int(nbytes)).view(cls)
        a._hjd_fin = weakref.finalize(buf, lib().hjd_host_free, ptr)
        a._hjd_buf = buf
        return a


class Batch:
    def __init__(self, dec, jpgs, out_format, parsed=None):
        self._dec = dec
        self._jpgs = list(jpgs)  # keep alive while preparing
        self.out_format = out_format
        if parsed is not None:
            self.n = len(parsed)
            arr = (JpegData * self.n)(*parsed)
            st = (ctypes.c_int * self.n)()
            self._h = lib().hjd_batch_prepare_parsed(dec._h, self.n, arr, out_format, st)
        else:
            self.n = len(self._jpgs)
            arr = (ctypes.c_char_p * self.n)(*self._jpgs)
            lens = (ctypes.c_size_t * self.n)(*[len(j) for j in self._jpgs])
            st = (ctypes.c_int * self.n)()
            self._h = lib().hjd_batch_prepare(dec._h, self.n, arr, lens, out_format, st)
        if not self._h:
            raise RuntimeError("hjd_batch_prepare failed: %s" % status_string(lib().hjd_last_status()))
        self.parse_status = list(st)

    def close(self):
        if getattr(self, "_h", None):
            lib().hjd_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    def run(self, stream=None):
        rc = lib().hjd_batch_run(self._h, stream)
        if rc:
            raise RuntimeError("hjd_batch_run: %s" % status_string(rc))

    def finish(self, stream=None):
        st = (ctypes.c_int * self.n)()
        rc = lib().hjd_batch_finish(self._h, stream, st)
        if rc:
            raise RuntimeError("hjd_batch_finish: %s" % status_string(rc))
        return list(st)

    def run_timed(self, stream=None):
        ms = (ctypes.c_float * MAX_STAGES)()
        names = (ctypes.c_char_p * MAX_STAGES)()
        n = lib().hjd_batch_run_timed(self._h, stream, ms, names, MAX_STAGES)
        if n < 0:
            raise RuntimeError("hjd_batch_run_timed failed")
        return [(names[i].decode(), float(ms[i])) for i in range(n)]

    def enable_timing(self, max_runs):
        rc = lib().hjd_batch_enable_timing(self._h, max_runs)
        if rc:
            raise RuntimeError("hjd_batch_enable_timing: %s" % status_string(rc))

    def collect_timing(self):
        """[(stage, average ms)] over the runs recorded since enable_timing / the last collect."""
        ms = (ctypes.c_float * MAX_STAGES)()
        names = (ctypes.c_char_p * MAX_STAGES)()
        runs = ctypes.c_int()
        n = lib().hjd_batch_collect_timing(self._h, ms, names, MAX_STAGES, ctypes.byref(runs))
        if n < 0:
            raise RuntimeError("hjd_batch_collect_timing failed")
        return [(names[i].decode(), float(ms[i])) for i in range(n)], runs.value

    def sync_stats(self, i):
        a = (ctypes.c_uint32 * 4)()
        lib().hjd_batch_sync_stats(self._h, i, a)
        return dict(iters=a[0], decodes=a[1], iters_max=a[2], nsub=a[3])

    def path_counts(self):
        a = (ctypes.c_uint32 * 4)()
        lib().hjd_batch_path_counts(self._h, a)
        return dict(sync_seq=a[0], sync_small=a[1], sync_large=a[2], write_small=a[3])

    def resync_count(self):
        return lib().hjd_batch_resync_count(self._h)

    def sequential_count(self):
        return lib().hjd_batch_sequential_count(self._h)

    def segment_count(self):
        return lib().hjd_batch_segment_count(self._h)

    def info(self, i):
        w, h = ctypes.c_uint(), ctypes.c_uint()
        off, nb = ctypes.c_size_t(), ctypes.c_size_t()
        rc = lib().hjd_batch_image_info(self._h, i, ctypes.byref(w), ctypes.byref(h), ctypes.byref(off), ctypes.byref(nb))
        return rc, w.value, h.value, off.value, nb.value

    def totals(self):
        a, b, c = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        lib().hjd_batch_totals(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return a.value, b.value, c.value

    def out_device_ptr(self):
        return lib().hjd_batch_out_device_ptr(self._h), lib().hjd_batch_out_bytes(self._h)

    def download(self, i):
        rc, w, h, off, nb = self.info(i)
        if rc:
            raise RuntimeError("image %d: %s" % (i, status_string(rc)))
        buf = np.empty(nb, np.uint8)
        rc = lib().hjd_batch_download(self._h, i, buf.ctypes.data, nb)
        if rc:
            raise RuntimeError("hjd_batch_download: %s" % status_string(rc))
        return buf.reshape(h, w, 3) if self.out_format == OUT_RGB else buf

    def download_coefs(self, i, nblocks):
        buf = np.empty((nblocks, 64), np.int16)
        rc = lib().hjd_batch_download_coefs(self._h, i, buf.ctypes.data, buf.size)
        if rc:
            raise RuntimeError("hjd_batch_download_coefs: %s" % status_string(rc))
        return buf


def convert_jpg_file(jpg_path, bmp_path):
    """ConvertJpgFile (openjpg.h:23): 1 on success, 0 on failure."""
    return lib().hjd_convert_jpg_file(os.fsencode(jpg_path), os.fsencode(bmp_path))
