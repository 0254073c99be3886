"""Host side of BASELINE config 3: what hjd_batch_reset costs for 1024 x 1920x1080 — header walks
(hjd_parse_header per image), layout, the copy of the scan bytes into the pinned staging buffer and the enqueueing of the
uploads — on a re-used batch object (no allocation), single-threaded and with the default host threads, beside the kernels'
time for the same batch."""
import ctypes, importlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch  # noqa
    import hjdtools as T
    hjd = importlib.import_module("hls-jpeg-decoder_amd")
    N = 1024
    src = [T.synth_jpeg(1920, 1080, 500 + k, 85, 2, 2) for k in range(8)]
    jpgs = [src[k % 8] for k in range(N)]
    L = hjd.lib()
    d = hjd.Decoder(0)
    bh = L.hjd_batch_create(d._h)
    arr = (ctypes.c_char_p * N)(*jpgs); lens = (ctypes.c_size_t * N)(*[len(x) for x in jpgs]); st = (ctypes.c_int * N)()
    ts = []
    for k in range(6):
        t0 = time.perf_counter()
        rc = L.hjd_batch_reset(bh, N, arr, lens, 0, st, None)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        assert rc == 0 and all(s == 0 for s in st)
        ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    L.hjd_batch_run(bh, None); torch.cuda.synchronize()
    t0 = time.perf_counter(); L.hjd_batch_run(bh, None); torch.cuda.synchronize(); tk = (time.perf_counter() - t0) * 1e3
    warm = ts[1:]
    print("HJD_HOST_THREADS=%s: hjd_batch_reset of %d x 1080p (%.0f MB of JPEG): host %.1f ms (best of %d; first, allocating call %.1f), "
          "until the uploads have landed %.1f ms = %.0f images/s; kernels for the same batch %.1f ms"
          % (os.environ.get("HJD_HOST_THREADS", "default"), N, sum(len(x) for x in jpgs) / 1e6, min(w[0] for w in warm), len(warm), ts[0][0],
             min(w[1] for w in warm), N / min(w[1] for w in warm) * 1e3, tk))
    L.hjd_batch_destroy(bh); d.close()
else:
    for th in ("1", None):
        env = dict(os.environ)
        if th: env["HJD_HOST_THREADS"] = th
        else: env.pop("HJD_HOST_THREADS", None)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
