// hjd_api.cpp — device runtime behind the C ABI (include/hjd.h): batch layout in HBM, table
// construction, kernel sequencing on a HIP stream, status collection.  The decode itself is the
// kernels in csrc/hjd_unstuff.hip, hjd_entropy.hip and hjd_pixels.hip; there is no host fallback.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <system_error>
#include <vector>

#include "../../include/hjd.h"
#include "../csrc/hjd_kernels.h"
#include "hjd_internal.h"

#define HIPCHK(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "hjd: %s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return HJD_ERR_INTERNAL;                                                              \
        }                                                                                         \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Host side of a large batch: the header walks and the copy of the scan bytes into the pinned staging buffer
// are independent per image.  One thread does 1024 x 1080p in ~35 ms (358 MB of memcpy) — several times the 9 ms of kernels it
// feeds — so both loops are cut over a few host threads when there is enough to do (HJD_HOST_THREADS, default up to 8).
static int host_threads()
{
    static int n = [] {
        int v = (int)std::thread::hardware_concurrency();
        if (v < 1) v = 1;
        if (v > 8) v = 8;
        if (const char *e = getenv("HJD_HOST_THREADS")) v = atoi(e) < 1 ? 1 : (atoi(e) > 64 ? 64 : atoi(e));
        return v;
    }();
    return n;
}

// Returns false when f threw (the callers turn that into HJD_ERR_INTERNAL: no exception leaves an extern "C" entry point).
template <class F>
static bool parallel_for(int n, size_t work_bytes, F f)
{
    int nt = host_threads();
    if (work_bytes < ((size_t)4 << 20) || n < 2 * nt) nt = 1;
    if (nt <= 1) {
        try {
            for (int i = 0; i < n; i++) f(i);
        } catch (...) {
            return false;
        }
        return true;
    }
    // The calling thread is worker 0; nt - 1 threads are spawned beside it.  A thread that cannot be created (EAGAIN under the
    // pool's per-card process / thread limits: 8 ranks x 8 threads) is simply not there — the others, the caller at least,
    // work the indices off — and an exception out of f is carried back to the caller after every started thread has been
    // joined: nothing may unwind past a joinable std::thread (std::terminate in a process that holds the GPU).
    std::vector<std::thread> th;
    std::atomic<int> next{0};
    std::atomic<bool> err{false};
    auto worker = [&] {
        try {
            for (;;) {
                const int i0 = next.fetch_add(16);
                if (i0 >= n) return;
                for (int i = i0; i < n && i < i0 + 16; i++) f(i);
            }
        } catch (...) {
            err.store(true);
            next.store(n); // the other workers stop at their next fetch
        }
    };
    try {
        th.reserve((size_t)nt - 1);
        for (int t = 1; t < nt; t++) th.emplace_back(worker);
    } catch (const std::system_error &) {
        // fewer helpers than asked for
    } catch (const std::bad_alloc &) {
    }
    worker();
    for (auto &t : th) t.join();
    return !err.load();
}

static const uint8_t kZzToSlot[64] = HJD_ZZ_TO_SLOT;

// bytes of the transposed stream of one image: whole groups of 64 subsequences, each subsequence a row of
// wps + 2 words (its own words and the first two of the next), plus one group of slack
static inline size_t hjd_ustream_bytes(uint32_t nsub_cap, uint32_t S)
{
    return ((size_t)(nsub_cap + 63) / 64 + 1) * ((S / 32) + 2) * 256;
}

// Makes the decoder's device current for the duration of a call and puts the caller's device back
// afterwards: a batch may be driven from a thread whose current device is another one.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (dev < 0) return; // layout-only use (hjd_layout_query): no device involved
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
        else prev = -1; // nothing to restore
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

struct hjd_decoder {
    int device = 0;
    uint32_t subseq_bits = 2048;
    bool subseq_auto = true;    // no explicit size: chosen per batch from the amount of scan data (batch_build)
    bool overlap_auto = true;
    int sync_launches = 1; // the synchronisation launch, plus this many - 1 in-stream repair launches (gates that return at once
                           // unless a workgroup's 16 kbit of warm-up were not enough).  Default none: k_write checks the
                           // induction step for every subsequence and hjd_batch_finish repairs from the host
    uint32_t sync_overlap = 8; // warm-up subsequences per synchronisation workgroup
    uint32_t phase_words = 16; // words of a subsequence staged in LDS at a time = the window one k_write lane decodes
    int streams = 0;           // image ranges a batch is split into (hjd_options.streams); 0 = by batch size
    // Front-end concurrency limit (HJD_SYNC_CONC): the synchronisation kernel is a latency chain that
    // fills a fraction of the chip's issue slots; batches run on several caller streams otherwise march in lock step (all in
    // their synchronisation at once, then all in their IDCT).  With a limit G the k-th run's synchronisation waits for that of
    // run k-G: the runs spread out in time, so that at any moment some are in the latency-bound stage and the others in the
    // issue-bound ones.  0 = no limit.
    int sync_conc = 2;
    std::mutex conc_mu;
    hipEvent_t conc_ev[16] = {};
    uint64_t conc_runs = 0;
    // The same for the back end and the final pass (HJD_IDCT_CONC / HJD_WRITE_CONC; 0 = no limit): two back-end kernels at a time each run
    // at half speed and leave a stretch of the step in which none runs and the latency-bound kernels have the chip to themselves
    // (tools/timeline.py on a kernel trace of the bench: no k_idct_color in flight 31 % of the time).  With a limit of one the
    // runs' back ends follow each other in the order of the runs: a software pipeline across the caller's streams.
    int idct_conc = 0, write_conc = 0;
    int no_side = 0;           // experiment knob HJD_NO_SIDE: the small images' kernels follow the large ones on the run's own stream
    void *recent_streams[8] = {}; // the streams of the last eight runs (conc_mu): how many caller streams are in flight
    // The library's own streams: three per decoder, created AND used once when the decoder is created.  The HIP runtime binds a
    // stream to one of the process's hardware queues (four by default, GPU_MAX_HW_QUEUES) at its first command — a new queue while
    // there are fewer than four, the least shared one afterwards — and two streams on one queue run one behind the other.  A decoder
    // created early in the process therefore holds queues of its own for the ranges of a split batch (ranges 1..3; range 0 runs on
    // the caller's stream) and for the small images' side streams.  Until round 5 every batch created its own: the ranges of one
    // batch shared queues with one another and with the caller's stream (rocprofv3 Queue_Id; BASELINE config 5 split four ways
    // inside the library: 119 Gpixel/s against 147 unsplit).
    hipStream_t lib_streams[3] = {};
    void *d_bind = nullptr;
    uint64_t recent_n = 0;
    hipEvent_t idct_ev[16] = {}, write_ev[16] = {};
    uint64_t idct_runs = 0, write_runs = 0;
    uint32_t write_windows = 1; // staging windows one k_write lane decodes (HJD_WRITE_WINDOWS: tuning knob, power of two)
    int stagger = 0;           // ranges of a batch run as a software pipeline (batch_enqueue): 0 off, 1 after sync, 2 after the final pass
    int dirty_mode = 0;        // 0: restart streams outside the clean-resync domain are emulated sequentially; 1: refused
    int conformant = 0;        // hjd_options.conformant: opt-in, outside the parity contract
    // hjd_options.debug_flags (test switches: they select among paths that must all give the same bytes) and the diagnostics /
    // tuning variables of the environment — all read ONCE, here, when the decoder is created: a batch never looks at the
    // environment (round 4 called getenv per image and per prepare, on caller threads)
    bool no_small_sync = false, no_seq_sync = false, no_seg_emulation = false;
    int siblings = -1;         // -1 = by launch size, 0..HJD_MAX_SIBLINGS = exactly that many
    uint32_t sib_limit = 128;  // HJD_SIB_LIMIT: workgroups (with siblings) a launch may have and still get siblings
    uint32_t seg_cut = 0;      // 0 = default (256 blocks or two restart intervals + 64), else exactly that many
    bool sync_prof = false, verbose = false;
    int max_resync = 4096;
    // host-to-host pipeline (hjd_decode_batch_host): a ring of re-usable batches and three streams, so that chunk
    // k+1's upload and chunk k-1's download run beside chunk k's kernels
    struct Slot {
        hjd_batch *b = nullptr;
        hipEvent_t up_done = nullptr, run_done = nullptr, down_done = nullptr;
        int first = 0, count = 0;
        bool busy = false;
        uint8_t *h_out = nullptr; // pinned staging for outputs whose destination is pageable memory
        size_t h_out_cap = 0;
        std::vector<uint8_t> direct; // per image of the chunk: copied straight to the caller's (pinned) buffer
    };
    Slot slot[3];
    hipStream_t s_up = nullptr, s_run = nullptr, s_down = nullptr;
    std::mutex pipe_mu;
};

struct hjd_batch {
    hjd_decoder *dec = nullptr;
    int device = 0;                  // dec->device, kept here so that a batch can outlive calls on its decoder
    int n = 0;                       // images offered
    int nimg = 0;                    // images that parsed and are on the device
    std::vector<int> parse_status;   // per offered image
    std::vector<int> dev_index;      // offered image -> device image index or -1
    std::vector<HjdImage> img;       // device images (host copy)
    std::vector<size_t> out_bytes;   // per device image
    std::vector<size_t> hdr_bytes;
    std::vector<HjdImageState> state;
    int out_format = 0;
    // sub-batches: contiguous ranges of device images, each enqueued on its own internal stream between a fork
    // and a join event on the caller's stream (one range's latency-bound synchronisation overlaps another's IDCT)
    struct Range { int i0, i1; uint32_t rst0, nrst; uint32_t chunk0, nchunk, syncwg0, nsyncwg, ssyncwg0, nssyncwg, repwg0, nrepwg, writewg0, nwritewg, swritewg0, nswritewg, seq0, nseq, side_feeds_big, dc0, ndc, group0, ngroup; size_t blk0, nblk; };
    std::vector<Range> ranges;
    std::vector<hipEvent_t> join_ev;
    std::vector<hipEvent_t> stagger_ev; // per range: recorded after its synchronisation (or final pass), the next range waits for it
    hipEvent_t fork_ev = nullptr;
    int resyncs = 0;                 // host-driven re-synchronisation rounds since prepare (0 in normal operation)
    int dirty_decodes = 0;           // images decoded by the emulation kernels (hjd_seq.hip) since prepare
    int segment_decodes = 0;         // ... of those, parallel over the resynchronisations (k_seg_*)
    // Restart streams outside the clean-resync domain: every run ends with a launch of k_seq_decode over the images that carry
    // DRI (it returns at once for those the parallel path did not flag) and a gated launch of the back end, on a side stream
    // that waits for the run and that hjd_batch_finish waits for.  Nothing is allocated and no host round trip is needed.
    std::vector<uint32_t> rst_images; // device images with DRI, ascending
    uint32_t nrst_groups = 0;         // back-end workgroups of those images
    // The small synchronisation workgroups of a range run BESIDE its large ones, on a side stream of the range (forked after the
    // unstuff stage, joined before the final pass): one kernel after the other in one stream cost a mixed batch (BASELINE
    // config 5) the whole of the second one's latency.
    struct SideStream { hipStream_t st = nullptr; hipEvent_t fork = nullptr, done = nullptr; };
    std::vector<SideStream> small_side; // per range (entry 0 also serves the single-range form)
    hipEvent_t seq_done = nullptr;    // recorded behind the emulation launches of a run (on the run's stream)
    bool seq_pending = false;
    bool emulate_in_run = false;      // a finish of these contents found restart streams to emulate: later runs enqueue the emulation themselves
    bool emul_enqueued = false;       // the last enqueue of this batch carried the emulation launches
    int callers_seen = 1;             // distinct streams among the decoder's last eight runs, as of this batch's last enqueue
    uint32_t nchunks = 0, nsub = 0, nsyncwg = 0, nsyncwg_small = 0, nrepwg = 0, nwritewg = 0, nwritewg_small = 0, ngroups = 0, ndcchunks = 0;
    uint32_t *d_dcsum = nullptr;
    uint32_t nmarks = 0;
    uint32_t *d_chunk_mark = nullptr, *d_mpos = nullptr;
    uint8_t *d_mval = nullptr;
    size_t raw_bytes = 0, ustream_bytes = 0, nblk_entries = 0, nrec = 0, out_total = 0;
    uint64_t pixels = 0, alg_bytes = 0, scan_bytes = 0;
    // device
    HjdImage *d_img = nullptr;
    HjdImageState *d_st = nullptr;
    HjdTableSet *d_tabs = nullptr;
    uint8_t *d_raw = nullptr, *d_ustream = nullptr, *d_out = nullptr;
    uint32_t *d_chunk_kept = nullptr;
    uint4 *d_substate = nullptr;
    uint32_t *d_subcnt = nullptr;
    uint32_t *d_submk = nullptr; // restart streams: per subsequence, index of the first marker past its first bit (k_sync -> k_write, repair)
    uint2 *d_ckpt = nullptr;
    uint2 *d_tkck = nullptr;        // sibling trajectories' checkpoints (small batches)
    uint32_t *d_tkflags = nullptr;
    uint32_t *d_repflags = nullptr;  // repair workgroups' stamps (HjdParams::rep_flags)
    uint32_t *d_syncflags = nullptr; // synchronisation workgroups' stamps (HjdParams::sync_flags)
    uint32_t *d_seqlist = nullptr;   // k_sync_seq: 64 device images per workgroup
    uint8_t *d_seg = nullptr;        // segment-parallel emulation: candidates, nodes, plans (one slab, carved up in batch_build)
    size_t seg_npos = 0, seg_nnodes = 0;
    std::vector<uint32_t> seqlist;
    uint32_t epoch = 0;             // run counter: what a sibling workgroup's flag must show
    unsigned long long *d_dbg = nullptr; // HJD_SYNC_PROF=1: k_sync's per-workgroup clock stamps
    uint2 *d_wgsum = nullptr;
    uint2 *d_blk = nullptr;
    int16_t *d_dca = nullptr;
    uint32_t *d_rec = nullptr;
    uint32_t *d_maps = nullptr; // workgroup -> image tables of all kernels, one allocation
    // A batch object can be re-prepared in place (hjd_batch_reset): slabs are only re-allocated when the new
    // contents need more than they hold.  h_stage = pinned host staging for everything uploaded (scan bytes,
    // image descriptors, tables, maps), so that the uploads are true asynchronous copies.
    bool dry = false;                   // hjd_layout_query: lay out only, touch no device
    std::map<const void *, size_t> cap; // capacity in bytes per slab, keyed by the address of the slab pointer
    uint8_t *h_stage = nullptr;
    size_t h_stage_cap = 0;
    HjdParams P{};
    hipEvent_t ev[HJD_MAX_STAGES + 1] = {};
    bool have_events = false;
    // optional per-run stage timing (hjd_batch_enable_timing): one event set per recorded run
    std::vector<std::vector<hipEvent_t>> tev;
    std::vector<int> tev_n;
    const char *tev_names[HJD_MAX_STAGES + 1] = {};
    int tev_cap = 0;
};

// (Huffman table construction: hjd_build_tabset, host/hjd_host.cpp — plain C++, covered by `make sanitize`)

static std::string tabset_key(const hjd_jpeg_data &j)
{
    std::string k;
    for (int c = 1; c <= 3; c++) {
        const hjd_huff_spec &d = j.htdc[j.dc_table_index[c]], &a = j.htac[j.ac_table_index[c]];
        k.append((const char *)d.bits, 17).append((const char *)d.vals, 256);
        k.append((const char *)a.bits, 17).append((const char *)a.vals, 256);
    }
    return k;
}

// ---- decoder ------------------------------------------------------------------------------------

extern "C" int hjd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" hjd_decoder *hjd_decoder_create(int device, const hjd_options *opt)
{
    int n = hjd_device_count();
    if (n <= 0 || device < 0 || device >= n) {
        hjd_set_last_status(HJD_ERR_NO_DEVICE);
        fprintf(stderr, "hjd: no usable HIP device %d (found %d); this path has no CPU fallback\n", device, n);
        return nullptr;
    }
    {
        DeviceGuard g(device); // only checks that the device can be made current
        if (!g.ok) {
            hjd_set_last_status(HJD_ERR_NO_DEVICE);
            return nullptr;
        }
    }
    hjd_decoder *d = new hjd_decoder();
    d->device = device;
    if (opt) {
        if (opt->subseq_bits) {
            if (opt->subseq_bits < 64 || opt->subseq_bits > 16384 || (opt->subseq_bits & (opt->subseq_bits - 1))) {
                delete d;
                hjd_set_last_status(HJD_ERR_ARG);
                return nullptr;
            }
            d->subseq_bits = (uint32_t)opt->subseq_bits;
            d->subseq_auto = false;
        }
        if (opt->sync_launches > 0) d->sync_launches = opt->sync_launches;
        if (opt->streams > 0) d->streams = opt->streams > 16 ? 16 : opt->streams;
        d->dirty_mode = opt->dirty_mode ? 1 : 0;
        d->conformant = opt->conformant ? 1 : 0;
        const unsigned f = (unsigned)opt->debug_flags;
        d->no_small_sync = (f & HJD_DBG_NO_SMALL_SYNC) != 0;
        d->no_seq_sync = (f & HJD_DBG_NO_SEQ_SYNC) != 0;
        d->no_seg_emulation = (f & HJD_DBG_NO_SEG_EMULATION) != 0;
        if ((f >> 8) & 15u) d->siblings = (int)((f >> 8) & 15u) - 1 > (int)HJD_MAX_SIBLINGS ? (int)HJD_MAX_SIBLINGS : (int)((f >> 8) & 15u) - 1;
        d->seg_cut = (f >> 16) & 0x7FFFu;
    }
    if (const char *e = getenv("HJD_SIB_LIMIT")) d->sib_limit = (uint32_t)(atoi(e) < 0 ? 0 : atoi(e)); // tuning knob
    d->sync_prof = getenv("HJD_SYNC_PROF") != nullptr;  // diagnostics
    d->verbose = getenv("HJD_VERBOSE") != nullptr;
    if (const char *e = getenv("HJD_MAX_RESYNC")) d->max_resync = atoi(e);
    if (const char *ew = getenv("HJD_WRITE_WINDOWS"))
        if (atoi(ew) >= 1 && atoi(ew) <= 64 && (atoi(ew) & (atoi(ew) - 1)) == 0) d->write_windows = (uint32_t)atoi(ew);
    if (const char *ec = getenv("HJD_SYNC_CONC")) d->sync_conc = atoi(ec) < 0 ? 0 : (atoi(ec) > 8 ? 8 : atoi(ec));
    if (const char *ec = getenv("HJD_IDCT_CONC")) d->idct_conc = atoi(ec) < 0 ? 0 : (atoi(ec) > 8 ? 8 : atoi(ec));
    d->no_side = getenv("HJD_NO_SIDE") != nullptr;
    if (const char *ec = getenv("HJD_WRITE_CONC")) d->write_conc = atoi(ec) < 0 ? 0 : (atoi(ec) > 8 ? 8 : atoi(ec));
    if (const char *es = getenv("HJD_STAGGER")) d->stagger = atoi(es) < 0 ? 0 : (atoi(es) > 2 ? 2 : atoi(es)); // experiment knob
    const char *e = getenv("HJD_SUBSEQ_BITS");
    if (e && atoi(e) >= 64 && atoi(e) <= 16384 && (atoi(e) & (atoi(e) - 1)) == 0) {
        d->subseq_bits = (uint32_t)atoi(e);
        d->subseq_auto = false;
    }
    {   // warm-up: ~16384 bits of look-back, which a wrong start state does not survive in practice (measured
        // survival on the q85 bench content: 30 % after 1024 bits, 8 % after 2048, about x0.27 per further 1024)
        uint32_t k = 16384 / d->subseq_bits;
        d->sync_overlap = k < 4 ? 4 : (k > 32 ? 32 : k);
        const char *ek = getenv("HJD_SYNC_OVERLAP");
        if (ek && atoi(ek) >= 0 && atoi(ek) <= 48) {
            d->sync_overlap = (uint32_t)atoi(ek);
            d->overlap_auto = false;
        }
        const char *ep = getenv("HJD_PHASE_WORDS"); // tuning knob: staging window in 32-bit words (power of two)
        if (ep && atoi(ep) >= 2 && atoi(ep) <= 64 && (atoi(ep) & (atoi(ep) - 1)) == 0) d->phase_words = (uint32_t)atoi(ep);
        if (opt && opt->sync_overlap != 0) {
            if (opt->sync_overlap < -1 || opt->sync_overlap > 48) {
                delete d;
                hjd_set_last_status(HJD_ERR_ARG);
                return nullptr;
            }
            d->sync_overlap = opt->sync_overlap < 0 ? 0u : (uint32_t)opt->sync_overlap;
            d->overlap_auto = false;
        }
    }
    {
        DeviceGuard g(device);
        bool ok = g.ok && hipMalloc(&d->d_bind, 256) == hipSuccess;
        for (auto &st : d->lib_streams) {
            ok = ok && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
            ok = ok && hipMemsetAsync(d->d_bind, 0, 256, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess; // bound to a queue
        }
        if (!ok) {
            for (auto &st : d->lib_streams)
                if (st) (void)hipStreamDestroy(st);
            (void)hipFree(d->d_bind);
            delete d;
            hjd_set_last_status(HJD_ERR_NO_DEVICE);
            return nullptr;
        }
    }
    return d;
}

extern "C" void hjd_batch_destroy(hjd_batch *b);

extern "C" void hjd_decoder_destroy(hjd_decoder *d)
{
    if (!d) return;
    {
        DeviceGuard g(d->device);
        for (auto &sl : d->slot) {
            if (sl.b) hjd_batch_destroy(sl.b);
            if (sl.up_done) (void)hipEventDestroy(sl.up_done);
            if (sl.run_done) (void)hipEventDestroy(sl.run_done);
            if (sl.down_done) (void)hipEventDestroy(sl.down_done);
            if (sl.h_out) (void)hipHostFree(sl.h_out);
        }
        for (auto &e : d->conc_ev)
            if (e) (void)hipEventDestroy(e);
        for (auto &e : d->idct_ev)
            if (e) (void)hipEventDestroy(e);
        for (auto &e : d->write_ev)
            if (e) (void)hipEventDestroy(e);
        for (auto &st : d->lib_streams)
            if (st) (void)hipStreamDestroy(st);
        (void)hipFree(d->d_bind);
        if (d->s_up) (void)hipStreamDestroy(d->s_up);
        if (d->s_run) (void)hipStreamDestroy(d->s_run);
        if (d->s_down) (void)hipStreamDestroy(d->s_down);
    }
    delete d;
}

// ---- batch --------------------------------------------------------------------------------------

// (Re)allocates *p when it holds less than `bytes`; keeps it otherwise.
// Bytes of the one slab behind the segment-parallel emulation (hjd_seq.hip), carved up in batch_build: per node two uint4 (segment
// sums, sums before it) and two words (successor, place on the chain); per candidate a position; per image with DRI a candidate
// count and a chain length; two work-item prefixes over those images; the two last-workgroup counters.
static size_t seg_slab_size(const hjd_batch *b)
{
    if (!b->seg_nnodes) return 0;
    const size_t nrst = b->rst_images.size(), words = b->seg_npos + 2 * nrst + 2 * (nrst + 1) + 2 + 2 * b->seg_nnodes + 8;
    return 2 * sizeof(uint4) * b->seg_nnodes + sizeof(uint32_t) * words;
}

template <class T>
static int slab_reserve(hjd_batch *b, T *&p, size_t bytes)
{
    if (bytes < 16) bytes = 16;
    size_t &c = b->cap[(const void *)&p];
    if (p && c >= bytes) return HJD_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    c = 0;
    const size_t want = bytes + bytes / 16; // a little head-room: chunks of a stream of batches differ slightly
    HIPCHK(hipMalloc(&p, want));
    c = want;
    return HJD_OK;
}

static void batch_free_device(hjd_batch *b)
{
    if (b->seq_pending && b->seq_done) (void)hipEventSynchronize(b->seq_done); // the emulation of the last run may still be at work on the slabs
    if (b->h_stage) (void)hipHostFree(b->h_stage);
    (void)hipFree(b->d_img);
    (void)hipFree(b->d_st);
    (void)hipFree(b->d_tabs);
    (void)hipFree(b->d_raw);
    (void)hipFree(b->d_ustream);
    (void)hipFree(b->d_out);
    (void)hipFree(b->d_chunk_kept);
    (void)hipFree(b->d_substate);
    (void)hipFree(b->d_subcnt);
    (void)hipFree(b->d_submk);
    (void)hipFree(b->d_ckpt);
    (void)hipFree(b->d_tkck);
    (void)hipFree(b->d_tkflags);
    (void)hipFree(b->d_repflags);
    (void)hipFree(b->d_syncflags);
    (void)hipFree(b->d_seqlist);
    (void)hipFree(b->d_seg);
    (void)hipFree(b->d_dbg);
    (void)hipFree(b->d_wgsum);
    (void)hipFree(b->d_rec);
    (void)hipFree(b->d_blk);
    (void)hipFree(b->d_dca);
    (void)hipFree(b->d_maps);
    (void)hipFree(b->d_dcsum);
    for (auto &ss : b->small_side) { // (ss.st belongs to the decoder)
        if (ss.fork) (void)hipEventDestroy(ss.fork);
        if (ss.done) (void)hipEventDestroy(ss.done);
    }
    if (b->seq_done) (void)hipEventDestroy(b->seq_done);
    (void)hipFree(b->d_chunk_mark);
    (void)hipFree(b->d_mpos);
    (void)hipFree(b->d_mval);
    if (b->have_events)
        for (auto &e : b->ev) (void)hipEventDestroy(e);
    for (auto &v : b->tev)
        for (auto &e : v) (void)hipEventDestroy(e);
    for (auto &e : b->join_ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : b->stagger_ev)
        if (e) (void)hipEventDestroy(e);
    if (b->fork_ev) (void)hipEventDestroy(b->fork_ev);
}

extern "C" void hjd_batch_destroy(hjd_batch *b)
{
    if (!b) return;
    DeviceGuard g(b->device);
    batch_free_device(b);
    delete b;
}

// Lays the images out, makes sure the slabs are large enough and enqueues the uploads on `up` (all of them
// asynchronous copies out of the batch's pinned staging buffer: nothing of the caller's memory is referenced
// once this returns).  Usable on a fresh batch and on one that has been used before.
static int batch_build(hjd_batch *b, const std::vector<hjd_jpeg_data> &parsed, hipStream_t up)
{
    hjd_decoder *d = b->dec;
    b->img.clear();
    b->out_bytes.clear();
    b->hdr_bytes.clear();
    b->ranges.clear();
    b->nchunks = b->nsub = b->nsyncwg = b->nsyncwg_small = b->nrepwg = b->nwritewg = b->nwritewg_small = b->ngroups = b->ndcchunks = b->nmarks = 0;
    b->pixels = b->alg_bytes = b->scan_bytes = 0;
    b->resyncs = 0;
    b->dirty_decodes = b->segment_decodes = 0;
    DeviceGuard guard(b->dry ? -1 : d->device);
    if (!guard.ok) return HJD_ERR_NO_DEVICE;
    if (b->seq_pending) { // the previous contents' sequential emulation (side stream) still owns the slabs
        HIPCHK(hipEventSynchronize(b->seq_done));
        b->seq_pending = false;
    }
    // Subsequence size: every lane of k_sync decodes one subsequence serially (k_write runs one lane per staging window whatever
    // the subsequence size), so a small batch wants short ones and a large batch long ones (fewer synchronisation
    // workgroups, tables loaded less often: 64 x 4K on four streams, 240 Gpixel/s at 2048 bits, 270 at 4096).  Unless the
    // caller fixed it: 4096 bits when that still leaves about 32768 subsequences, else 2048.
    uint32_t S = d->subseq_bits, K = d->sync_overlap;
    if (d->subseq_auto) {
        uint64_t bits = 0;
        for (int i = 0; i < b->n; i++)
            if (b->parse_status[i] == HJD_OK) bits += (uint64_t)parsed[i].stream_len * 8u;
        S = 2048; // also for one image (3840x2160: 0.49 ms at 2048 bits, 0.70 ms at 256: short subsequences mean more refinement rounds)
        while (S < 4096 && (uint64_t)S * 2u * 32768u <= bits) S *= 2;
        if (d->overlap_auto) {
            const uint32_t k = 16384 / S;
            K = k < 4 ? 4 : (k > 32 ? 32 : k);
        }
    }
    uint32_t wpp = d->phase_words; // staging window of the entropy kernels, in words; k_write runs one lane per window
    if (wpp > S / 32) wpp = S / 32;
    const uint32_t nph = (S / 32) / wpp; // windows per subsequence
    uint32_t wwn = d->write_windows;     // windows per k_write lane
    if (wwn > nph) wwn = nph;
    while (wwn > 1 && (size_t)(S / 32 + 2) * (HJD_WRITE_WG / (nph / wwn) + 1) * 4 > (88u << 10)) wwn /= 2; // k_write keeps its rows in LDS
    const uint32_t nu = nph / wwn;       // k_write lanes per subsequence
    std::map<std::string, uint32_t> tab_index;
    std::vector<HjdTableSet> tabs;
    size_t raw_cur = 0, us_cur = 0, blk_cur = 0, rec_cur = 0, out_cur = 0;
    uint32_t max_nlut = 1, max_nsec = 0, max_nsec_s = 0;
    const uint32_t Q = HJD_SYNC_WG - K; // owned subsequences per synchronisation workgroup

    for (int i = 0; i < b->n; i++) {
        b->dev_index[i] = -1;
        if (b->parse_status[i] != HJD_OK) continue;
        const hjd_jpeg_data &j = parsed[i];
        {   // the same checks for parsed headers and for caller-filled structs (hjd_jpeg_decode_hw)
            const int vst = hjd_validate_jpeg_data(&j, d->conformant ? HJD_PARSE_CONFORMANT : 0);
            if (vst != HJD_OK) {
                b->parse_status[i] = vst;
                continue;
            }
        }
        HjdImage im;
        memset(&im, 0, sizeof(im));
        im.width = j.width;
        im.height = j.height;
        im.hf = j.hfactor;
        im.vf = j.vfactor;
        im.nby = im.hf * im.vf;
        im.nb = im.nby + 2;
        im.mcux = (im.width + 8 * im.hf - 1) / (8 * im.hf); // loadjpg.cpp:1170-1176 loop trip counts
        im.mcuy = (im.height + 8 * im.vf - 1) / (8 * im.vf);
        im.nblocks = im.mcux * im.mcuy * im.nb;
        im.scan_len = (uint32_t)j.stream_len;
        const std::string key = tabset_key(j);
        auto it = tab_index.find(key);
        if (it == tab_index.end()) {
            HjdTableSet ts;
            if (!hjd_build_tabset(ts, j)) { // more long-code prefixes than the secondary tables hold
                b->parse_status[i] = HJD_ERR_UNSUPPORTED;
                continue;
            }
            max_nlut = ts.nlut > max_nlut ? ts.nlut : max_nlut;
            max_nsec = ts.nsec > max_nsec ? ts.nsec : max_nsec;
            max_nsec_s = ts.nsec_s > max_nsec_s ? ts.nsec_s : max_nsec_s;
            it = tab_index.emplace(key, (uint32_t)tabs.size()).first;
            tabs.push_back(ts);
        }
        im.tabset = it->second;
        im.chunk0 = b->nchunks;
        im.nchunks = (im.scan_len + HJD_UNSTUFF_CHUNK - 1) / HJD_UNSTUFF_CHUNK;
        b->nchunks += im.nchunks;
        im.sub0 = b->nsub;
        im.nsub_cap = (uint32_t)(((uint64_t)im.scan_len * 8 + S - 1) / S);
        b->nsub += im.nsub_cap;
        // An image whose whole scan fits one small synchronisation workgroup (with the same warm-up allowance) gets one:
        // k_sync_small, numbered on its own
        im.sync_small = (K < HJD_SYNC_WG_SMALL && im.nsub_cap <= HJD_SYNC_WG_SMALL - K && !d->no_small_sync) ? 1u : 0u;
        // ... and a thumbnail (no restart markers, a scan of at most HJD_SEQ_MAX_BITS) none at all: one lane of k_sync_seq decodes it
        if (im.sync_small && j.restart_interval <= 0 && (uint64_t)im.scan_len * 8u <= HJD_SEQ_MAX_BITS && !d->no_seq_sync) im.sync_small = 2u;
        if (im.sync_small == 2u) {
            im.syncwg0 = 0;
            im.nsyncwg = 0;
        } else if (im.sync_small) {
            im.syncwg0 = b->nsyncwg_small;
            im.nsyncwg = 1;
            b->nsyncwg_small += 1;
        } else {
            im.syncwg0 = b->nsyncwg;
            im.nsyncwg = (im.nsub_cap + Q - 1) / Q;
            b->nsyncwg += im.nsyncwg;
        }
        im.repwg0 = b->nrepwg;
        im.nrepwg = (im.nsub_cap + HJD_SYNC_WG - 1) / HJD_SYNC_WG;
        b->nrepwg += im.nrepwg;
        // one lane per span of wwn windows; the images with a small synchronisation workgroup get small final-pass workgroups too
        // (up to two of them: a 256 x 256 image in five small workgroups — five loads of the tables — took 0.55 ms per 4096 images
        // against 0.37 in two large ones)
        im.write_small = (im.sync_small && nu <= HJD_WRITE_WG_SMALL && (uint64_t)im.nsub_cap * nu <= 2u * HJD_WRITE_WG_SMALL) ? 1u : 0u;
        if (im.write_small) {
            im.writewg0 = b->nwritewg_small;
            im.nwritewg = (uint32_t)(((uint64_t)im.nsub_cap * nu + HJD_WRITE_WG_SMALL - 1) / HJD_WRITE_WG_SMALL);
            b->nwritewg_small += im.nwritewg;
        } else {
            im.writewg0 = b->nwritewg;
            im.nwritewg = (uint32_t)(((uint64_t)im.nsub_cap * nu + HJD_WRITE_WG - 1) / HJD_WRITE_WG);
            b->nwritewg += im.nwritewg;
        }
        im.dcchunk0 = b->ndcchunks;
        im.ndcchunk = (im.mcux * im.mcuy + HJD_DC_CHUNK - 1) / HJD_DC_CHUNK;
        b->ndcchunks += im.ndcchunk;
        im.group0 = b->ngroups;
        im.groups_per_row = (im.mcux + HJD_GROUP_MCUS - 1) / HJD_GROUP_MCUS;
        im.ngroups = im.groups_per_row * im.mcuy;
        b->ngroups += im.ngroups;
        im.out_format = (uint32_t)b->out_format;
        im.restart = j.restart_interval > 0 ? 1u : 0u; // the reference only tests "> 0" (loadjpg.cpp:535)
        im.conformant = d->conformant ? 1u : 0u;
        im.ri_blocks = im.restart ? j.restart_mcus * im.nb : 0u;
        im.mark_off = b->nmarks;
        if (im.restart && j.restart_mcus == 0) {
            // The interval is not known: what an unmodified reference host hands over at the HW/SW cut (openjpg.cpp:441-446 keeps
            // the DRI segment's LENGTH, never Ri), or a DRI segment that says 0.  The reference itself never knows it either — it
            // resynchronises wherever its byte cursor meets FF xx at a block start (loadjpg.cpp:535-550) — so the markers are taken
            // where the scan holds them: counted here by the unstuff stage's own pair rule (keepmask_rule, restart form), which
            // sizes the marker and candidate slabs; ri_blocks = 0 tells the final pass that no block count goes with a marker.
            uint32_t pairs = 0, before_eoi = 0xFFFFFFFFu;
            unsigned prev = 0;
            if (b->dry) pairs = (uint32_t)(j.stream_len / 2); // hjd_layout_query never reads the scan: the bound instead of the count
            for (size_t q = 0; q < j.stream_len && !b->dry; q++) {
                const unsigned bv = j.stream[q];
                const bool second = prev == 0xFFu;
                if (second && bv != 0) {
                    if (bv == 0xD9 && before_eoi == 0xFFFFFFFFu) before_eoi = pairs;
                    pairs++;
                }
                prev = second ? 0u : bv;
            }
            im.mark_cap = pairs + 8u;
            im.n_rst = before_eoi == 0xFFFFFFFFu ? pairs : before_eoi; // everything before EOI has to be an RSTn (k_unstuff_write)
        } else {
            im.mark_cap = im.restart ? (im.mcux * im.mcuy + j.restart_mcus - 1) / j.restart_mcus + 8u : 0u;
            im.n_rst = im.restart ? (im.mcux * im.mcuy + j.restart_mcus - 1) / j.restart_mcus - 1 : 0u;
        }
        b->nmarks += im.mark_cap;
        size_t obytes;
        if (b->out_format == HJD_OUT_BMP) {
            im.out_stride = (im.width * 3 + 3) & ~3u;
            obytes = hjd_bmp_bytes(im.width, im.height);
            out_cur = align_up(out_cur, 256) + 2; // pixel rows start 54 bytes in: keep them 4-byte aligned
        } else {
            im.out_stride = im.width * 3;
            obytes = hjd_rgb_bytes(im.width, im.height);
            out_cur = align_up(out_cur, 256);
        }
        im.out_off = out_cur;
        out_cur += obytes;
        im.raw_off = raw_cur;
        raw_cur += align_up((size_t)im.nchunks * HJD_UNSTUFF_CHUNK + 16, 256);
        im.ustream_off = us_cur;
        us_cur += hjd_ustream_bytes(im.nsub_cap, S); // whole 64-row groups of wps+2 words
        im.blk_off = blk_cur;
        blk_cur += (size_t)im.nblocks + 1; // + the entry that closes the last block's record range
        {   // record capacity: one per non-zero AC coefficient (<= 63 per block, and every AC symbol with value
            // bits takes at least 2 bits of scan: a 1-bit code for (run 0, size 1) plus the value bit), plus
            // the rounding of every subsequence's run to a whole burst
            const uint64_t by_blocks = (uint64_t)im.nblocks * 63, by_bits = (uint64_t)im.scan_len * 8 / 2;
            // With DRI the reference may leave the marker grid and finish its blocks on the zeros past the end of the
            // scan (sequential emulation, hjd_seq.hip): only the per-block bound holds there.
            const uint64_t bound = im.restart ? by_blocks : (by_blocks < by_bits ? by_blocks : by_bits);
            const uint64_t cap = bound + (uint64_t)HJD_REC_RUN * im.nsub_cap * nu + 64; // every k_write lane's run ends in a whole burst
            im.rec_cap = (uint32_t)(cap > 0xFFFFFF00ull ? 0xFFFFFF00ull : cap) & ~(uint32_t)(HJD_REC_RUN - 1);
        }
        im.rec_off = rec_cur;
        rec_cur += im.rec_cap;
        // dequantisation tables as the reference USES them: Y -> Y's, Cb -> Cr's (loadjpg.cpp:984), Cr -> Cr's;
        // stored by slot u*8+v (where DeZigZag + TransformArray put zig-zag entry k, loadjpg.cpp:156-180)
        for (int k = 0; k < 64; k++) {
            im.qt[0][kZzToSlot[k]] = j.qtable[1][k];
            im.qt[1][kZzToSlot[k]] = d->conformant ? j.qtable[2][k] : j.qtable[3][k]; // conformant mode: Cb's own table
            im.qt[2][kZzToSlot[k]] = j.qtable[3][k];
        }
        b->dev_index[i] = (int)b->img.size();
        b->img.push_back(im);
        b->out_bytes.push_back(obytes);
        b->hdr_bytes.push_back(j.header_bytes);
        b->pixels += (uint64_t)im.width * im.height;
        b->scan_bytes += im.scan_len;
        b->alg_bytes += (uint64_t)im.scan_len + j.header_bytes + (uint64_t)im.width * im.height * 3;
    }
    b->nimg = (int)b->img.size();
    for (auto &im : b->img) im.wgsum0 = im.write_small ? b->nwritewg + im.writewg0 : im.writewg0; // wgsum[]: the large workgroups' totals, then the small ones'
    b->rst_images.clear();
    for (int k = 0; k < b->nimg; k++)
        if (b->img[k].restart) b->rst_images.push_back((uint32_t)k);
    {   // Node space of the segment-parallel emulation (hjd_seq.hip: candidates x block-in-MCU labels, 40 bytes a node), in image
        // order until the budget is spent; an image without any is left to the one-lane kernel.  hjd_options.debug_flags HJD_DBG_NO_SEG_EMULATION: test switch.
        // The budget is counted in NODES and capped in bytes: 2 M nodes x 40 B = 80 MB per batch at most (round 4 reserved up to
        // 16 M nodes = 640 MB, up front, for every image with DRI whether or not it is ever flagged dirty); images beyond it fall
        // back to the one-lane kernel.
        size_t pos_cur = 0, node_cur = 0;
        const size_t budget = ((size_t)80 << 20) / 40;
        const bool off = d->no_seg_emulation || d->dirty_mode != 0;
        for (uint32_t q : b->rst_images) {
            HjdImage &im = b->img[q];
            const uint32_t cap = im.mark_cap + im.n_rst / 8u; // a well-formed scan holds n_rst of them
            const size_t nodes = (size_t)(cap + 1u) * im.nb;
            im.seg_cap = im.seg_off = im.seg_node = 0u;
            im.seg_ri = im.ri_blocks ? im.ri_blocks : (im.nblocks + im.n_rst) / (im.n_rst + 1u);
            if (off || node_cur + nodes > budget || cap >= (1u << 24)) continue;
            im.seg_cap = cap;
            im.seg_off = (uint32_t)pos_cur;
            im.seg_node = (uint32_t)node_cur;
            pos_cur += cap;
            node_cur += nodes;
        }
        b->seg_npos = pos_cur;
        b->seg_nnodes = node_cur;
    }
    b->seq_pending = false;
    b->emulate_in_run = false; // new contents
    b->emul_enqueued = false;
    b->raw_bytes = raw_cur;
    b->ustream_bytes = us_cur;
    b->nblk_entries = blk_cur;
    b->nrec = rec_cur;
    b->out_total = out_cur;
    b->state.assign(b->nimg, HjdImageState{});
    HjdParams &P = b->P;
    P.conformant = d->conformant ? 1u : 0u;
    P.gate_emulated = 0u;
    P.subseq_bits = S;
    P.log2_words = 0;
    while ((32u << P.log2_words) < S) P.log2_words++;
    P.sync_overlap = K;
    {   // dynamic LDS of the entropy kernels: primary and secondary tables, per-workgroup state, then one
        // staging window of phase_words + 2 words per lane
        P.phase_words = wpp;
        P.write_windows = wwn;
        P.lds_sec_word = max_nlut * HJD_LUT_SIZE / 2;
        P.lds_state_word = (P.lds_sec_word + max_nsec * (HJD_SEC_SIZE / 2) + 15) & ~15u; // 64-byte aligned: k_write's slot table is addressed by OR
        P.lds_sec_word_s = max_nlut * HJD_LUT_SIZE_S; // two u16 entries per index
        P.lds_state_word_s = (P.lds_sec_word_s + max_nsec_s * (HJD_SEC_SIZE_S / 2) + 3) & ~3u;
        // k_sync: five word arrays + the u8 marks + 96 words (scratch, block tables), then one staging window per slot (the first
        // 2 x HJD_SYNC_WG words of which double as the work list)
        P.lds_bytes_sync = (P.lds_state_word_s + 5 * HJD_SYNC_WG + HJD_SYNC_WG / 4 + 96 + (HJD_SYNC_WG / 64) * (wpp + 2) * 64) * 4;
        P.lds_bytes_sync_seq = (P.lds_state_word_s + 48 + (wpp + 2) * 64) * 4;
        P.lds_bytes_sync_small = (P.lds_state_word_s + 5 * HJD_SYNC_WG_SMALL + HJD_SYNC_WG_SMALL / 4 + 96 + (HJD_SYNC_WG_SMALL / 64) * (wpp + 2) * 64) * 4;
        // k_write: scan scratch, slot table, block table, the lanes' record bursts, then the workgroup's rows ([word][row], rows + 1 apart)
        P.lds_bytes_write = (P.lds_state_word + 96 + HJD_REC_RUN * HJD_WRITE_WG + HJD_ENT_CAP + (S / 32 + 2) * (HJD_WRITE_WG / nu + 1)) * 4;
        P.lds_bytes_write_small = (P.lds_state_word + 96 + HJD_REC_RUN * HJD_WRITE_WG_SMALL + HJD_ENT_CAP_SMALL +
                                   (S / 32 + 2) * (HJD_WRITE_WG_SMALL / (nu <= HJD_WRITE_WG_SMALL ? nu : HJD_WRITE_WG_SMALL) + 1)) * 4;
    }
    if (b->dry || !b->nimg) return HJD_OK;

    int rcs = HJD_OK;
    auto need = [&](int r) { if (r != HJD_OK && rcs == HJD_OK) rcs = r; };
    need(slab_reserve(b, b->d_img, sizeof(HjdImage) * b->nimg));
    need(slab_reserve(b, b->d_st, sizeof(HjdImageState) * b->nimg));
    need(slab_reserve(b, b->d_tabs, sizeof(HjdTableSet) * tabs.size()));
    need(slab_reserve(b, b->d_raw, raw_cur));
    need(slab_reserve(b, b->d_ustream, us_cur));
    need(slab_reserve(b, b->d_out, out_cur));
    need(slab_reserve(b, b->d_chunk_kept, sizeof(uint32_t) * b->nchunks));
    need(slab_reserve(b, b->d_chunk_mark, sizeof(uint32_t) * b->nchunks));
    need(slab_reserve(b, b->d_mpos, sizeof(uint32_t) * (b->nmarks + 1)));
    need(slab_reserve(b, b->d_mval, b->nmarks + 1));
    need(slab_reserve(b, b->d_substate, sizeof(uint4) * b->nsub));
    need(slab_reserve(b, b->d_subcnt, sizeof(uint32_t) * b->nsub));
    need(slab_reserve(b, b->d_submk, sizeof(uint32_t) * b->nsub));
    P.ck_scratch = (uint64_t)b->nsub * nph; // canonical entries first, then k_sync's private ones for its warm-up slots
    need(slab_reserve(b, b->d_ckpt, sizeof(uint2) * ((size_t)b->nsub * nph + (size_t)b->nsyncwg * nph * K + 1)));
    {   // Sibling trajectories (HjdParams::tk_n): when the whole batch is a few dozen synchronisation workgroups the chip sits idle
        // under k_sync's latency chain — further decodes of every subsequence cost no time then, and a refinement walk falls into
        // step with one of up to twelve trajectories instead of one.  (128 workgroups in all: a launch of 16 4K images, about 90
        // workgroups, gets none — the bench runs four such launches side by side, and one sibling each cost it 4 %.)
        // hjd_options.debug_flags (HJD_DBG_SIBLINGS): tuning / test switch.
        uint32_t tk = 0, nbmax = 3; // as many siblings (2 nb - 1 = every other label, at a block's start and one symbol into it; nb - 1 = every
                                    // other label; 3, 2, 1) as leave the whole launch on the chip at once
        for (auto &im : b->img) nbmax = im.nb > nbmax ? im.nb : nbmax;
        const uint32_t limit = d->sib_limit;
        for (uint32_t cand : {2u * nbmax - 1u, nbmax - 1u, 3u, 2u, 1u})
            if (tk == 0 && cand <= HJD_MAX_SIBLINGS && b->nsyncwg != 0 && b->nsyncwg * (cand + 1u) <= limit) tk = cand;
        if (d->siblings >= 0) tk = (uint32_t)d->siblings; // hjd_options.debug_flags: exactly that many
        if (b->nsyncwg == 0) tk = 0;
        P.tk_n = tk;
        P.tk_stride = (uint64_t)b->nsub * nph;
        P.tk_nwg = 0;
        if (tk) {
            need(slab_reserve(b, b->d_tkck, sizeof(uint2) * ((size_t)b->nsub * nph * tk + 1)));
            const size_t before = b->cap[(const void *)&b->d_tkflags];
            need(slab_reserve(b, b->d_tkflags, sizeof(uint32_t) * ((size_t)b->nsyncwg * tk + 1)));
            if (rcs == HJD_OK && b->cap[(const void *)&b->d_tkflags] != before) { // a fresh slab: no flag may look like a run's epoch
                HIPCHK(hipMemsetAsync(b->d_tkflags, 0, b->cap[(const void *)&b->d_tkflags], up));
            }
        }
    }
    {   // a fresh slab of repair stamps must not look like a launch's stamp (epochs start at 1)
        const size_t before = b->cap[(const void *)&b->d_repflags];
        need(slab_reserve(b, b->d_repflags, sizeof(uint32_t) * (b->nrepwg + 1)));
        if (rcs == HJD_OK && b->cap[(const void *)&b->d_repflags] != before) HIPCHK(hipMemsetAsync(b->d_repflags, 0, b->cap[(const void *)&b->d_repflags], up));
    }
    {
        const size_t before = b->cap[(const void *)&b->d_syncflags];
        need(slab_reserve(b, b->d_syncflags, sizeof(uint32_t) * (b->nsyncwg + 1)));
        if (rcs == HJD_OK && b->cap[(const void *)&b->d_syncflags] != before) HIPCHK(hipMemsetAsync(b->d_syncflags, 0, b->cap[(const void *)&b->d_syncflags], up));
    }
    if (b->seg_nnodes) need(slab_reserve(b, b->d_seg, seg_slab_size(b)));
    need(slab_reserve(b, b->d_wgsum, sizeof(uint2) * (b->nwritewg + b->nwritewg_small)));
    need(slab_reserve(b, b->d_rec, sizeof(uint32_t) * (rec_cur + HJD_REC_RUN)));
    need(slab_reserve(b, b->d_blk, sizeof(uint2) * (blk_cur + 1)));
    need(slab_reserve(b, b->d_dca, sizeof(int16_t) * (blk_cur + 2)));
    need(slab_reserve(b, b->d_dcsum, sizeof(uint32_t) * 3 * (b->ndcchunks + 1)));
    if (rcs != HJD_OK) return rcs;
    HIPCHK(hipMemsetAsync(b->d_blk, 0, sizeof(uint2) * (blk_cur + 1), up));
    std::vector<uint32_t> maps;
    {   // workgroup -> image for every kernel's workgroup numbering; IDCT groups also get their first block
        size_t off_chunk = 0, off_sync, off_rep, off_write, off_dc, off_group;
        auto fill = [&](uint32_t HjdImage::*first, uint32_t HjdImage::*count) {
            const size_t at = maps.size();
            for (int k = 0; k < b->nimg; k++) {
                if (maps.size() != at + b->img[k].*first) return (size_t)-1;
                maps.insert(maps.end(), b->img[k].*count, (uint32_t)k);
            }
            return at;
        };
        off_chunk = fill(&HjdImage::chunk0, &HjdImage::nchunks);
        // the two numberings of the synchronisation workgroups (HjdImage::sync_small)
        auto fill_sync = [&](uint32_t small) {
            const size_t at = maps.size();
            for (int k = 0; k < b->nimg; k++) {
                if (b->img[k].sync_small != small || !b->img[k].nsyncwg) continue;
                if (maps.size() != at + b->img[k].syncwg0) return (size_t)-1;
                maps.insert(maps.end(), b->img[k].nsyncwg, (uint32_t)k);
            }
            return at;
        };
        off_sync = fill_sync(0u);
        const size_t off_sync_small = fill_sync(1u);
        if (off_sync_small == (size_t)-1) return HJD_ERR_INTERNAL;
        off_rep = fill(&HjdImage::repwg0, &HjdImage::nrepwg);
        auto fill_write = [&](uint32_t small) {
            const size_t at = maps.size();
            for (int k = 0; k < b->nimg; k++) {
                if (b->img[k].write_small != small) continue;
                if (maps.size() != at + b->img[k].writewg0) return (size_t)-1;
                maps.insert(maps.end(), b->img[k].nwritewg, (uint32_t)k);
            }
            return at;
        };
        off_write = fill_write(0u);
        const size_t off_write_small = fill_write(1u);
        if (off_write_small == (size_t)-1) return HJD_ERR_INTERNAL;
        off_dc = fill(&HjdImage::dcchunk0, &HjdImage::ndcchunk);
        if (off_chunk == (size_t)-1 || off_sync == (size_t)-1 || off_rep == (size_t)-1 || off_write == (size_t)-1 || off_dc == (size_t)-1)
            return HJD_ERR_INTERNAL;
        const size_t off_rst = maps.size(); // the sequential emulation's candidates
        maps.insert(maps.end(), b->rst_images.begin(), b->rst_images.end());
        const size_t off_rstg = maps.size(); // ... and their back-end workgroups: the gated second launch runs exactly these
        for (uint32_t q : b->rst_images)
            for (uint32_t g = 0; g < b->img[q].ngroups; g++) maps.push_back(b->img[q].group0 + g);
        b->nrst_groups = (uint32_t)(maps.size() - off_rstg);
        if (maps.size() & 1) maps.push_back(0); // the uint2 table that follows is 8-byte aligned
        off_group = maps.size();
        for (int k = 0; k < b->nimg; k++) {
            const HjdImage &im = b->img[k];
            for (uint32_t g = 0; g < im.ngroups; g++) {
                const uint32_t mrow = g / im.groups_per_row, gx = g % im.groups_per_row;
                maps.push_back((uint32_t)k);
                maps.push_back((uint32_t)(im.blk_off + ((size_t)mrow * im.mcux + (size_t)gx * HJD_GROUP_MCUS) * im.nb));
            }
        }
        if (blk_cur >= 0xFFFFFFFFull) return HJD_ERR_UNSUPPORTED;
        need(slab_reserve(b, b->d_maps, sizeof(uint32_t) * (maps.size() + 2)));
        if (rcs != HJD_OK) return rcs;
        b->P.map_chunk = b->d_maps + off_chunk;
        b->P.map_syncwg = b->d_maps + off_sync;
        b->P.map_syncwg_small = b->d_maps + off_sync_small;
        b->P.map_repwg = b->d_maps + off_rep;
        b->P.map_writewg = b->d_maps + off_write;
        b->P.map_writewg_small = b->d_maps + off_write_small;
        b->P.map_dcchunk = b->d_maps + off_dc;
        b->P.map_group = reinterpret_cast<const uint2 *>(b->d_maps + off_group);
        b->P.rst_list = b->d_maps + off_rst;
        b->P.rst_groups = b->d_maps + off_rstg;
    }
    {   // stage everything that goes up in ONE pinned buffer: [scan bytes | image descriptors | tables | maps]
        const size_t o_img = align_up(raw_cur, 256), o_tab = align_up(o_img + sizeof(HjdImage) * b->nimg, 256);
        const size_t o_map = align_up(o_tab + sizeof(HjdTableSet) * tabs.size(), 256);
        const size_t total = o_map + sizeof(uint32_t) * maps.size() + 256;
        if (total > b->h_stage_cap) {
            if (b->h_stage) (void)hipHostFree(b->h_stage);
            b->h_stage = nullptr;
            b->h_stage_cap = 0;
            HIPCHK(hipHostMalloc((void **)&b->h_stage, total + total / 16, hipHostMallocDefault));
            b->h_stage_cap = total + total / 16;
        }
        uint8_t *h = b->h_stage;
        if (!parallel_for(b->n, raw_cur, [&](int i) {
                const int di = b->dev_index[i];
                if (di < 0) return;
                const HjdImage &im = b->img[di];
                memcpy(h + im.raw_off, parsed[i].stream, parsed[i].stream_len);
                const size_t slab = align_up((size_t)im.nchunks * HJD_UNSTUFF_CHUNK + 16, 256);
                memset(h + im.raw_off + parsed[i].stream_len, 0, slab - parsed[i].stream_len); // the kernels read whole 16-byte pieces
            }))
            return HJD_ERR_INTERNAL;
        memcpy(h + o_img, b->img.data(), sizeof(HjdImage) * b->nimg);
        memcpy(h + o_tab, tabs.data(), sizeof(HjdTableSet) * tabs.size());
        memcpy(h + o_map, maps.data(), sizeof(uint32_t) * maps.size());
        HIPCHK(hipMemcpyAsync(b->d_raw, h, raw_cur, hipMemcpyHostToDevice, up));
        HIPCHK(hipMemcpyAsync(b->d_img, h + o_img, sizeof(HjdImage) * b->nimg, hipMemcpyHostToDevice, up));
        HIPCHK(hipMemcpyAsync(b->d_tabs, h + o_tab, sizeof(HjdTableSet) * tabs.size(), hipMemcpyHostToDevice, up));
        HIPCHK(hipMemcpyAsync(b->d_maps, h + o_map, sizeof(uint32_t) * maps.size(), hipMemcpyHostToDevice, up));
    }
    HIPCHK(hipMemsetAsync(b->d_ustream, 0, us_cur, up));
    HIPCHK(hipMemsetAsync(b->d_st, 0, sizeof(HjdImageState) * b->nimg, up));
    HIPCHK(hipMemsetAsync(b->d_substate, 0, sizeof(uint4) * (b->nsub ? b->nsub : 1), up));
    HIPCHK(hipMemsetAsync(b->d_subcnt, 0, sizeof(uint32_t) * (b->nsub ? b->nsub : 1), up));
    HIPCHK(hipMemsetAsync(b->d_submk, 0, sizeof(uint32_t) * (b->nsub ? b->nsub : 1), up));

    P.img = b->d_img;
    P.st = b->d_st;
    P.tabs = b->d_tabs;
    P.raw = b->d_raw;
    P.ustream = b->d_ustream;
    P.chunk_kept = b->d_chunk_kept;
    P.chunk_mark = b->d_chunk_mark;
    P.mpos = b->d_mpos;
    P.mval = b->d_mval;
    P.substate = b->d_substate;
    P.subcnt = b->d_subcnt;
    P.submk = b->d_submk;
    P.ckpt = b->d_ckpt;
    P.tk_ckpt = b->d_tkck;
    P.tk_flags = b->d_tkflags;
    P.rep_flags = b->d_repflags;
    P.sync_flags = b->d_syncflags;
    P.nrst = b->seg_nnodes ? (uint32_t)b->rst_images.size() : 0u; // 0: no segment-parallel emulation in this batch
    P.seg_cut = 256u;
    if (d->seg_cut) P.seg_cut = 0x80000000u | d->seg_cut; // test switch: exactly this many (1: every chain meets a cut-off segment at once)
    if (b->seg_nnodes) {
        const size_t nrst = b->rst_images.size();
        P.seg_w = reinterpret_cast<uint4 *>(b->d_seg);
        P.seg_base = P.seg_w + b->seg_nnodes;
        P.seg_next = reinterpret_cast<uint32_t *>(P.seg_base + b->seg_nnodes);
        P.seg_chain = P.seg_next + b->seg_nnodes;
        P.seg_pos = P.seg_chain + b->seg_nnodes;
        P.seg_cnt = P.seg_pos + b->seg_npos;
        P.seg_nchain = P.seg_cnt + nrst;
        P.seg_plan = P.seg_nchain + nrst;
        HIPCHK(hipMemsetAsync(P.seg_plan + 2 * (nrst + 1), 0, 2 * sizeof(uint32_t), up)); // the counters: every run leaves them at 0
    }
    P.dbg = nullptr;
    if (d->sync_prof) {
        need(slab_reserve(b, b->d_dbg, sizeof(unsigned long long) * 32 * (b->nsyncwg + 1)));
        if (rcs != HJD_OK) return rcs;
        HIPCHK(hipMemsetAsync(b->d_dbg, 0, sizeof(unsigned long long) * 32 * (b->nsyncwg + 1), up));
        P.dbg = b->d_dbg;
    }
    P.wgsum = b->d_wgsum;
    P.rec = b->d_rec;
    P.blk = b->d_blk;
    P.dca = b->d_dca;
    P.dcsum = b->d_dcsum;
    P.ndcchunks = b->ndcchunks;
    P.out = b->d_out;
    P.nimg = b->nimg;
    P.off_chunk = P.off_syncwg = P.off_syncwg_small = P.off_repwg = P.off_writewg = P.off_writewg_small = P.off_img = P.off_dcchunk = P.off_group = 0;
    {   // sub-batch ranges: contiguous runs of images with about equal pixel counts
        uint64_t total_px = 0;
        for (auto &im : b->img) total_px += (uint64_t)im.width * im.height;
        int ns = d->streams;
        if (ns == 0) { // measured on 4K images, one caller stream: 8..32 images +4 % with two ranges, 64 images +17 % with three;
                       // large batches: ranges of about 130 Mpixel (the size of launch that overlaps best), at most 16 — dealt to
                       // four streams by batch_enqueue
            ns = total_px >= 400000000ull ? (int)((total_px + 66000000ull) / 133000000ull) : (total_px >= 60000000ull ? 2 : 1);
            if (ns > 16) ns = 16;
        }
        if (ns > b->nimg) ns = b->nimg;
        if (ns < 1) ns = 1;
        int i0 = 0;
        uint64_t acc = 0;
        for (int k = 0; k < ns; k++) {
            int i1 = i0;
            const uint64_t target = total_px * (uint64_t)(k + 1) / (uint64_t)ns;
            while (i1 < b->nimg && (acc < target || i1 == i0) && (b->nimg - i1) > (ns - 1 - k)) {
                acc += (uint64_t)b->img[i1].width * b->img[i1].height;
                i1++;
            }
            if (k == ns - 1) i1 = b->nimg;
            hjd_batch::Range r{};
            r.i0 = i0;
            r.i1 = i1;
            const HjdImage &a = b->img[i0], &z = b->img[i1 - 1];
            r.chunk0 = a.chunk0;       r.nchunk = z.chunk0 + z.nchunks - a.chunk0;
            {   // synchronisation workgroups of the range, per numbering (images of one kind are numbered in image order)
                bool have_big = false, have_small = false;
                for (int q = i0; q < i1; q++) {
                    const HjdImage &im = b->img[q];
                    if (im.sync_small == 2u) continue; // k_sync_seq: below
                    if (im.sync_small) {
                        if (!have_small) r.ssyncwg0 = im.syncwg0;
                        have_small = true;
                        r.nssyncwg = im.syncwg0 + im.nsyncwg - r.ssyncwg0;
                    } else {
                        if (!have_big) r.syncwg0 = im.syncwg0;
                        have_big = true;
                        r.nsyncwg = im.syncwg0 + im.nsyncwg - r.syncwg0;
                    }
                }
            }
            r.repwg0 = a.repwg0;       r.nrepwg = z.repwg0 + z.nrepwg - a.repwg0;
            {   // final-pass workgroups of the range, per numbering
                bool have_big = false, have_small = false;
                for (int q = i0; q < i1; q++) {
                    const HjdImage &im = b->img[q];
                    if (im.sync_small && !im.write_small) r.side_feeds_big = 1; // synchronised by a small kernel, written by the large one
                    if (im.write_small) {
                        if (!have_small) r.swritewg0 = im.writewg0;
                        have_small = true;
                        r.nswritewg = im.writewg0 + im.nwritewg - r.swritewg0;
                    } else {
                        if (!have_big) r.writewg0 = im.writewg0;
                        have_big = true;
                        r.nwritewg = im.writewg0 + im.nwritewg - r.writewg0;
                    }
                }
            }
            r.dc0 = a.dcchunk0;        r.ndc = z.dcchunk0 + z.ndcchunk - a.dcchunk0;
            r.group0 = a.group0;       r.ngroup = z.group0 + z.ngroups - a.group0;
            r.blk0 = (size_t)a.blk_off;
            r.nblk = (size_t)(z.blk_off + z.nblocks + 1 - a.blk_off);
            for (size_t q = 0; q < b->rst_images.size(); q++)
                if ((int)b->rst_images[q] >= i0 && (int)b->rst_images[q] < i1) {
                    if (!r.nrst) r.rst0 = (uint32_t)q;
                    r.nrst++;
                }
            b->ranges.push_back(r);
            i0 = i1;
        }
        {   // k_sync_seq: the thumbnails of every range in groups of 64 that share table set and sampling
            b->seqlist.clear();
            for (auto &r : b->ranges) {
                r.seq0 = (uint32_t)(b->seqlist.size() / 64u);
                std::map<uint64_t, std::vector<uint32_t>> groups;
                for (int q = r.i0; q < r.i1; q++)
                    if (b->img[q].sync_small == 2u) groups[((uint64_t)b->img[q].tabset << 8) | b->img[q].nby].push_back((uint32_t)q);
                for (auto &kv : groups)
                    for (size_t at = 0; at < kv.second.size(); at += 64) {
                        for (size_t t = 0; t < 64; t++) b->seqlist.push_back(at + t < kv.second.size() ? kv.second[at + t] : 0xFFFFFFFFu);
                    }
                r.nseq = (uint32_t)(b->seqlist.size() / 64u) - r.seq0;
            }
            b->P.seq_list = nullptr;
            if (!b->seqlist.empty()) {
                need(slab_reserve(b, b->d_seqlist, sizeof(uint32_t) * b->seqlist.size()));
                if (rcs != HJD_OK) return rcs;
                // (pageable source: the copy is staged by the runtime before the call returns)
                HIPCHK(hipMemcpyAsync(b->d_seqlist, b->seqlist.data(), sizeof(uint32_t) * b->seqlist.size(), hipMemcpyHostToDevice, up));
                b->P.seq_list = b->d_seqlist;
            }
        }
        bool side_needed = false;
        for (auto &r : b->ranges) side_needed |= (r.nssyncwg || r.nseq) && r.nsyncwg;
        if (side_needed) {
            while (b->small_side.size() < b->ranges.size()) {
                hjd_batch::SideStream ss;
                HIPCHK(hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming));
                HIPCHK(hipEventCreateWithFlags(&ss.done, hipEventDisableTiming));
                b->small_side.push_back(ss);
            }
        }
        // the decoder's three streams: ranges 1..3 first, what is left to the side streams of ranges 0, 1, ... (none: that range's
        // small kernels follow its large one on the range's own stream) — a batch and the caller's stream stay within four queues
        for (size_t k = 0; k < b->small_side.size(); k++) {
            const size_t nr = b->ranges.size() > 4 ? 4 : (b->ranges.size() > 1 ? b->ranges.size() : 1), idx = nr - 1 + k; // queues the ranges take
            b->small_side[k].st = idx < 3 ? b->dec->lib_streams[idx] : nullptr;
        }
        if (!b->rst_images.empty() && !b->seq_done) {
            HIPCHK(hipEventCreateWithFlags(&b->seq_done, hipEventDisableTiming));
        }
        if (b->ranges.size() > 1) {
            while (b->join_ev.size() < b->ranges.size()) { // (the ranges' streams are the decoder's: batch_enqueue)
                hipEvent_t e = nullptr, se = nullptr;
                HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                HIPCHK(hipEventCreateWithFlags(&se, hipEventDisableTiming));
                b->join_ev.push_back(e);
                b->stagger_ev.push_back(se);
            }
            if (!b->fork_ev) HIPCHK(hipEventCreateWithFlags(&b->fork_ev, hipEventDisableTiming));
        }
    }
    return HJD_OK;
}

static int batch_reset_parsed(hjd_batch *b, int n, const std::vector<hjd_jpeg_data> &parsed, const std::vector<int> &pstatus,
                              int out_format, int *status, hipStream_t up, bool wait)
{
    b->n = n;
    b->out_format = out_format;
    b->parse_status = pstatus;
    b->dev_index.assign(n, -1);
    b->tev_n.clear();
    int rc = batch_build(b, parsed, up);
    if (rc == HJD_OK && wait && hipStreamSynchronize(up) != hipSuccess) rc = HJD_ERR_INTERNAL;
    if (status)
        for (int i = 0; i < n; i++) status[i] = b->parse_status[i];
    return rc;
}

static hjd_batch *batch_from_parsed(hjd_decoder *d, int n, const std::vector<hjd_jpeg_data> &parsed,
                                    const std::vector<int> &pstatus, int out_format, int *status)
{
    hjd_batch *b = new hjd_batch();
    b->dec = d;
    b->device = d->device;
    const int rc = batch_reset_parsed(b, n, parsed, pstatus, out_format, status, nullptr, true);
    if (rc != HJD_OK) {
        hjd_set_last_status(rc);
        hjd_batch_destroy(b);
        return nullptr;
    }
    return b;
}

extern "C" int hjd_layout_query(const hjd_jpeg_data *j, int subseq_bits, int sync_overlap, int out_format, hjd_layout_info *o)
{
    if (!j || !o) return HJD_ERR_ARG;
    hjd_decoder d; // never touches a device
    if (subseq_bits) d.subseq_bits = (uint32_t)subseq_bits;
    d.subseq_auto = d.overlap_auto = false; // the layout of exactly this size (0 = 2048)
    {
        uint32_t k = 16384 / d.subseq_bits;
        d.sync_overlap = k < 4 ? 4 : (k > 32 ? 32 : k);
        if (sync_overlap) d.sync_overlap = sync_overlap < 0 ? 0u : (uint32_t)sync_overlap;
    }
    hjd_batch b;
    b.dec = &d;
    b.dry = true;
    b.n = 1;
    b.out_format = out_format;
    b.parse_status.assign(1, HJD_OK);
    b.dev_index.assign(1, -1);
    std::vector<hjd_jpeg_data> parsed(1, *j);
    const int rc = batch_build(&b, parsed, nullptr);
    if (rc != HJD_OK) return rc;
    if (b.parse_status[0] != HJD_OK || b.img.empty()) return b.parse_status[0] ? b.parse_status[0] : HJD_ERR_UNSUPPORTED;
    const HjdImage &im = b.img[0];
    memset(o, 0, sizeof(*o));
    o->subseq_bits = b.P.subseq_bits;
    o->sync_overlap = b.P.sync_overlap;
    o->phase_words = b.P.phase_words;
    o->sync_wg = HJD_SYNC_WG;
    o->write_wg = HJD_WRITE_WG;
    o->group_mcus = HJD_GROUP_MCUS;
    o->unstuff_chunk = HJD_UNSTUFF_CHUNK;
    o->dc_chunk = HJD_DC_CHUNK;
    o->rec_run = HJD_REC_RUN;
    o->nblocks = im.nblocks;
    o->mcux = im.mcux;
    o->mcuy = im.mcuy;
    o->nb = im.nb;
    o->nchunks = im.nchunks;
    o->nsub_cap = im.nsub_cap;
    o->nsyncwg = im.nsyncwg;
    o->nrepwg = im.nrepwg;
    o->nwritewg = im.nwritewg;
    o->ndcchunk = im.ndcchunk;
    o->ngroups = im.ngroups;
    o->groups_per_row = im.groups_per_row;
    o->mark_cap = im.mark_cap;
    o->rec_cap = im.rec_cap;
    o->out_stride = im.out_stride;
    o->raw_bytes = b.raw_bytes;
    o->ustream_bytes = b.ustream_bytes;
    o->blk_entries = b.nblk_entries + 1;       // what batch_build allocates
    o->dca_entries = b.nblk_entries + 2;
    o->rec_entries = b.nrec + HJD_REC_RUN;
    o->out_bytes = b.out_total;
    o->substate_entries = b.nsub;
    o->wgsum_entries = (size_t)b.nwritewg + b.nwritewg_small;
    o->dcsum_entries = 3 * ((size_t)b.ndcchunks + 1);
    {   // what batch_build allocates
        const size_t nph = (b.P.subseq_bits / 32) / b.P.phase_words;
        o->ckpt_entries = (size_t)b.nsub * nph + (size_t)b.nsyncwg * nph * b.P.sync_overlap + 1;
    }
    o->seg_cap = im.seg_cap;
    o->seg_nodes = b.seg_nnodes;
    o->seg_slab_bytes = seg_slab_size(&b);
    o->lds_bytes_sync = b.P.lds_bytes_sync;
    o->lds_bytes_write = b.P.lds_bytes_write;
    o->lds_state_word = b.P.lds_state_word;
    o->lds_state_word_sync = b.P.lds_state_word_s;
    b.dec = nullptr;
    return HJD_OK;
}

extern "C" hjd_batch *hjd_batch_create(hjd_decoder *d)
{
    if (!d) {
        hjd_set_last_status(HJD_ERR_ARG);
        return nullptr;
    }
    hjd_batch *b = new hjd_batch();
    b->dec = d;
    b->device = d->device;
    return b;
}

extern "C" int hjd_batch_reset(hjd_batch *b, int n, const uint8_t *const *jpg, const size_t *jpg_len, int out_format,
                               int *status, void *hip_stream)
{
    if (!b || n < 0 || (n && (!jpg || !jpg_len)) || (out_format != HJD_OUT_RGB && out_format != HJD_OUT_BMP)) return HJD_ERR_ARG;
    std::vector<hjd_jpeg_data> parsed(n);
    std::vector<int> pst(n);
    const int pf = b->dec->conformant ? HJD_PARSE_CONFORMANT : 0;
    size_t total = 0;
    for (int i = 0; i < n; i++) total += jpg[i] ? jpg_len[i] : 0;
    if (!parallel_for(n, total, [&](int i) { pst[i] = jpg[i] ? hjd_parse_header_ex(jpg[i], jpg_len[i], &parsed[i], pf) : HJD_ERR_ARG; }))
        return HJD_ERR_INTERNAL;
    return batch_reset_parsed(b, n, parsed, pst, out_format, status, (hipStream_t)hip_stream, false);
}

extern "C" hjd_batch *hjd_batch_prepare(hjd_decoder *d, int n, const uint8_t *const *jpg, const size_t *jpg_len,
                                        int out_format, int *status)
{
    if (!d || n < 0 || (n && (!jpg || !jpg_len)) || (out_format != HJD_OUT_RGB && out_format != HJD_OUT_BMP)) {
        hjd_set_last_status(HJD_ERR_ARG);
        return nullptr;
    }
    std::vector<hjd_jpeg_data> parsed(n);
    std::vector<int> pst(n);
    const int pf = d->conformant ? HJD_PARSE_CONFORMANT : 0;
    size_t total = 0;
    for (int i = 0; i < n; i++) total += jpg[i] ? jpg_len[i] : 0;
    if (!parallel_for(n, total, [&](int i) { pst[i] = jpg[i] ? hjd_parse_header_ex(jpg[i], jpg_len[i], &parsed[i], pf) : HJD_ERR_ARG; })) {
        hjd_set_last_status(HJD_ERR_INTERNAL);
        return nullptr;
    }
    return batch_from_parsed(d, n, parsed, pst, out_format, status);
}

// The batch-level twin of hjd_jpeg_decode_hw: N images whose headers the CALLER has parsed (the reference's own JpegParseHeader
// through include/hjd_reference_shim.h, say).  Every struct gets the checks hjd_parse_header applies (batch_build).
extern "C" hjd_batch *hjd_batch_prepare_parsed(hjd_decoder *d, int n, const hjd_jpeg_data *jdata, int out_format, int *status)
{
    if (!d || n < 0 || (n && !jdata) || (out_format != HJD_OUT_RGB && out_format != HJD_OUT_BMP)) {
        hjd_set_last_status(HJD_ERR_ARG);
        return nullptr;
    }
    std::vector<hjd_jpeg_data> parsed(jdata, jdata + n);
    std::vector<int> pst(n, HJD_OK);
    for (int i = 0; i < n; i++)
        if (!parsed[i].stream) pst[i] = HJD_ERR_ARG;
    return batch_from_parsed(d, n, parsed, pst, out_format, status);
}

// the kernel sequence; ev != nullptr records an event before each stage and one after the last
// the kernel sequence for one range of images; names != nullptr records an event before each stage
static int range_enqueue(hjd_batch *b, const hjd_batch::Range &r, hipStream_t s, bool from_resync, int *nstage,
                         const char **names, hipEvent_t *evs, hipEvent_t wait_before_sync = nullptr,
                         hipEvent_t record_stagger = nullptr, int stagger_after = 0, size_t side = 0)
{
    HjdParams P = b->P;
    P.off_chunk = r.chunk0;
    P.off_syncwg = r.syncwg0;
    P.off_syncwg_small = r.ssyncwg0;
    P.off_seqwg = r.seq0;
    P.off_repwg = r.repwg0;
    P.off_writewg = r.writewg0;
    P.off_writewg_small = r.swritewg0;
    P.off_img = (uint32_t)r.i0;
    P.off_dcchunk = r.dc0;
    P.off_group = r.group0;
    int k = 0;
    auto mark = [&](const char *name) -> int {
        if (names) {
            if (k < HJD_MAX_STAGES) {
                names[k] = name;
                if (hipEventRecord(evs[k], s) != hipSuccess) return HJD_ERR_INTERNAL;
            }
            k++;
        }
        return HJD_OK;
    };
    int rc;
    bool small_write_on_side = false;
    hipEvent_t side_done = nullptr;
    struct SideJoin { // joins a forked side stream to the run's stream on every way out of this function
        hjd_batch::SideStream *ss = nullptr;
        hipStream_t s = nullptr;
        ~SideJoin()
        {
            if (ss && hipEventRecord(ss->done, ss->st) == hipSuccess) (void)hipStreamWaitEvent(s, ss->done, 0);
        }
    } joiner;
    const int nrep = b->dec->sync_launches - 1; // repair launches; the last one also leaves k_write's totals
    if (!from_resync) {
        if ((rc = mark("unstuff"))) return rc; // k_unstuff_count also clears the per-image state
        if (hjd_launch_unstuff(P, r.nchunk, s)) return HJD_ERR_INTERNAL;
        // the images with a small workgroup of their own: beside the large ones when the range holds both kinds.  The side stream
        // forks HERE, right behind the unstuff stage: the thumbnails' kernels do not wait for another run's k_sync (the waits
        // below concern the large synchronisation kernel only)
        // (a process has four hardware queues: a batch split into three or four ranges keeps its streams within them — the side streams
        // of the last ranges are dropped, their small kernels follow the large one on the range's own stream; with eight streams for
        // four ranges BASELINE config 5 ran at 80 Gpixel/s against 145 for two ranges; and none at all when the decoder's recent runs
        // came on three or more caller streams: those runs overlap one another, a side stream each only adds queues to share)
        hjd_batch::SideStream *ss = ((r.nssyncwg || r.nseq) && r.nsyncwg && side < b->small_side.size() && b->small_side[side].st && !names && !b->dec->no_side && b->callers_seen < 3) ? &b->small_side[side] : nullptr;
        // the small images' final pass follows their synchronisation on the side stream — unless the large final pass needs what
        // the side stream computes (an image synchronised by a small kernel but written by the large one): then the side stream
        // joins before it
        small_write_on_side = ss && nrep == 0 && !r.side_feeds_big;
        if (ss) {
            if (hipEventRecord(ss->fork, s) != hipSuccess || hipStreamWaitEvent(ss->st, ss->fork, 0) != hipSuccess) return HJD_ERR_INTERNAL;
            joiner.ss = ss; // from here on every return joins the side stream to `s` (include/hjd.h: work on library streams is joined
                            // before hjd_batch_run returns — also when a launch fails half-way)
            joiner.s = s;
            if (hjd_launch_sync(P, r.nssyncwg, 3, ss->st)) return HJD_ERR_INTERNAL;
            if (hjd_launch_sync_seq(P, r.nseq, ss->st)) return HJD_ERR_INTERNAL;
            if (small_write_on_side && hjd_launch_write(P, r.nswritewg, ss->st, 1)) return HJD_ERR_INTERNAL;
            if (hipEventRecord(ss->done, ss->st) != hipSuccess) return HJD_ERR_INTERNAL;
        }
        // staggered ranges (batch_enqueue): this range's synchronisation — a latency chain that leaves most issue slots
        // empty — starts when the previous range has finished its own, so that it runs beside that range's final pass and
        // IDCT instead of beside three other synchronisation kernels
        // The waits below belong to no kernel: with a limit of two synchronisation kernels at a time and four caller streams a
        // run spends longer waiting for another run's k_sync than in its own unstuff stage.  They get a stage of their own
        // ("sync_wait"), so that the unstuff stage's interval is its kernels' and nothing else.
        const bool waits = wait_before_sync || (b->dec->sync_conc > 0 && !b->dry);
        if (waits && (rc = mark("sync_wait"))) return rc;
        if (wait_before_sync && hipStreamWaitEvent(s, wait_before_sync, 0) != hipSuccess) return HJD_ERR_INTERNAL;
        hipEvent_t conc_record = nullptr;
        if (b->dec->sync_conc > 0 && !b->dry) { // front-end concurrency limit: see hjd_decoder::sync_conc
            hjd_decoder *d = b->dec;
            std::lock_guard<std::mutex> lk(d->conc_mu);
            const uint64_t k = d->conc_runs++;
            if (!d->conc_ev[k & 15] && hipEventCreateWithFlags(&d->conc_ev[k & 15], hipEventDisableTiming) != hipSuccess) return HJD_ERR_INTERNAL;
            if (k >= (uint64_t)d->sync_conc && hipStreamWaitEvent(s, d->conc_ev[(k - d->sync_conc) & 15], 0) != hipSuccess) return HJD_ERR_INTERNAL;
            conc_record = d->conc_ev[k & 15];
        }
        if ((rc = mark("huff_sync"))) return rc;
        if (hjd_launch_sync(P, r.nsyncwg, 0, s)) return HJD_ERR_INTERNAL;
        // the gate for other runs' synchronisation kernels is THIS kernel's end, not the small kernels' (recorded before the join)
        if (conc_record && hipEventRecord(conc_record, s) != hipSuccess) return HJD_ERR_INTERNAL;
        if (ss && !small_write_on_side) {
            if (hipStreamWaitEvent(s, ss->done, 0) != hipSuccess) return HJD_ERR_INTERNAL;
            joiner.ss = nullptr; // joined
        } else if (!ss && (hjd_launch_sync(P, r.nssyncwg, 3, s) || hjd_launch_sync_seq(P, r.nseq, s))) {
            return HJD_ERR_INTERNAL;
        }
        side_done = ss ? ss->done : nullptr;
        if (record_stagger && stagger_after == 1 && hipEventRecord(record_stagger, s) != hipSuccess) return HJD_ERR_INTERNAL;
        if (nrep > 0 && (rc = mark("huff_repair"))) return rc;
        for (int l = 0; l < nrep; l++)
            if (hjd_launch_sync(P, r.nrepwg, l + 1 == nrep ? 2 : 1, s)) return HJD_ERR_INTERNAL;
    } else {
        if (hjd_launch_sync(P, r.nrepwg, 1, s)) return HJD_ERR_INTERNAL;
        if (hjd_launch_sync(P, r.nrepwg, 2, s)) return HJD_ERR_INTERNAL;
    }
    // a stage's concurrency limit (see hjd_decoder::idct_conc): the k-th run's kernel waits for that of run k - limit
    auto gate = [&](int limit, hipEvent_t *evs, uint64_t &runs, hipEvent_t &rec) -> int {
        rec = nullptr;
        if (limit <= 0 || b->dry || from_resync || names) return HJD_OK;
        hjd_decoder *d = b->dec;
        std::lock_guard<std::mutex> lk(d->conc_mu);
        const uint64_t k = runs++;
        if (!evs[k & 15] && hipEventCreateWithFlags(&evs[k & 15], hipEventDisableTiming) != hipSuccess) return HJD_ERR_INTERNAL;
        if (k >= (uint64_t)limit && hipStreamWaitEvent(s, evs[(k - limit) & 15], 0) != hipSuccess) return HJD_ERR_INTERNAL;
        rec = evs[k & 15];
        return HJD_OK;
    };
    hipEvent_t wrec = nullptr, irec = nullptr;
    if ((rc = mark("huff_write"))) return rc;
    if ((rc = gate(b->dec->write_conc, b->dec->write_ev, b->dec->write_runs, wrec))) return rc;
    if (hjd_launch_write(P, r.nwritewg, s)) return HJD_ERR_INTERNAL;
    if (wrec && hipEventRecord(wrec, s) != hipSuccess) return HJD_ERR_INTERNAL;
    if (small_write_on_side) {
        if (hipStreamWaitEvent(s, side_done, 0) != hipSuccess) return HJD_ERR_INTERNAL; // the side stream joins here
        joiner.ss = nullptr;
    } else if (hjd_launch_write(P, r.nswritewg, s, 1)) {
        return HJD_ERR_INTERNAL;
    }
    if (record_stagger && stagger_after == 2 && !from_resync && hipEventRecord(record_stagger, s) != hipSuccess) return HJD_ERR_INTERNAL;
    if ((rc = mark("dc_scan"))) return rc;
    if (hjd_launch_dcscan(P, r.ndc, s)) return HJD_ERR_INTERNAL;
    if ((rc = mark("idct_color"))) return rc;
    if ((rc = gate(b->dec->idct_conc, b->dec->idct_ev, b->dec->idct_runs, irec))) return rc;
    if (hjd_launch_idct_color(P, r.ngroup, s)) return HJD_ERR_INTERNAL;
    if (irec && hipEventRecord(irec, s) != hipSuccess) return HJD_ERR_INTERNAL;
    if ((rc = mark("end"))) return rc;
    if (nstage) *nstage = k;
    return HJD_OK;
}

// After the run proper (everything of it is on `s` or joined to it): the emulation of the reference on the restart streams the
// parallel path has just flagged DIRTY (hjd_seq.hip: four launches parallel over the resynchronisation points, one of the
// one-lane kernel for what those leave), then the back end once more for exactly those images — on the batch's side stream,
// which waits for the run.  All launches are gated on the device: for a batch whose restart streams are all clean they are six
// launches of workgroups that return at once (config 5, clean: 0.5 % of the step).
static int emulation_enqueue(hjd_batch *b, hipStream_t s, bool force = false)
{
    b->emul_enqueued = false;
    if (b->rst_images.empty() || b->dec->dirty_mode != 0 || !b->seq_done) return HJD_OK;
    // Not before a run of this batch has been seen to need them (hjd_batch_finish, which then enqueues them itself and sets
    // emulate_in_run for the later runs of the same contents): the six launches are gated on the device and return at once for
    // clean restart streams, but they are six dependent launches at the end of every run, and with several caller streams in
    // flight the command processor let a run start only when the run two before it (another stream's) had got through them —
    // BASELINE config 5, clean, on four caller streams: 148 Gpixel/s with them, 167-176 without (round 5; which of the six made
    // no difference, their number did).
    if (!force && !b->emulate_in_run) return HJD_OK;
    // On the run's own stream (round 5).  Rounds 3-4 put these launches on a side stream of the batch and joined it to the caller's
    // stream before hjd_batch_run returned — serial either way, but a stream more per batch: four caller streams with restart
    // streams in their batches were twelve streams on the process's four hardware queues, and a launch that shares a queue with
    // another batch's synchronisation kernel waits for it (BASELINE config 5 on four caller streams: 87 Gpixel/s against 137 as one batch).
    // parallel over the resynchronisations where the image has node space; the one-lane kernel takes what that leaves
    if (hjd_launch_seg_emulation(b->P, s)) return HJD_ERR_INTERNAL;
    if (hjd_launch_seq_decode(b->P, 0u, (uint32_t)b->rst_images.size(), s, b->P.nrst != 0u)) return HJD_ERR_INTERNAL;
    {   // the back end again, over the candidates' workgroups only (round 3 launched the whole span from the first to the last
        // candidate: a batch with restart streams at both ends ran a million workgroups that returned at once)
        HjdParams P = b->P;
        P.gate_emulated = 1u;
        if (hjd_launch_idct_color(P, b->nrst_groups, s)) return HJD_ERR_INTERNAL;
    }
    HIPCHK(hipEventRecord(b->seq_done, s)); // a later run of this batch on ANOTHER stream, and a re-prepare, wait for it
    b->emul_enqueued = true;
    b->seq_pending = true;
    return HJD_OK;
}

// Whole batch.  With one range (default) everything goes to the caller's stream; with several, each range
// runs on an internal stream between a fork and a join event on the caller's stream.  Stage events
// (names != nullptr) always use the single-stream form.
static int batch_enqueue(hjd_batch *b, hipStream_t s, bool from_resync, int *nstage, const char **names,
                         hipEvent_t *evs = nullptr)
{
    if (!evs) evs = b->ev;
    if (!b->nimg) {
        if (nstage) *nstage = 0;
        return HJD_OK;
    }
    // a run of this batch whose sequential emulation may still be at work on the side stream owns the batch's slabs until it
    // is through (other batches are not held up)
    if (b->seq_pending) HIPCHK(hipStreamWaitEvent(s, b->seq_done, 0));
    b->P.epoch = ++b->epoch; // what this run's sibling workgroups write into their flags (never 0)
    if (!from_resync) {
        hjd_decoder *d = b->dec;
        std::lock_guard<std::mutex> lk(d->conc_mu);
        d->recent_streams[d->recent_n++ & 7u] = (void *)s;
        const int n = (int)(d->recent_n < 8u ? d->recent_n : 8u);
        int distinct = 0;
        for (int i = 0; i < n; i++) {
            bool seen = false;
            for (int j = 0; j < i; j++) seen |= d->recent_streams[j] == d->recent_streams[i];
            distinct += seen ? 0 : 1;
        }
        b->callers_seen = distinct;
    }
    if (b->epoch == 0xFFFFFFFFu) b->epoch = 0;
    // With three or more caller streams in flight (callers_seen) the batch runs whole, on the caller's stream: the other runs are
    // what this one overlaps with.  Its ranges one behind the other on that stream were measured too — 64 x 4K as four batches of
    // sixteen on four caller streams 328 Gpixel/s, each batch in two ranges of eight 285; config 3 as four batches of 256, each in
    // four ranges, 260 against 295 whole: the limit of two synchronisation kernels at a time hands out its turns in the order of
    // the host's calls, and a stream's four ranges take four turns in a row.  A caller with large batches deals them to its
    // streams in pieces of about 130 Mpixel itself (bench.py --launches-per-stream: config 3 as sixteen batches 310-314).
    const bool whole = names || b->ranges.size() <= 1 || b->callers_seen >= 3;
    if (whole) {
        hjd_batch::Range all{};
        all.i0 = 0;
        all.i1 = b->nimg;
        all.nchunk = b->nchunks;
        all.nsyncwg = b->nsyncwg;
        all.nssyncwg = b->nsyncwg_small;
        all.nseq = (uint32_t)(b->seqlist.size() / 64u);
        for (auto &im : b->img) all.side_feeds_big |= (im.sync_small && !im.write_small) ? 1u : 0u;
        all.nrepwg = b->nrepwg;
        all.nwritewg = b->nwritewg;
        all.nswritewg = b->nwritewg_small;
        all.ndc = b->ndcchunks;
        all.ngroup = b->ngroups;
        all.nblk = b->nblk_entries;
        all.nrst = (uint32_t)b->rst_images.size();
        int rc = range_enqueue(b, all, s, from_resync, nstage, names, evs);
        return rc ? rc : emulation_enqueue(b, s);
    }
    // The ranges are dealt round robin to at most four streams — the caller's and the decoder's three (a process has four hardware
    // queues) — and run one behind the other on each: a batch of 2 Gpixel is sixteen ranges of the size that
    // overlaps best (about 130 Mpixel: the headline's launches), four at a time, like a caller that deals sixteen batches to four
    // streams (BASELINE config 3 as ONE batch on one caller stream: 302 Gpixel/s in sixteen ranges, 274 in three).
    // Ranges are enqueued in order; with HJD_STAGGER range k's synchronisation waits for range k-1's (or for its final pass).
    const int stagger = from_resync ? 0 : b->dec->stagger;
    const size_t nr = b->ranges.size();
    const size_t nq = nr < 4 ? nr : 4;
    HIPCHK(hipEventRecord(b->fork_ev, s));
    // include/hjd.h: work on the library's streams is joined to the caller's before hjd_batch_run returns — also when a launch
    // fails half-way (the streams forked so far are joined on every way out)
    size_t forked = 1; // streams 1 .. forked-1 of this run have been given work
    auto join_forked = [&]() -> int {
        int rc = HJD_OK;
        for (size_t q = 1; q < forked; q++)
            if (hipEventRecord(b->join_ev[q], b->dec->lib_streams[q - 1]) != hipSuccess || hipStreamWaitEvent(s, b->join_ev[q], 0) != hipSuccess)
                rc = HJD_ERR_INTERNAL;
        return rc;
    };
    for (size_t k = 0; k < nr; k++) {
        const size_t q = k % nq;
        hipStream_t sk = q ? b->dec->lib_streams[q - 1] : s;
        if (q && k < nq) { // a stream's first range of this run
            if (hipStreamWaitEvent(sk, b->fork_ev, 0) != hipSuccess) {
                (void)join_forked();
                return HJD_ERR_INTERNAL;
            }
            forked = q + 1;
        }
        int rc = range_enqueue(b, b->ranges[k], sk, from_resync, nullptr, nullptr, nullptr,
                               (stagger && k) ? b->stagger_ev[k - 1] : nullptr, stagger ? b->stagger_ev[k] : nullptr, stagger, k);
        if (rc) {
            (void)join_forked();
            return rc;
        }
    }
    if (join_forked() != HJD_OK) return HJD_ERR_INTERNAL;
    return emulation_enqueue(b, s);
}

extern "C" int hjd_batch_run(hjd_batch *b, void *hip_stream)
{
    if (!b) return HJD_ERR_ARG;
    DeviceGuard guard(b->device); // a non-NULL hip_stream must belong to the decoder's device
    if (!guard.ok) return HJD_ERR_NO_DEVICE;
    if ((int)b->tev_n.size() < b->tev_cap) { // timing enabled and room left: record this run's stage events
        const int r = (int)b->tev_n.size();
        int ns = 0;
        int rc = batch_enqueue(b, (hipStream_t)hip_stream, false, &ns, b->tev_names, b->tev[r].data());
        b->tev_n.push_back(rc == HJD_OK ? ns : 0);
        return rc;
    }
    return batch_enqueue(b, (hipStream_t)hip_stream, false, nullptr, nullptr);
}

extern "C" int hjd_batch_enable_timing(hjd_batch *b, int max_runs)
{
    if (!b || max_runs < 0 || max_runs > 4096) return HJD_ERR_ARG;
    DeviceGuard guard(b->device);
    if (!guard.ok) return HJD_ERR_NO_DEVICE;
    for (auto &v : b->tev)
        for (auto &e : v) (void)hipEventDestroy(e);
    b->tev.clear();
    b->tev_n.clear();
    b->tev_cap = 0;
    b->tev.resize(max_runs);
    for (auto &v : b->tev) {
        v.resize(HJD_MAX_STAGES + 1);
        for (auto &e : v) HIPCHK(hipEventCreate(&e));
    }
    b->tev_cap = max_runs;
    return HJD_OK;
}

extern "C" int hjd_batch_collect_timing(hjd_batch *b, float *stage_ms_avg, const char **stage_names, int cap, int *runs)
{
    if (!b || !stage_ms_avg || !stage_names) return -1;
    DeviceGuard guard(b->device);
    if (!guard.ok) return -1;
    int nout = 0, nr = 0;
    for (size_t r = 0; r < b->tev_n.size(); r++) {
        const int ns = b->tev_n[r];
        if (ns < 2) continue;
        nout = ns - 1 < cap ? ns - 1 : cap;
        for (int k = 0; k < nout; k++) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, b->tev[r][k], b->tev[r][k + 1]) != hipSuccess) return -1;
            if (nr == 0) stage_ms_avg[k] = 0;
            stage_ms_avg[k] += ms;
            stage_names[k] = b->tev_names[k];
        }
        nr++;
    }
    for (int k = 0; k < nout && nr; k++) stage_ms_avg[k] /= (float)nr;
    if (runs) *runs = nr;
    b->tev_n.clear(); // ready to record again
    return nout;
}

extern "C" int hjd_batch_run_timed(hjd_batch *b, void *hip_stream, float *stage_ms, const char **stage_names, int cap)
{
    if (!b || !stage_ms || !stage_names) return -1;
    DeviceGuard guard(b->device);
    if (!guard.ok) return -1;
    hipStream_t s = (hipStream_t)hip_stream;
    if (!b->have_events) {
        for (auto &e : b->ev)
            if (hipEventCreate(&e) != hipSuccess) return -1;
        b->have_events = true;
    }
    const char *names[HJD_MAX_STAGES + 1] = {};
    int ns = 0;
    if (batch_enqueue(b, s, false, &ns, names) != HJD_OK) return -1;
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    int nout = 0;
    for (int k = 0; k + 1 < ns && nout < cap; k++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, b->ev[k], b->ev[k + 1]) != hipSuccess) return -1;
        stage_ms[nout] = ms;
        stage_names[nout] = names[k];
        nout++;
    }
    return nout;
}

extern "C" int hjd_batch_finish(hjd_batch *b, void *hip_stream, int *status)
{
    if (!b) return HJD_ERR_ARG;
    DeviceGuard guard(b->device);
    if (!guard.ok) return HJD_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)hip_stream;
    int overall = HJD_OK;
    bool emulation_ran = false;
    if (b->nimg) {
        // normally one pass; if the induction check failed somewhere, run more synchronisation
        // launches (each propagates true states across one more workgroup boundary) and redo the output
        for (int attempt = 0;; attempt++) {
            HIPCHK(hipStreamSynchronize(s));
            if (b->seq_pending) { // the sequential emulation of the restart streams this run flagged (enqueued by the run itself,
                                  // on the side stream): its verdicts replace the parallel path's for those images
                HIPCHK(hipEventSynchronize(b->seq_done));
                b->seq_pending = false;
                emulation_ran = true;
            }
            HIPCHK(hipMemcpy(b->state.data(), b->d_st, sizeof(HjdImageState) * b->nimg, hipMemcpyDeviceToHost));
            bool stale = false;
            for (auto &st : b->state) stale |= (st.flags & HJD_FLAG_NOT_CONVERGED) != 0;
            if (!stale) {
                // restart streams the parallel path flagged and no emulation has looked at yet: the first run of these contents
                // does not carry the emulation launches (emulation_enqueue) — they follow here, and the later runs enqueue them
                bool need = false;
                if (b->dec->dirty_mode == 0 && !b->emul_enqueued)
                    for (uint32_t q : b->rst_images) need |= (b->state[q].flags & (HJD_FLAG_DIRTY | HJD_FLAG_BAD_MARKER)) && !b->state[q].emulated;
                if (!need) break;
                attempt--; // not a re-synchronisation
                b->emulate_in_run = true;
                int rc = emulation_enqueue(b, s, true);
                if (rc) return rc;
                continue;
            }
            if (b->dec->verbose)
                for (size_t k = 0; k < b->state.size(); k++)
                    if (b->state[k].flags & HJD_FLAG_NOT_CONVERGED)
                        fprintf(stderr, "hjd: attempt %d image %zu flags %#x sync rounds %u (max %u) decodes %u\n", attempt, k, b->state[k].flags,
                                b->state[k].sync_iters, b->state[k].sync_iters_max, b->state[k].sync_decodes);
            if (attempt >= b->dec->max_resync) { // (HJD_MAX_RESYNC: diagnostics)
                overall = HJD_ERR_INTERNAL;
                break;
            }
            // everything from k_repair onwards runs again and raises its flags again (the ones a k_write pass
            // over stale states raised mean nothing); what k_unstuff_write found is not recomputed and must survive
            for (auto &st : b->state) {
                st.flags &= HJD_FLAG_BAD_MARKER;
                st.emulated = 0;
            }
            b->resyncs++;
            // keep ustream_len, nmarkers
            std::vector<HjdImageState> tmp = b->state;
            HIPCHK(hipMemcpyAsync(b->d_st, tmp.data(), sizeof(HjdImageState) * b->nimg, hipMemcpyHostToDevice, s));
            HIPCHK(hipStreamSynchronize(s));
            int rc = batch_enqueue(b, s, true, nullptr, nullptr);
            if (rc) return rc;
        }
    }
    if (emulation_ran)
        for (uint32_t q : b->rst_images) b->dirty_decodes += b->state[q].emulated ? 1 : 0, b->segment_decodes += b->state[q].emulated == 2u ? 1 : 0;
    for (int i = 0; i < b->n; i++) {
        int st = b->parse_status[i];
        const int di = b->dev_index[i];
        if (st == HJD_OK && di >= 0) {
            const uint32_t f = b->state[di].flags;
            if (f & HJD_FLAG_NOT_CONVERGED) st = HJD_ERR_INTERNAL;
            else if (!b->state[di].emulated && (f & (HJD_FLAG_DIRTY | HJD_FLAG_BAD_MARKER))) st = HJD_ERR_DIRTY_RESTART; // dirty_mode = 1
            else if (f & (HJD_FLAG_BAD_CODE | HJD_FLAG_OVERRUN | HJD_FLAG_SHORT)) st = HJD_ERR_CORRUPT;
        }
        if (status) status[i] = st;
    }
    return overall;
}

extern "C" int hjd_batch_image_info(const hjd_batch *b, int i, unsigned *w, unsigned *h, size_t *off, size_t *bytes)
{
    if (!b || i < 0 || i >= b->n) return HJD_ERR_ARG;
    const int di = b->dev_index[i];
    if (di < 0) return b->parse_status[i] ? b->parse_status[i] : HJD_ERR_UNSUPPORTED;
    if (w) *w = b->img[di].width;
    if (h) *h = b->img[di].height;
    if (off) *off = (size_t)b->img[di].out_off;
    if (bytes) *bytes = b->out_bytes[di];
    return HJD_OK;
}

extern "C" void *hjd_batch_out_device_ptr(const hjd_batch *b) { return b ? b->d_out : nullptr; }
extern "C" size_t hjd_batch_out_bytes(const hjd_batch *b) { return b ? b->out_total : 0; }

extern "C" int hjd_batch_download(hjd_batch *b, int i, uint8_t *dst, size_t cap)
{
    if (!b || !dst || i < 0 || i >= b->n) return HJD_ERR_ARG;
    const int di = b->dev_index[i];
    if (di < 0) return b->parse_status[i] ? b->parse_status[i] : HJD_ERR_UNSUPPORTED;
    if (cap < b->out_bytes[di]) return HJD_ERR_ARG;
    DeviceGuard guard(b->device);
    if (!guard.ok) return HJD_ERR_NO_DEVICE;
    HIPCHK(hipMemcpy(dst, b->d_out + b->img[di].out_off, b->out_bytes[di], hipMemcpyDeviceToHost));
    return HJD_OK;
}

extern "C" int hjd_batch_download_coefs(hjd_batch *b, int i, int16_t *dst, size_t elems)
{
    if (!b || !dst || i < 0 || i >= b->n) return HJD_ERR_ARG;
    const int di = b->dev_index[i];
    if (di < 0) return HJD_ERR_UNSUPPORTED;
    const HjdImage &im = b->img[di];
    const size_t ne = (size_t)im.nblocks * 64;
    if (elems < ne) return HJD_ERR_ARG;
    DeviceGuard guard(b->device);
    if (!guard.ok) return HJD_ERR_NO_DEVICE;
    // test hook: rebuild the dense DCT[] rows from the device's sparse form (records + offsets + predicted DC)
    std::vector<uint2> blk(im.nblocks + 1);
    std::vector<uint32_t> rec(im.rec_cap);
    uint8_t slot_to_zz[64];
    for (int k = 0; k < 64; k++) slot_to_zz[kZzToSlot[k]] = (uint8_t)k;
    std::vector<int16_t> dca(im.nblocks);
    HIPCHK(hipMemcpy(blk.data(), b->d_blk + im.blk_off, blk.size() * sizeof(uint2), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(dca.data(), b->d_dca + im.blk_off, dca.size() * sizeof(int16_t), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(rec.data(), b->d_rec + im.rec_off, rec.size() * 4, hipMemcpyDeviceToHost));
    memset(dst, 0, ne * 2);
    for (uint32_t k = 0; k < im.nblocks; k++) {
        dst[(size_t)k * 64] = dca[k];
        for (uint32_t t = blk[k].x; t < blk[k + 1].x && t < im.rec_cap; t++)
            if (rec[t] != HJD_REC_PAD) dst[(size_t)k * 64 + slot_to_zz[(rec[t] >> 16) & 63]] = (int16_t)(rec[t] & 0xFFFF);
    }
    return HJD_OK;
}

extern "C" int hjd_test_color_table(hjd_decoder *d, uint8_t *dst, size_t cap)
{
    const size_t n = (size_t)3 << 24;
    if (!d || !dst || cap < n) return HJD_ERR_ARG;
    DeviceGuard guard(d->device);
    if (!guard.ok) return HJD_ERR_NO_DEVICE;
    uint8_t *dev = nullptr;
    HIPCHK(hipMalloc(&dev, n));
    int rc = hjd_launch_color_table(dev, nullptr) ? HJD_ERR_INTERNAL : HJD_OK;
    if (rc == HJD_OK && hipMemcpy(dst, dev, n, hipMemcpyDeviceToHost) != hipSuccess) rc = HJD_ERR_INTERNAL;
    (void)hipFree(dev);
    return rc;
}

extern "C" int hjd_batch_sync_stats(const hjd_batch *b, int i, uint32_t *stats4)
{
    if (!b || !stats4 || i < 0 || i >= b->n || b->dev_index[i] < 0) return HJD_ERR_ARG;
    const HjdImageState &st = b->state[b->dev_index[i]];
    const HjdImage &im = b->img[b->dev_index[i]];
    stats4[0] = st.sync_iters;
    stats4[1] = st.sync_decodes;
    stats4[2] = st.sync_iters_max;
    stats4[3] = (uint32_t)(((uint64_t)st.ustream_len * 8 + b->P.subseq_bits - 1) / b->P.subseq_bits);
    (void)im;
    return HJD_OK;
}

extern "C" int hjd_debug_sync_profile(hjd_batch *b, unsigned long long *dst, size_t cap_words)
{
    if (!b || !dst || !b->d_dbg) return -1;
    DeviceGuard guard(b->device);
    const size_t n = (size_t)b->nsyncwg * 32;
    if (cap_words < n) return -1;
    if (hipMemcpy(dst, b->d_dbg, n * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)b->nsyncwg;
}

extern "C" int hjd_batch_path_counts(const hjd_batch *b, uint32_t *counts4)
{
    if (!b || !counts4) return HJD_ERR_ARG;
    counts4[0] = counts4[1] = counts4[2] = counts4[3] = 0;
    for (const HjdImage &im : b->img) {
        counts4[im.sync_small == 2u ? 0 : (im.sync_small == 1u ? 1 : 2)]++;
        counts4[3] += im.write_small ? 1u : 0u;
    }
    return HJD_OK;
}

extern "C" int hjd_batch_resync_count(const hjd_batch *b) { return b ? b->resyncs : -1; }
extern "C" int hjd_batch_sequential_count(const hjd_batch *b) { return b ? b->dirty_decodes : -1; }
extern "C" int hjd_batch_segment_count(const hjd_batch *b) { return b ? b->segment_decodes : -1; }

extern "C" int hjd_batch_totals(const hjd_batch *b, uint64_t *pixels, uint64_t *alg, uint64_t *scan)
{
    if (!b) return HJD_ERR_ARG;
    if (pixels) *pixels = b->pixels;
    if (alg) *alg = b->alg_bytes;
    if (scan) *scan = b->scan_bytes;
    return HJD_OK;
}

// ---- host to host, pipelined ---------------------------------------------------------------------
// The file-in / file-out contract of the reference (openjpg.cpp:593-684) for a whole batch: host JPEG bytes in,
// host RGB/BMP bytes out.  The batch is cut into chunks of about 12 MB of JPEG data; chunk k's header walk,
// staging copy and H2D run while chunk k-1's kernels and chunk k-2's D2H are in flight (three streams, a ring
// of three re-usable batch objects: no allocation once the ring has grown to the chunk size).  Outputs are
// copied straight into the caller's buffers when those are pinned (hjd_host_alloc / hipHostMalloc /
// hipHostRegister); pageable destinations go through pinned staging and a host memcpy.

extern "C" void *hjd_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

extern "C" void hjd_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

static bool is_pinned(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError(); // plain malloc memory: not an error for us
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

static int pipe_retire(hjd_decoder *d, hjd_decoder::Slot &sl, uint8_t *const *out, const size_t *out_cap, int *status)
{
    if (!sl.busy) return HJD_OK;
    sl.busy = false;
    HIPCHK(hipEventSynchronize(sl.down_done));
    std::vector<int> st(sl.count, 0);
    const int before = sl.b->resyncs + sl.b->dirty_decodes;
    int rc = hjd_batch_finish(sl.b, d->s_run, st.data());
    if (rc != HJD_OK) return rc;
    // outputs changed after the download was enqueued (host re-synchronisation, sequential emulation of dirty restart streams)
    const bool redo = sl.b->resyncs + sl.b->dirty_decodes != before;
    for (int k = 0; k < sl.count; k++) {
        const int i = sl.first + k;
        if (status) status[i] = st[k];
        if (st[k] != HJD_OK) continue;
        size_t off = 0, nb = 0;
        hjd_batch_image_info(sl.b, k, nullptr, nullptr, &off, &nb);
        if (redo) {
            HIPCHK(hipMemcpy(out[i], sl.b->d_out + off, nb, hipMemcpyDeviceToHost));
        } else if (!sl.direct[k]) {
            memcpy(out[i], sl.h_out + off, nb);
        }
    }
    return HJD_OK;
}

extern "C" int hjd_decode_batch_host(hjd_decoder *d, int n, const uint8_t *const *jpg, const size_t *jpg_len,
                                     int out_format, uint8_t *const *out, const size_t *out_cap, int *status)
{
    if (!d || n < 0 || (n && (!jpg || !jpg_len || !out || !out_cap)) || (out_format != HJD_OUT_RGB && out_format != HJD_OUT_BMP))
        return HJD_ERR_ARG;
    std::lock_guard<std::mutex> lk(d->pipe_mu);
    DeviceGuard guard(d->device);
    if (!guard.ok) return HJD_ERR_NO_DEVICE;
    if (!d->s_up) {
        HIPCHK(hipStreamCreateWithFlags(&d->s_up, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&d->s_run, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&d->s_down, hipStreamNonBlocking));
        for (auto &sl : d->slot) {
            HIPCHK(hipEventCreateWithFlags(&sl.up_done, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&sl.run_done, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&sl.down_done, hipEventDisableTiming));
            sl.b = hjd_batch_create(d);
        }
    }
    const size_t chunk_bytes = 12u << 20;
    int rc = HJD_OK, nslot = 0;
    for (int i0 = 0; i0 < n && rc == HJD_OK;) {
        int i1 = i0;
        size_t acc = 0;
        while (i1 < n && (i1 == i0 || (acc + jpg_len[i1] <= chunk_bytes && i1 - i0 < 64))) acc += jpg_len[i1++];
        hjd_decoder::Slot &sl = d->slot[nslot++ % 3];
        if ((rc = pipe_retire(d, sl, out, out_cap, status)) != HJD_OK) break; // the slot's previous chunk must be out first
        sl.first = i0;
        sl.count = i1 - i0;
        std::vector<int> pst(sl.count, 0);
        rc = hjd_batch_reset(sl.b, sl.count, jpg + i0, jpg_len + i0, out_format, pst.data(), d->s_up);
        if (rc != HJD_OK) break;
        // the caller's output buffers are checked BEFORE anything of this chunk is enqueued on the run stream: an error
        // return must not leave kernels in flight that the next call's hjd_batch_reset would rewrite the inputs of
        sl.direct.assign(sl.count, 0);
        size_t stage_need = 0;
        for (int k = 0; k < sl.count; k++) {
            size_t off = 0, nb = 0;
            if (hjd_batch_image_info(sl.b, k, nullptr, nullptr, &off, &nb) != HJD_OK) continue; // did not parse: status says so
            if (out_cap[i0 + k] < nb || !out[i0 + k]) {
                rc = HJD_ERR_ARG;
                break;
            }
            sl.direct[k] = is_pinned(out[i0 + k]) ? 1 : 0;
            if (!sl.direct[k] && off + nb > stage_need) stage_need = off + nb;
        }
        if (rc != HJD_OK) break;
        HIPCHK(hipEventRecord(sl.up_done, d->s_up));
        HIPCHK(hipStreamWaitEvent(d->s_run, sl.up_done, 0));
        if ((rc = hjd_batch_run(sl.b, d->s_run)) != HJD_OK) break;
        HIPCHK(hipEventRecord(sl.run_done, d->s_run));
        HIPCHK(hipStreamWaitEvent(d->s_down, sl.run_done, 0));
        if (stage_need > sl.h_out_cap) {
            if (sl.h_out) (void)hipHostFree(sl.h_out);
            sl.h_out = nullptr;
            sl.h_out_cap = 0;
            HIPCHK(hipHostMalloc((void **)&sl.h_out, stage_need + stage_need / 16, hipHostMallocDefault));
            sl.h_out_cap = stage_need + stage_need / 16;
        }
        for (int k = 0; k < sl.count; k++) {
            size_t off = 0, nb = 0;
            if (hjd_batch_image_info(sl.b, k, nullptr, nullptr, &off, &nb) != HJD_OK) continue;
            uint8_t *dst = sl.direct[k] ? out[i0 + k] : sl.h_out + off;
            HIPCHK(hipMemcpyAsync(dst, sl.b->d_out + off, nb, hipMemcpyDeviceToHost, d->s_down));
        }
        HIPCHK(hipEventRecord(sl.down_done, d->s_down));
        sl.busy = true;
        i0 = i1;
    }
    for (auto &sl : d->slot) {
        const int r = pipe_retire(d, sl, out, out_cap, status);
        if (rc == HJD_OK) rc = r;
    }
    if (rc != HJD_OK) { // error exit: nothing of this call may still be in flight when the ring is used again
        (void)hipStreamSynchronize(d->s_up);
        (void)hipStreamSynchronize(d->s_run);
        (void)hipStreamSynchronize(d->s_down);
    }
    return rc;
}

// ---- single-image entry points (the reference's own API surface) -------------------------------

// These calls are the reference's whole API (ConvertJpgFile, openjpg.cpp:593-684; JpegDecodeHW, loadjpg.h:180): one image per
// call.  They share ONE decoder and ONE re-usable batch (hjd_batch_create + reset: the HBM slabs and the pinned staging
// buffers stay allocated between calls — round 2 allocated and freed every slab per call), under one lock.
static std::mutex g_mu;
static hjd_decoder *g_dec = nullptr;
static hjd_batch *g_batch = nullptr;
static uint8_t *g_pin = nullptr; // pinned landing buffer for the output of the file-to-file call
static size_t g_pin_cap = 0;

// g_mu held
static hjd_decoder *default_decoder_locked()
{
    if (!g_dec) {
        int dev = 0;
        const char *e = getenv("HJD_DEVICE");
        if (e) dev = atoi(e);
        g_dec = hjd_decoder_create(dev, nullptr);
    }
    return g_dec;
}

// Decodes one parsed image.  dst != nullptr: the output is copied there (dst_cap bytes at least); else it lands in the pinned
// buffer and *pinned_out points at it (valid until the next call; the caller holds no lock then, so only hjd_convert_jpg_file,
// which writes it out before returning under the lock, uses this form through `sink`).
template <class Sink>
static int decode_one_parsed(const hjd_jpeg_data &j, int out_format, uint8_t *dst, size_t dst_cap, Sink sink)
{
    std::lock_guard<std::mutex> lk(g_mu);
    hjd_decoder *d = default_decoder_locked();
    if (!d) return HJD_ERR_NO_DEVICE;
    DeviceGuard guard(d->device);
    if (!guard.ok) return HJD_ERR_NO_DEVICE;
    if (!g_batch && !(g_batch = hjd_batch_create(d))) return HJD_ERR_INTERNAL;
    std::vector<hjd_jpeg_data> parsed(1, j);
    std::vector<int> pst(1, HJD_OK);
    int st = 0;
    int rc = batch_reset_parsed(g_batch, 1, parsed, pst, out_format, &st, nullptr, false);
    if (rc == HJD_OK) rc = st;
    if (rc == HJD_OK) rc = hjd_batch_run(g_batch, nullptr);
    if (rc == HJD_OK) {
        rc = hjd_batch_finish(g_batch, nullptr, &st);
        if (rc == HJD_OK) rc = st;
    }
    if (rc != HJD_OK) return rc;
    size_t off = 0, nb = 0;
    hjd_batch_image_info(g_batch, 0, nullptr, nullptr, &off, &nb);
    if (dst) {
        if (dst_cap < nb) return HJD_ERR_ARG;
        HIPCHK(hipMemcpy(dst, g_batch->d_out + off, nb, hipMemcpyDeviceToHost));
        return HJD_OK;
    }
    if (nb > g_pin_cap) {
        if (g_pin) (void)hipHostFree(g_pin);
        g_pin = nullptr;
        g_pin_cap = 0;
        HIPCHK(hipHostMalloc((void **)&g_pin, nb + nb / 16, hipHostMallocDefault));
        g_pin_cap = nb + nb / 16;
    }
    HIPCHK(hipMemcpy(g_pin, g_batch->d_out + off, nb, hipMemcpyDeviceToHost));
    return sink(g_pin, nb);
}

static int no_sink(const uint8_t *, size_t) { return HJD_ERR_INTERNAL; }

extern "C" int hjd_jpeg_decode_hw(const hjd_jpeg_data *jdata, unsigned int height, unsigned int width,
                                  unsigned char hFactor, unsigned char vFactor, unsigned char *rgb_out)
{
    if (!jdata || !rgb_out || !jdata->stream) return HJD_ERR_ARG;
    hjd_jpeg_data j = *jdata;
    j.height = height; // the reference passes these separately (openjpg.cpp:655)
    j.width = width;
    j.hfactor = hFactor;
    j.vfactor = vFactor;
    if (!width || !height || !(hFactor == 1 || hFactor == 2 || hFactor == 4) ||
        !(vFactor == 1 || vFactor == 2 || vFactor == 4) || hFactor * vFactor > 4)
        return HJD_ERR_UNSUPPORTED; // hFactor = 0 is a SIGFPE in the reference (loadjpg.cpp:1153)
    {   // a caller-filled struct gets the checks hjd_parse_header applies, before anything is built from it
        const int vst = hjd_validate_jpeg_data(&j, 0);
        if (vst != HJD_OK) {
            hjd_set_last_status(vst);
            return vst;
        }
    }
    const int rc = decode_one_parsed(j, HJD_OUT_RGB, rgb_out, hjd_rgb_bytes(width, height), no_sink);
    hjd_set_last_status(rc);
    return rc;
}

extern "C" int hjd_decode_jpg_file_data(const unsigned char *buf, int sizeBuf, unsigned char **rgbpix,
                                        unsigned int *width, unsigned int *height)
{
    if (!buf || sizeBuf <= 0 || !rgbpix || !width || !height) {
        hjd_set_last_status(HJD_ERR_ARG);
        return 0;
    }
    hjd_jpeg_data j;
    int rc = hjd_parse_header(buf, (size_t)sizeBuf, &j);
    unsigned char *pix = nullptr;
    if (rc == HJD_OK) {
        const size_t nb = hjd_rgb_bytes(j.width, j.height);
        pix = (unsigned char *)malloc(nb ? nb : 1);
        rc = pix ? decode_one_parsed(j, HJD_OUT_RGB, pix, nb, no_sink) : HJD_ERR_INTERNAL;
    }
    hjd_set_last_status(rc);
    if (rc != HJD_OK) {
        free(pix);
        return 0;
    }
    *rgbpix = pix;
    *width = j.width;
    *height = j.height;
    return 1;
}

extern "C" int hjd_convert_jpg_file(const char *in, const char *outp)
{
    if (!in || !outp) {
        hjd_set_last_status(HJD_ERR_ARG);
        return 0;
    }
    std::vector<uint8_t> file;
    if (!hjd_read_file(in, file)) { // openjpg.cpp:603-608
        hjd_set_last_status(HJD_ERR_IO);
        fprintf(stderr, "hjd: cannot open jpg file: %s\n", in);
        return 0;
    }
    hjd_jpeg_data j;
    int rc = hjd_parse_header(file.data(), file.size(), &j);
    if (rc == HJD_OK) // the BMP file image lands in the pinned buffer and is written out from there (WriteBMP24, openjpg.cpp:504-570)
        rc = decode_one_parsed(j, HJD_OUT_BMP, nullptr, 0, [&](const uint8_t *p, size_t nb) { return hjd_write_file(outp, p, nb) ? HJD_OK : HJD_ERR_IO; });
    hjd_set_last_status(rc);
    if (rc != HJD_OK) {
        fprintf(stderr, "hjd: %s: %s\n", in, hjd_status_string(rc));
        return 0;
    }
    return 1; // openjpg.cpp:683
}
