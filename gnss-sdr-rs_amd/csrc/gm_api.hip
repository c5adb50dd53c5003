// gm_api.hip — the extern "C" boundary (include/gnss_mi355x.h) over the HIP kernels.
// Host-side scalar pieces of the reference API (code table, Doppler table construction, manager,
// loop-filter constants) are restated here in C++; everything that touches sample data runs on the GPU.
// There is no CPU fallback for the compute entries: without a device they return GM_ERR_NO_DEVICE.
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "gm_internal.h"

using gm::cf;

// The diagnostic gate (gm_internal.h).  The process environment is walked ONCE, and only for the literal GM_DIAGNOSTICS=1;
// the GM_* overrides are looked up (in the same block) only when it is there.
extern char** environ;
namespace gm {
static const char* env_lookup(const char* name) {
    const size_t n = strlen(name);
    for (char** e = environ; e && *e; ++e)
        if (!strncmp(*e, name, n) && (*e)[n] == '=') return *e + n + 1;
    return nullptr;
}
int diag_int(const char* name, int dflt) {
    static const bool on = [] { const char* v = env_lookup("GM_DIAGNOSTICS"); return v && !strcmp(v, "1"); }();
    if (!on) return dflt;
    const char* v = env_lookup(name);
    return (v && *v) ? atoi(v) : dflt;
}
}  // namespace gm

namespace {

thread_local std::string g_last_error;
int g_device = -1;

int set_err(int code, const char* what) {
    g_last_error = what ? what : "";
    return code;
}
int hip_fail(hipError_t e, const char* where) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", where, hipGetErrorString(e));
    g_last_error = buf;
    return GM_ERR_HIP;
}
#define HIPC(expr)                                         \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) return hip_fail(_e, #expr);  \
    } while (0)

int ensure_device(int dev) {
    if (dev < 0) return set_err(GM_ERR_NO_DEVICE, "gm_init() has not selected a HIP device");
    HIPC(hipSetDevice(dev));
    return GM_OK;
}

#define PI_F 3.14159265358979323846f
const float CA_RATE = 1.023e6f, CA_LEN = 1023.0f;

// ---- C/A code: IS-GPS-200 G1/G2 generator == GPS_CA_CODE_32_PRN (gps_ca_constants.rs)
const uint16_t kG2Delay[32] = {5,   6,   7,   8,   17,  18,  139, 140, 141, 251, 252, 254, 255, 256, 257, 258,
                               469, 470, 471, 472, 473, 474, 509, 512, 513, 514, 515, 516, 859, 860, 861, 862};
struct CaTable {
    int8_t rows[32][1023];
    CaTable() {
        uint8_t g1[1023], g2[1023];
        uint16_t r1 = 0x3ff, r2 = 0x3ff;   // bit k-1 = stage k
        for (int i = 0; i < 1023; ++i) {
            g1[i] = (r1 >> 9) & 1;
            g2[i] = (r2 >> 9) & 1;
            const uint16_t f1 = ((r1 >> 2) ^ (r1 >> 9)) & 1;                                     // 3,10
            const uint16_t f2 = ((r2 >> 1) ^ (r2 >> 2) ^ (r2 >> 5) ^ (r2 >> 7) ^ (r2 >> 8) ^ (r2 >> 9)) & 1;  // 2,3,6,8,9,10
            r1 = uint16_t(((r1 << 1) | f1) & 0x3ff);
            r2 = uint16_t(((r2 << 1) | f2) & 0x3ff);
        }
        for (int p = 0; p < 32; ++p)
            for (int i = 0; i < 1023; ++i)
                rows[p][i] = (g1[i] ^ g2[(i + 1023 - kG2Delay[p]) % 1023]) ? 1 : -1;
    }
};
const CaTable& ca_table() {
    static const CaTable t;
    return t;
}

// ---- BeiDou B1I ranging codes (BASELINE configs[3]'s third constellation; the reference has no BeiDou code: SURVEY §8c5).
// BDS-SIS-ICD-B1I: a Gold code of two 11-stage LFSRs, G1(x) = 1 + x + x^7 + x^8 + x^9 + x^10 + x^11 and
// G2(x) = 1 + x + x^2 + x^3 + x^4 + x^5 + x^8 + x^9 + x^11, both started from 01010101010, truncated by one chip to 2046;
// satellite i takes G1's output XOR the modulo-2 sum of two G2 stages (phase assignment table below, PRN 1..37).
const uint8_t kB1iPhase[37][2] = {{1, 3}, {1, 4}, {1, 5}, {1, 6}, {1, 8}, {1, 9}, {1, 10}, {1, 11}, {2, 7}, {3, 4}, {3, 5}, {3, 6},
                                  {3, 8}, {3, 9}, {3, 10}, {3, 11}, {4, 5}, {4, 6}, {4, 8}, {4, 9}, {4, 10}, {4, 11}, {5, 6}, {5, 8},
                                  {5, 9}, {5, 10}, {5, 11}, {6, 8}, {6, 9}, {6, 10}, {6, 11}, {8, 9}, {8, 10}, {8, 11}, {9, 10},
                                  {9, 11}, {10, 11}};
int b1i_code(uint32_t prn, int8_t* out, uint32_t n_chips) {   // n_chips <= 2047 (2046 = the ICD's truncated code)
    if (prn < 1 || prn > 37 || n_chips > 2047) return GM_ERR_OUT_OF_RANGE;
    uint8_t g1[11], g2[11];                      // stage k at index k-1
    for (int k = 0; k < 11; ++k) g1[k] = g2[k] = uint8_t(k & 1);   // 0 1 0 1 0 1 0 1 0 1 0
    const int a = kB1iPhase[prn - 1][0] - 1, b = kB1iPhase[prn - 1][1] - 1;
    for (uint32_t i = 0; i < n_chips; ++i) {
        const uint8_t chip = g1[10] ^ g2[a] ^ g2[b];
        out[i] = chip ? -1 : 1;                  // logic 0 -> +1, logic 1 -> -1
        const uint8_t f1 = g1[0] ^ g1[6] ^ g1[7] ^ g1[8] ^ g1[9] ^ g1[10];
        const uint8_t f2 = g2[0] ^ g2[1] ^ g2[2] ^ g2[3] ^ g2[4] ^ g2[7] ^ g2[8] ^ g2[10];
        for (int k = 10; k > 0; --k) { g1[k] = g1[k - 1]; g2[k] = g2[k - 1]; }
        g1[0] = f1; g2[0] = f2;
    }
    return GM_OK;
}

size_t num_samples_per_code(float code_rate, float fs, float len) {   // ca_code.rs:13-16
    const float v = roundf(fs / (code_rate / len));
    return v > 0.0f ? size_t(v) : 0;
}

// resample a +-1 chip sequence: idx = floor((i as f32 * code_rate) / fs)  (ca_code.rs:17-22)
int resample_code(const int8_t* chips, size_t code_len, bool wrap, float code_rate, float fs, size_t n, int8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        const float f = floorf((float(i) * code_rate) / fs);
        size_t ind = f > 0.0f ? size_t(f) : 0;
        if (ind >= code_len) {
            if (!wrap) return GM_ERR_OUT_OF_RANGE;   // ca_code[ind] panics in the reference
            ind %= code_len;
        }
        out[i] = chips[ind];
    }
    return GM_OK;
}

void loop_filter_new(float bw, float zeta, float gain, float* tau1, float* tau2) {   // do_tracking.rs:59-65
    const float w = bw * 8.0f * zeta / (4.0f * (zeta * zeta) + 1.0f);
    *tau1 = gain / (w * w);
    *tau2 = (2.0f * zeta) / w;
}

// HIP-event timing of the acquisition kernels on the handle's stream: a pool of event triples, one per
// gm_acq_search_dev call since timing was (re-)enabled, averaged by gm_acq_timing_summary.
struct Timing {
    static constexpr int CAP = 512;
    std::vector<hipEvent_t> ev;   // CAP * 4 events, created on first enable
    bool on = false;
    int stride = 1;               // record every stride-th search (an event record costs ~2 us of stream time)
    int calls = 0;                // searches since enable
    bool this_call = false;       // the search in flight is a recorded one (its decision event follows)
    int count = 0;                // calls recorded since enable (only the last CAP are kept)
    bool decide_valid = false;
};

}  // namespace

// device mirror of MulticastRingBuffer (functions further down)
struct gm_ring {
    int device = -1;
    cf* d_buf = nullptr;
    size_t size = 0, mask = 0;
    // single writer / many readers like the reference ring: the writer publishes `head` (release) after its
    // synchronous H2D copy, readers load it (acquire) before launching kernels on the mirror
    std::atomic<uint64_t> head{0};
    // `notifier` + `condvar` of the reference ring (multicast_ring_buffer.rs:42-43,95-98): gm_ring_wait_head sleeps here
    std::mutex notifier;
    std::condition_variable condvar;
    // asynchronous writer (gm_ring_write_samples_async): pinned staging slots, a copy stream, and `head` published once the copy
    // has landed.  Published by PULL: every chunk leaves an entry {event, head after it} in `pending`; whoever looks at the head
    // (gm_ring_get_head / _wait_head / the stage entries that snapshot the ring) first retires the entries whose events have
    // completed (ring_refresh_head).  Round 4 published from a hipLaunchHostFunc callback on the copy stream: the runtime's
    // callback thread wakes up milliseconds late now and then, and the stream — hence the next block's copy and front-end kernel
    // — waits behind the callback (7-9 ms stalls every five or six blocks of the receiver loop, tools/trk_async_probe.py).
    static constexpr int SLOTS = 4;
    static constexpr size_t SLOT_SAMPLES_MAX = size_t(1) << 19;      // 32 ms at 16.4 Msps: one copy + one front-end launch per block of that size (the
                                                                     // front-end kernel's fixed ~45 us weigh 15 % on a 2^18 block, 8 % on this one)
    size_t slot_samples = 0;                                        // min(ring size, SLOT_SAMPLES_MAX), set at create
    cf* staging[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t slot_done[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    bool slot_used[SLOTS] = {false, false, false, false};
    hipStream_t copy_stream = nullptr;
    // gm_frontend_write_ring: the front-end kernels run on a stream of their own behind the copies (event per slot), so the copy
    // of block k + 1 travels while the kernel of block k runs
    hipStream_t fe_stream = nullptr;
    hipEvent_t h2d_done[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t last_enq_stream = nullptr;       // the stream ev_enq was last recorded on (see ring_enqueue_publish)
    uint64_t write_pos = 0;       // writer-private: samples enqueued so far (>= head)
    uint64_t slot_seq = 0;
    struct Pending { int slot; uint64_t new_head; };
    std::mutex pend_mu;
    std::vector<Pending> pending;           // FIFO, at most SLOTS entries (a slot is re-used only after its entry has been retired)
    // what the writer has ENQUEUED on copy_stream so far, and an event recorded there behind it: a consumer that must not wait on
    // the host (gm_trk_update_all_async) orders its own stream behind `ev_enq` on the DEVICE and may then use `enqueued` as its head
    // — the device-side counterpart of the Condvar wait of do_tracking::run (do_tracking.rs:392-406).  enq_mu: writer and
    // consumer may be different host threads; the pair (enqueued, ev_enq) is read and written under it.
    std::mutex enq_mu;
    hipEvent_t ev_enq = nullptr;
    uint64_t enqueued = 0;
    bool ev_enq_armed = false;
};

// Wait for an event the way a latency path wants it: poll (hipEventQuery) for up to ~2 ms before handing over to the runtime's blocking
// wait.  hipEventSynchronize's wake-up through the interrupt path costs milliseconds on this platform when the event is already (or
// almost) complete — measured 7-9 ms per wait in the receiver loop, against the ~0.3 ms the awaited front-end block takes
// (tools/trk_async_probe.py) — and the device idles meanwhile, because the host is what feeds it.
static hipError_t wait_event_polling(hipEvent_t ev) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipEventQuery(ev);
        if (q != hipErrorNotReady) return q;
        (void)hipGetLastError();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) return hipEventSynchronize(ev);
    }
}

// head := max(head, new_head), never backwards: readers and the writer may retire chunks concurrently (head by pull), and a
// plain store after a stale load would let the older of two retirements land last.  compare-exchange max loop, then the wake-up.
static void ring_publish(gm_ring* r, uint64_t new_head) {
    uint64_t cur = r->head.load(std::memory_order_relaxed);
    bool moved = false;
    while (int64_t(new_head - cur) > 0) {
        if (r->head.compare_exchange_weak(cur, new_head, std::memory_order_release, std::memory_order_relaxed)) { moved = true; break; }
    }
    if (!moved) return;
    std::lock_guard<std::mutex> g(r->notifier);     // "Wake up Tracking after writing new samples" (:94-98)
    r->condvar.notify_all();
}
// retire the chunks whose copies (and front-end kernels) have run: head := the newest completed one.  Any host thread; the
// ring's device must be current.  `all`: the copy stream has just been synchronised — everything is complete.
static void ring_refresh_head(gm_ring* r, bool all = false) {
    if (!r->copy_stream) return;
    uint64_t newest = 0;
    bool any = false;
    {
        std::lock_guard<std::mutex> g(r->pend_mu);
        size_t done = 0;
        while (done < r->pending.size()) {
            if (!all) {
                const hipError_t q = hipEventQuery(r->slot_done[r->pending[done].slot]);
                if (q != hipSuccess) { (void)hipGetLastError(); break; }
            }
            newest = r->pending[done].new_head; any = true; ++done;
        }
        if (done) r->pending.erase(r->pending.begin(), r->pending.begin() + done);
    }
    if (any) ring_publish(r, newest);               // monotonic by construction (two readers retiring at once: ADVICE round 5)
}
static int ring_async_init(gm_ring* r) {
    if (r->copy_stream) return GM_OK;
    {   // The writer's copies at the device's LOWEST stream priority, the front-end kernels (gm_frontend_write_ring) at the HIGHEST: HIP deals
        // streams of one priority to that priority's hardware queues in creation order, and the normal-priority ones are where every other
        // stream of the process lives (a first torch.cuda.Stream() creates 32 of them).  With both at the default priority the receiver
        // chain ran at 218 x real time in a clean process and at 117 x behind four torch streams; low / high: 218 / 216 / 212 x with 0 / 4 / 16
        // of them (tools/rx_order_probe.py, DESIGN_HISTORY R6.12).  GM_RING_COPY_PRIORITY / GM_RING_FE_PRIORITY: diagnostic overrides.
        int least = 0, greatest = 0;
        HIPC(hipDeviceGetStreamPriorityRange(&least, &greatest));
        const int pr = gm::diag_int("GM_RING_COPY_PRIORITY", 99);
        HIPC(hipStreamCreateWithPriority(&r->copy_stream, hipStreamNonBlocking, pr == 99 ? least : pr));
    }
    for (int i = 0; i < gm_ring::SLOTS; ++i) {
        HIPC(hipHostMalloc(reinterpret_cast<void**>(&r->staging[i]), r->slot_samples * 8, hipHostMallocDefault));
        HIPC(hipEventCreateWithFlags(&r->slot_done[i], hipEventDisableTiming));
    }
    r->write_pos = r->head.load(std::memory_order_relaxed);
    return GM_OK;
}
// after the work of one chunk (staging slot `slot`, its event recorded) has been enqueued on `st` (copy_stream or fe_stream)
static int ring_enqueue_publish(gm_ring* r, int slot, hipStream_t st) {
    {
        std::lock_guard<std::mutex> g(r->pend_mu);
        r->pending.push_back(gm_ring::Pending{slot, r->write_pos});
    }
    {   // the enqueued head and its event (see struct gm_ring): "everything enqueued so far" — when the writer changes streams
        // (plain copies and front-end blocks on one ring), the new stream first waits for what the event covered before
        std::lock_guard<std::mutex> g(r->enq_mu);
        if (!r->ev_enq) HIPC(hipEventCreateWithFlags(&r->ev_enq, hipEventDisableTiming));
        if (r->ev_enq_armed && r->last_enq_stream != st) HIPC(hipStreamWaitEvent(st, r->ev_enq, 0));
        r->last_enq_stream = st;
        HIPC(hipEventRecord(r->ev_enq, st));
        r->enqueued = r->write_pos;
        r->ev_enq_armed = true;
    }
    return GM_OK;
}
// a staging slot is about to be re-used: its previous chunk must have run — and is retired first (its entry names the slot's event)
static int ring_reclaim_slot(gm_ring* r, int slot) {
    if (!r->slot_used[slot]) return GM_OK;
    HIPC(wait_event_polling(r->slot_done[slot]));
    ring_refresh_head(r);
    return GM_OK;
}

// ====================================================================== acquisition handle
struct gm_acq {
    int device = -1;
    gm_acq_cfg cfg{};
    const gm::PlanOps* plan = nullptr;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t N = 0, D = 0, M = 0, P = 0;
    float code_rate = CA_RATE;
    std::vector<float> table_freq;
    std::vector<uint8_t> prn_ids, dev_prn_ids;
    cf* d_tw_mix = nullptr;       // forward base twiddles of stage F's own plan (PlanOps::fill_tw_mix)
    cf *d_tables = nullptr, *d_tw_fwd = nullptr, *d_tw_inv = nullptr, *d_code_fft = nullptr, *d_spectra = nullptr;
    cf* d_code_fft_paired = nullptr;   // Q == 1: the code spectra in the layout acq_corr_kernel reads (PairLayout)
    uint16_t* d_order = nullptr;       // Q == 1, permuted storage orders: element index stored at each position (PlanOps::fill_order)
    float* d_table_freq = nullptr;
    int8_t* d_code_samples = nullptr;
    void* d_samples = nullptr;
    size_t samples_cap = 0;
    uint32_t* d_metrics = nullptr;         // [3][P][D] words
    uint32_t* d_worker_list = nullptr;
    std::vector<uint32_t> worker_list;
    uint64_t mask = ~0ull;
    uint32_t n_workers = 0;
    gm_acq_result* d_results = nullptr;
    uint8_t *d_found = nullptr, *d_prn_ids = nullptr;
    uint32_t results_cap = 0;
    const void* last_metrics = nullptr;
    Timing tm;
    // composite transform size N = Q * plan->n (acq_composite.hip); Q == 1: the fused single-LDS-buffer kernels
    float* d_split_scratch = nullptr;      // partial power planes of the correlation grid's tail split
    int split_planes = 0;                  // ... how many of them the scratch holds (sized from the handle's geometry at create)
    uint32_t* d_split_counter = nullptr;   // arrival tickets, one per split item, zero between launches
    uint32_t Q = 1, Nb = 0;
    const gm::CompOps* comp = nullptr;     // Q > 1: the (Q, base plan) kernels
    cf* d_comp_tmp = nullptr;              // [max(D*M, P)][Q][Nb]: forward sub-transforms before the Q-point DFTs
    cf* d_comp_twn = nullptr;              // [Q][Nb] inverse twiddles W_N^{-n1 k2}, paired positions
    float* d_planes = nullptr;             // composite sizes with strict_sum_order: the accumulated power planes [P * D][N] (launch_plane_strict_sum)
    cf* d_code_comb = nullptr;             // Q > 1: [P][Q][Q][Nb] conj(code) x W_Q^{-n1 k1} x W_N^{-n1 k2}, what comp corr multiplies the spectra by
    // gm_acq_decide_dev on the metrics of the last search, on a handle of an in-LDS size: the decision is NOT launched but kept here
    // and rides along with the next search's stage F (PlanOps::mix_fft: trailing workgroups) — anything else that consumes or
    // invalidates it (fetch, synchronize, another decide, a stream change, timing) launches decide_kernel first (acq_flush_decision)
    // gm_acq_prepare_dev: stage F of the NEXT dwell on a stream of the handle's own into a second spectrum buffer.  d_spectra is
    // the buffer the next in-stream stage F writes / the last stage C read; d_spectra_alt the other one, whose last reader is an
    // EARLIER stage C than the most recent one.  ev_s: recorded on the handle's stream just in front of every stage C launch — a
    // preparation that waits for it starts together with the most recent stage C (every earlier one has ended by then) and, at the
    // lowest priority, gets the CUs that stage C no longer needs
    // A preparation is named by a TOKEN (a generation number, never 0), not by the address of the samples it was made from: a
    // ring-backed receiver reuses addresses by construction, and a search that matched on the pointer would silently take the OLD
    // samples' spectra.  Only gm_acq_search_prepared_dev(token) consumes it.
    struct Ahead {
        cf* d_spectra_alt = nullptr;
        hipStream_t side = nullptr;
        hipEvent_t ev_s = nullptr, ev_f = nullptr, ev_in = nullptr;   // ev_in: the caller's `ready_stream` at the time of the call
        bool valid = false;                 // d_spectra_alt holds (or will hold: ev_f) the spectra of the preparation `token`
        uint64_t token = 0, next_token = 1;
        const void* samples = nullptr;      // composite sizes only (nothing is prepared: the search runs from these at search_prepared)
        int fmt = 0;
        bool comp_wait = false;             // composite sizes: ev_in was recorded on the caller's ready_stream, the search waits for it
    } ahead;
    bool defer_decisions = false;   // gm_acq_set_deferred_decision
    bool dec_deferred = false;
    gm::DecideArgs dec_args{};
    bool comp_post_folded = false;         // ... and, for base plans with an order table, the signal's forward step 2 (CompOps::fold_post): d_spectra = the sub-transforms
    // fine Doppler (gm_acq_finer_doppler): host copy of the chip rows, lazily built device state
    std::vector<int8_t> chips;             // [P][code_len]
    uint32_t code_len = 1023;
    const void* last_samples = nullptr;    // device snapshot of the last search
    int last_fmt = GM_FMT_C32;
    struct Fine {
        const gm::PlanOps *p1 = nullptr, *p2 = nullptr;
        uint32_t size_use = 0, sats_cap = 0;
        int8_t* d_chips = nullptr;
        cf *d_tw1 = nullptr, *d_tw2 = nullptr, *d_B = nullptr;
        float *d_mean = nullptr, *d_rowmax = nullptr, *d_peak_pow = nullptr;
        uint32_t *d_rowarg = nullptr, *d_sat_worker = nullptr, *d_sat_cp = nullptr, *d_peak_idx = nullptr;
    } fine;
};

static int acq_set_mask(gm_acq* a, uint64_t mask) {
    std::vector<uint32_t> wl;
    for (uint32_t i = 0; i < a->P; ++i)
        if (i >= 64 || ((mask >> i) & 1ull)) wl.push_back(i);
    a->mask = mask;
    if (wl != a->worker_list || a->n_workers != wl.size()) {
        a->worker_list = wl;
        a->n_workers = uint32_t(wl.size());
        if (!wl.empty())
            HIPC(hipMemcpyAsync(a->d_worker_list, a->worker_list.data(), wl.size() * sizeof(uint32_t),
                                hipMemcpyHostToDevice, a->stream));
        HIPC(hipStreamSynchronize(a->stream));
    }
    return GM_OK;
}

static int acq_flush_decision(gm_acq* a) {
    if (a->dec_deferred) {
        a->dec_deferred = false;
        gm::launch_decide(a->stream, a->dec_args);
        HIPC(hipGetLastError());
    }
    return GM_OK;
}

static int acq_reserve_results(gm_acq* a, uint32_t n) {
    if (n <= a->results_cap) return GM_OK;
    if (int rc = acq_flush_decision(a)) return rc;
    // results + found flags live in ONE host-pinned, device-visible block: decide_kernel writes its P x 41 bytes straight into
    // host memory and gm_acq_fetch_results is a stream synchronisation and a memcpy — the two device-to-host copies it used to
    // issue cost the host-buffer entry (gm_acq_search) ~25 us per dwell
    // (the new blocks are allocated BEFORE the old ones go: a failed allocation leaves the handle as it was)
    void* blk = nullptr;
    uint8_t* ids = nullptr;
    HIPC(hipHostMalloc(&blk, (sizeof(gm_acq_result) + 1) * size_t(n), hipHostMallocDefault));
    if (hipError_t e = hipMalloc(&ids, n); e != hipSuccess) { hipHostFree(blk); return hip_fail(e, "hipMalloc(prn ids)"); }
    if (a->d_results) {
        if (hipError_t e = hipStreamSynchronize(a->stream); e != hipSuccess) { hipHostFree(blk); hipFree(ids); return hip_fail(e, "hipStreamSynchronize"); }
        hipHostFree(a->d_results);
        hipFree(a->d_prn_ids);
    }
    a->dev_prn_ids.clear();
    a->d_results = static_cast<gm_acq_result*>(blk);
    a->d_found = reinterpret_cast<uint8_t*>(a->d_results + n);
    a->d_prn_ids = ids;
    a->results_cap = n;
    return GM_OK;
}

extern "C" {

int gm_abi_version(void) { return GM_ABI_VERSION; }

int gm_device_count(int* count) {
    if (!count) return set_err(GM_ERR_INVALID_ARG, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return hip_fail(e, "hipGetDeviceCount"); }
    *count = n;
    return GM_OK;
}

int gm_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return set_err(GM_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return set_err(GM_ERR_INVALID_ARG, "device index out of range");
    HIPC(hipSetDevice(device));
    g_device = device;
    return GM_OK;
}

const char* gm_last_error(void) { return g_last_error.c_str(); }

const char* gm_status_string(int s) {
    switch (s) {
        case GM_OK: return "ok";
        case GM_ERR_INVALID_ARG: return "invalid argument";
        case GM_ERR_UNSUPPORTED_N: return "no FFT plan for this fft_size";
        case GM_ERR_NO_DEVICE: return "no HIP device";
        case GM_ERR_HIP: return "HIP runtime error";
        case GM_ERR_OUT_OF_RANGE: return "index out of range";
        case GM_ERR_ALIGNMENT: return "fft_size must be a multiple of 8";
        case GM_ERR_NOMEM: return "out of memory";
        default: return "unknown status";
    }
}

// ---------------------------------------------------------------- code table / Doppler table (host)
int gm_b1i_code(uint32_t prn, int8_t* out_chips, uint32_t n_chips) {
    if (!out_chips) return set_err(GM_ERR_INVALID_ARG, "null out");
    if (int rc = b1i_code(prn, out_chips, n_chips)) return set_err(rc, "B1I PRN 1..37, at most 2047 chips");
    return GM_OK;
}

int gm_ca_code_row(int row, int8_t out[1023]) {
    if (row < 0 || row > 31 || !out) return set_err(GM_ERR_OUT_OF_RANGE, "row must be 0..31");
    memcpy(out, ca_table().rows[row], 1023);
    return GM_OK;
}

int gm_generate_ca_code_samples(uint8_t prn, float code_rate, float fs, int8_t* out, size_t cap, size_t* n_out) {
    if (prn < 1 || prn > 32) return set_err(GM_ERR_OUT_OF_RANGE, "prn must be 1..=32");
    const size_t n = num_samples_per_code(code_rate, fs, CA_LEN);
    if (n_out) *n_out = n;
    std::vector<int8_t> tmp(n);
    const int rc = resample_code(ca_table().rows[prn - 1], 1023, false, code_rate, fs, n, tmp.data());
    if (rc) return set_err(rc, "chip index reached 1023 (the reference panics)");
    if (out) memcpy(out, tmp.data(), n < cap ? n : cap);
    return GM_OK;
}

int gm_doppler_table_new(float f_if, float doppler_hz, float fs, size_t n, float* freq_out, gm_c32* table) {
    if (!table && n) return set_err(GM_ERR_INVALID_ARG, "table_out is null");
    const float carr_freq = f_if + doppler_hz;               // doppler_shift.rs:13
    const float phase_step = 2.0f * PI_F * carr_freq / fs;   // :14
    for (size_t i = 0; i < n; ++i) {
        const float phase = float(i) * phase_step;           // :17
        table[i].re = cosf(phase);                           // :18
        table[i].im = -sinf(phase);
    }
    if (freq_out) *freq_out = carr_freq;                     // :20
    return GM_OK;
}

int gm_apply_doppler_shift(const gm_c32* samples, const gm_c32* table, gm_c32* output, size_t n) {
    if (!samples || !table || !output) return set_err(GM_ERR_INVALID_ARG, "null pointer");
    if (int rc = ensure_device(g_device)) return rc;
    const size_t n4 = (n / 4) * 4;
    if (!n4) return GM_OK;
    cf *ds = nullptr, *dt = nullptr, *dout = nullptr;
    HIPC(hipMalloc(&ds, n4 * 8)); HIPC(hipMalloc(&dt, n4 * 8)); HIPC(hipMalloc(&dout, n4 * 8));
    HIPC(hipMemcpy(ds, samples, n4 * 8, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(dt, table, n4 * 8, hipMemcpyHostToDevice));
    gm::launch_apply_doppler(nullptr, ds, dt, dout, n);
    HIPC(hipGetLastError());
    HIPC(hipMemcpy(output, dout, n4 * 8, hipMemcpyDeviceToHost));   // tail n%4 untouched, like the reference
    hipFree(ds); hipFree(dt); hipFree(dout);
    return GM_OK;
}

// ---------------------------------------------------------------- FFT<T> / RealFFT<T>
int gm_fft_supported_sizes(uint32_t* sizes, int cap) { return gm::list_plans(sizes, cap); }

// power-of-two lengths above one LDS buffer: L = N1 * N2, both in-LDS power-of-two plans (N1 >= N2, as balanced as they come)
static bool pow2_split(size_t n, const gm::PlanOps** p1, const gm::PlanOps** p2) {
    if (n < 2 || (n & (n - 1)) || n > (size_t(1) << 28)) return false;
    for (size_t n2 = 256; n2 * n2 <= n || n2 <= 16384; n2 *= 2) {
        if (n % n2) continue;
        const size_t n1 = n / n2;
        if (n1 < n2) break;
        const gm::PlanOps *a = n1 <= 16384 ? gm::find_plan(int(n1)) : nullptr, *b = gm::find_plan(int(n2));
        if (a && b && a->big_cols && b->big_rows) { *p1 = a; *p2 = b; return true; }
    }
    return false;
}
static int fft_run_big(size_t n, int dir, cf* d_data, size_t batch, hipStream_t st) {
    const gm::PlanOps *p1 = nullptr, *p2 = nullptr;
    if (!pow2_split(n, &p1, &p2)) return set_err(GM_ERR_UNSUPPORTED_N, "no in-LDS FFT plan for this length");
    std::vector<cf> t1(size_t(p1->tw_total) + 1), t2(size_t(p2->tw_total) + 1);
    p1->fill_tw(t1.data(), dir != 0);
    p2->fill_tw(t2.data(), dir != 0);
    cf *d_t1 = nullptr, *d_t2 = nullptr, *d_b = nullptr;
    auto done = [&](int code) { hipFree(d_t1); hipFree(d_t2); hipFree(d_b); return code; };
    if (hipMalloc(&d_t1, t1.size() * 8) != hipSuccess || hipMalloc(&d_t2, t2.size() * 8) != hipSuccess || hipMalloc(&d_b, n * 8) != hipSuccess)
        return done(set_err(GM_ERR_NOMEM, "hipMalloc (long FFT)"));
    if (hipMemcpy(d_t1, t1.data(), t1.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_t2, t2.data(), t2.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return done(set_err(GM_ERR_HIP, "hipMemcpy"));
    for (size_t it = 0; it < batch; ++it) {
        cf* x = d_data + it * n;
        p1->big_cols(st, x, d_b, d_t1, uint32_t(p2->n), dir != 0);
        p2->big_rows(st, d_b, x, d_t2, uint32_t(p1->n), dir != 0);
    }
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return done(set_err(GM_ERR_HIP, "long FFT kernels"));
    return done(GM_OK);
}

static int fft_run(size_t n, int dir, cf* d_data, size_t batch, hipStream_t st) {
    const gm::PlanOps* pl = gm::find_plan(int(n));
    if (!pl) return fft_run_big(n, dir, d_data, batch, st);
    std::vector<cf> tw(size_t(pl->tw_total) + 1);
    pl->fill_tw(tw.data(), dir != 0);
    cf* d_tw = nullptr;
    HIPC(hipMalloc(&d_tw, tw.size() * sizeof(cf)));
    HIPC(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(cf), hipMemcpyHostToDevice));
    pl->fft_batch(st, d_data, d_tw, dir != 0, int(batch));
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(st));
    hipFree(d_tw);
    return GM_OK;
}

// FFT<T>::new is generic over the length (src/fft.rs:10-19: rustfft plans any N).  Lengths without an in-LDS plan go
// through Bluestein's chirp-z identity on the smallest power-of-two plan L >= 2N - 1:
//   X[k] = c[k] * sum_n (x[n] c[n]) conj(c[k - n]),   c[n] = exp(-j pi n^2 / N)   (phase from n^2 mod 2N: exact in integers)
// i.e. one length-L circular convolution = two forward transforms (the chirp's is cached per N) and one inverse on the
// device; the three O(N) chirp products are host loops (this entry takes and returns host buffers).  L up to 2^24 (N <= 2^23):
// powers of two above one LDS buffer run as a four-step transform (fft_run_big).
static int fft_bluestein(size_t n, int dir, gm_c32* inout, size_t batch) {
    size_t L = 256;
    while (L < 2 * n - 1) L *= 2;
    {   // the next power of two that has a transform (32768 = 256 x 128 has no two in-LDS factors: 65536 then)
        const gm::PlanOps *p1 = nullptr, *p2 = nullptr;
        while (L <= (size_t(1) << 24) && !gm::find_plan(int(L)) && !pow2_split(L, &p1, &p2)) L *= 2;
        if (L > (size_t(1) << 24))
            return set_err(GM_ERR_UNSUPPORTED_N, "length above 2^23: no FFT plan (Bluestein needs a power of two >= 2N - 1, at most 2^24)");
    }
    std::vector<cf> c(n), b(L, gm::cf_make(0.f, 0.f));
    for (size_t i = 0; i < n; ++i) {
        const double a = M_PI * double((uint64_t(i) * i) % (2 * n)) / double(n);
        c[i] = gm::cf_make(float(cos(a)), float(-sin(a)));
    }
    b[0] = gm::cf_make(1.f, 0.f);
    for (size_t i = 1; i < n; ++i) b[i] = b[L - i] = gm::cf_make(c[i].x, -c[i].y);
    cf *d_b = nullptr, *d_a = nullptr, *d_y = nullptr;
    int rc = GM_OK;
    auto done = [&](int code) { hipFree(d_b); hipFree(d_a); hipFree(d_y); return code; };
    if (hipMalloc(&d_b, L * 8) != hipSuccess || hipMalloc(&d_a, L * 8) != hipSuccess || hipMalloc(&d_y, L * 8) != hipSuccess)
        return done(set_err(GM_ERR_HIP, "hipMalloc"));
    if (hipMemcpy(d_b, b.data(), L * 8, hipMemcpyHostToDevice) != hipSuccess) return done(set_err(GM_ERR_HIP, "hipMemcpy"));
    if ((rc = fft_run(L, 0, d_b, 1, nullptr))) return done(rc);
    std::vector<cf> a(L), y(L);
    const float inv_l = 1.0f / float(L);
    for (size_t it = 0; it < batch; ++it) {
        gm_c32* x = inout + it * n;
        for (size_t i = 0; i < L; ++i) a[i] = gm::cf_make(0.f, 0.f);
        for (size_t i = 0; i < n; ++i) {       // inverse transform: conj(FFT(conj(x)))
            const cf xi = gm::cf_make(x[i].re, dir ? -x[i].im : x[i].im);
            a[i] = gm::cf_make(xi.x * c[i].x - xi.y * c[i].y, xi.x * c[i].y + xi.y * c[i].x);
        }
        if (hipMemcpy(d_a, a.data(), L * 8, hipMemcpyHostToDevice) != hipSuccess) return done(set_err(GM_ERR_HIP, "hipMemcpy"));
        if ((rc = fft_run(L, 0, d_a, 1, nullptr))) return done(rc);
        gm::launch_apply_doppler(nullptr, d_a, d_b, d_y, L);                    // elementwise complex product (L % 4 == 0)
        if (hipGetLastError() != hipSuccess) return done(set_err(GM_ERR_HIP, "product kernel"));
        if ((rc = fft_run(L, 1, d_y, 1, nullptr))) return done(rc);
        if (hipMemcpy(y.data(), d_y, L * 8, hipMemcpyDeviceToHost) != hipSuccess) return done(set_err(GM_ERR_HIP, "hipMemcpy"));
        for (size_t k = 0; k < n; ++k) {
            const float yr = y[k].x * inv_l, yi = y[k].y * inv_l;
            const float re = yr * c[k].x - yi * c[k].y, im = yr * c[k].y + yi * c[k].x;
            x[k].re = re; x[k].im = dir ? -im : im;
        }
    }
    return done(GM_OK);
}

int gm_fft_c2c_f32(size_t n, int dir, gm_c32* inout, size_t batch) {
    if (!inout || !n || !batch) return set_err(GM_ERR_INVALID_ARG, "null or empty");
    if (int rc = ensure_device(g_device)) return rc;
    {
        const gm::PlanOps *p1 = nullptr, *p2 = nullptr;
        if (!gm::find_plan(int(n)) && !pow2_split(n, &p1, &p2)) return fft_bluestein(n, dir, inout, batch);
    }
    cf* d = nullptr;
    HIPC(hipMalloc(&d, n * batch * 8));
    HIPC(hipMemcpy(d, inout, n * batch * 8, hipMemcpyHostToDevice));
    const int rc = fft_run(n, dir, d, batch, nullptr);
    if (!rc) HIPC(hipMemcpy(inout, d, n * batch * 8, hipMemcpyDeviceToHost));
    hipFree(d);
    return rc;
}

int gm_fft_power_spectrum_f32(size_t n, gm_c32* inout, float* power) {
    if (!inout || !power || !n) return set_err(GM_ERR_INVALID_ARG, "null or empty");
    if (int rc = ensure_device(g_device)) return rc;
    const gm::PlanOps *pp1 = nullptr, *pp2 = nullptr;
    if (!gm::find_plan(int(n)) && !pow2_split(n, &pp1, &pp2)) {            // any other length: Bluestein, then |X|^2 on the host
        if (int rc = fft_bluestein(n, 0, inout, 1)) return rc;
        for (size_t i = 0; i < n; ++i) power[i] = inout[i].re * inout[i].re + inout[i].im * inout[i].im;   // norm_sqr (fft.rs:28)
        return GM_OK;
    }
    cf* d = nullptr; float* dp = nullptr;
    HIPC(hipMalloc(&d, n * 8)); HIPC(hipMalloc(&dp, n * 4));
    HIPC(hipMemcpy(d, inout, n * 8, hipMemcpyHostToDevice));
    int rc = fft_run(n, 0, d, 1, nullptr);
    if (!rc) {
        gm::launch_power(nullptr, d, dp, n);
        HIPC(hipGetLastError());
        HIPC(hipMemcpy(inout, d, n * 8, hipMemcpyDeviceToHost));
        HIPC(hipMemcpy(power, dp, n * 4, hipMemcpyDeviceToHost));
    }
    hipFree(d); hipFree(dp);
    return rc;
}

int gm_rfft_f32(size_t n, const float* in, gm_c32* out) {
    if (!in || !out || !n) return set_err(GM_ERR_INVALID_ARG, "null or empty");
    std::vector<gm_c32> buf(n);
    for (size_t i = 0; i < n; ++i) { buf[i].re = in[i]; buf[i].im = 0.0f; }
    const int rc = gm_fft_c2c_f32(n, 0, buf.data(), 1);
    if (!rc) memcpy(out, buf.data(), (n / 2 + 1) * sizeof(gm_c32));
    return rc;
}

// ---------------------------------------------------------------- acquisition
int gm_acq_destroy(gm_acq* a) {
    if (!a) return GM_OK;
    if (a->device >= 0) hipSetDevice(a->device);
    hipFree(a->d_planes);
    hipFree(a->d_comp_tmp); hipFree(a->d_comp_twn); hipFree(a->d_code_comb); hipFree(a->d_split_scratch); hipFree(a->d_split_counter);
    hipFree(a->fine.d_chips); hipFree(a->fine.d_tw1); hipFree(a->fine.d_tw2); hipFree(a->fine.d_B); hipFree(a->fine.d_mean);
    hipFree(a->fine.d_rowmax); hipFree(a->fine.d_rowarg); hipFree(a->fine.d_sat_worker); hipFree(a->fine.d_sat_cp);
    hipFree(a->fine.d_peak_pow); hipFree(a->fine.d_peak_idx);
    if (a->device >= 0) hipSetDevice(a->device);
    hipFree(a->d_tw_mix);
    hipFree(a->d_tables); hipFree(a->d_tw_fwd); hipFree(a->d_tw_inv); hipFree(a->d_code_fft); hipFree(a->d_code_fft_paired); hipFree(a->d_order);
    hipFree(a->d_spectra); hipFree(a->d_table_freq); hipFree(a->d_code_samples); hipFree(a->d_samples);
    hipFree(a->d_metrics); hipFree(a->d_worker_list); if (a->d_results) hipHostFree(a->d_results);
    hipFree(a->d_prn_ids);
    for (auto& e : a->tm.ev) if (e) hipEventDestroy(e);
    if (a->ahead.side) { hipStreamSynchronize(a->ahead.side); hipStreamDestroy(a->ahead.side); }
    if (a->ahead.ev_s) hipEventDestroy(a->ahead.ev_s);
    if (a->ahead.ev_f) hipEventDestroy(a->ahead.ev_f);
    if (a->ahead.ev_in) hipEventDestroy(a->ahead.ev_in);
    hipFree(a->ahead.d_spectra_alt);
    if (a->own_stream && a->stream) hipStreamDestroy(a->stream);
    delete a;
    return GM_OK;
}

int gm_acq_create(const gm_acq_cfg* cfg, gm_acq** out) {
    if (!cfg || !out) return set_err(GM_ERR_INVALID_ARG, "null cfg/out");
    *out = nullptr;
    if (!cfg->fft_size || !cfg->n_integrations || !cfg->n_bins || !cfg->n_prn || !cfg->prn_ids)
        return set_err(GM_ERR_INVALID_ARG, "fft_size, n_integrations, n_bins, n_prn, prn_ids are required");
    if (!cfg->tables && !cfg->doppler_hz) return set_err(GM_ERR_INVALID_ARG, "doppler_hz or tables required");
    if (cfg->tables && !cfg->table_freq) return set_err(GM_ERR_INVALID_ARG, "table_freq required with tables");
    if (cfg->fft_size % 8) return set_err(GM_ERR_ALIGNMENT, "fft_size % 8 != 0");
    const gm::PlanOps* pl = gm::find_plan(int(cfg->fft_size));
    const gm::CompOps* comp = nullptr;
    uint32_t comp_q = 1;
    if (pl && !cfg->strict_sum_order && !cfg->reference_products && gm::diag_int("GM_COMP_BASE", 0) > 0) {      // diagnostic: a size with a fused plan through the composite path
        const gm::CompOps* c = gm::find_comp(cfg->fft_size);
        if (c && c->nb == gm::diag_int("GM_COMP_BASE", 0)) pl = nullptr;
    }
    if (!pl) {   // N = Q * Nb with Nb one of the composite base plans (largest first): acq_composite.hip
        comp = gm::find_comp(cfg->fft_size);
        if (comp) { pl = gm::find_plan(comp->nb); comp_q = uint32_t(comp->q); }
    }
    if (!pl) return set_err(GM_ERR_UNSUPPORTED_N, "no in-LDS FFT plan for this fft_size, nor Q x {16384, 16368, 16000, 8192, 8184, 8000, 6000, 5000, 4000} with Q in {2,3,4,5,6,8}");
    // (strict_sum_order on a composite size: the kernels also store the power planes and a second kernel sums them in the reference's
    //  order — P * D * N * 4 bytes of planes: 189 MB at the configs[3] Galileo geometry)
    if (cfg->reference_products && comp_q > 1)
        return set_err(GM_ERR_INVALID_ARG, "reference_products needs an fft_size with an in-LDS plan (gm_fft_supported_sizes)");
    if (int rc = ensure_device(g_device)) return rc;

    gm_acq* a = new gm_acq();
    a->device = g_device;
    a->cfg = *cfg;
    a->plan = pl;
    a->N = cfg->fft_size; a->D = cfg->n_bins; a->M = cfg->n_integrations; a->P = cfg->n_prn;
    a->Q = comp_q; a->Nb = uint32_t(pl->n); a->comp = comp;
    if (cfg->threshold == 0.0f) a->cfg.threshold = 7.0f;
    a->code_rate = cfg->codes ? (cfg->code_rate > 0 ? cfg->code_rate : CA_RATE) : CA_RATE;
    a->prn_ids.assign(cfg->prn_ids, cfg->prn_ids + a->P);
    const size_t N = a->N, D = a->D, M = a->M, P = a->P;
    int rc = GM_OK;
    auto fail = [&](int code) { gm_acq_destroy(a); return code; };

    // Doppler tables (do_acquisition.rs:252-262) — host glibc cosf/sinf like the reference
    std::vector<gm_c32> tables(D * N);
    a->table_freq.resize(D);
    if (cfg->tables) {
        memcpy(tables.data(), cfg->tables, D * N * sizeof(gm_c32));
        memcpy(a->table_freq.data(), cfg->table_freq, D * sizeof(float));
    } else {
        for (size_t d = 0; d < D; ++d)
            gm_doppler_table_new(cfg->f_if, cfg->doppler_hz[d], cfg->fs, N, &a->table_freq[d], &tables[d * N]);
    }
    // replica samples (AcquisitionWorker::new :132-135)
    std::vector<int8_t> code_samples(P * N);
    a->code_len = cfg->codes ? cfg->code_len : 1023u;
    a->chips.resize(P * size_t(a->code_len));
    for (size_t p = 0; p < P; ++p) {
        if (cfg->codes) {
            if (!cfg->code_len) return fail(set_err(GM_ERR_INVALID_ARG, "code_len required with codes"));
            memcpy(&a->chips[p * a->code_len], cfg->codes + p * cfg->code_len, cfg->code_len);
            rc = resample_code(cfg->codes + p * cfg->code_len, cfg->code_len, true, a->code_rate, cfg->fs, N,
                               &code_samples[p * N]);
        } else {
            const uint8_t prn = a->prn_ids[p];
            if (prn < 1 || prn > 32) return fail(set_err(GM_ERR_OUT_OF_RANGE, "prn must be 1..=32"));
            // rustfft's process() panics unless the replica length equals fft_size (:135-137)
            if (num_samples_per_code(CA_RATE, cfg->fs, CA_LEN) != N)
                return fail(set_err(GM_ERR_INVALID_ARG, "fft_size != round(fs / 1 kHz)"));
            memcpy(&a->chips[p * 1023], ca_table().rows[prn - 1], 1023);
            rc = resample_code(ca_table().rows[prn - 1], 1023, false, CA_RATE, cfg->fs, N, &code_samples[p * N]);
        }
        if (rc) return fail(set_err(rc, "code resampling index out of range"));
    }
    // inverse twiddles: the in-LDS correlation kernel's own plan (CorrPlanOf) for a size that fits one LDS image, the registered plan's
    // (what the composite kernels run on) for Q x base
    const bool twi_corr = a->Q == 1;
    std::vector<cf> twf(size_t(pl->tw_total) + 1), twi(size_t(twi_corr ? pl->tw_total_corr : pl->tw_total) + 1);
    pl->fill_tw(twf.data(), false);
    if (twi_corr) pl->fill_tw_corr(twi.data());
    else pl->fill_tw(twi.data(), true);
    std::vector<cf> twm(size_t(pl->tw_total_mix) + 1);
    pl->fill_tw_mix(twm.data(), false);

#define HIPA(expr)                                                     \
    do {                                                               \
        hipError_t _e = (expr);                                        \
        if (_e != hipSuccess) return fail(hip_fail(_e, #expr));        \
    } while (0)
    HIPA(hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking));
    a->own_stream = true;
    HIPA(hipMalloc(&a->d_tables, D * N * 8));
    HIPA(hipMalloc(&a->d_table_freq, D * 4));
    HIPA(hipMalloc(&a->d_tw_fwd, twf.size() * 8));
    HIPA(hipMalloc(&a->d_tw_inv, twi.size() * 8));
    HIPA(hipMalloc(&a->d_tw_mix, twm.size() * 8));
    HIPA(hipMalloc(&a->d_code_samples, P * N));
    HIPA(hipMalloc(&a->d_code_fft, P * N * 8));
    HIPA(hipMalloc(&a->d_spectra, D * M * N * 8));
    a->samples_cap = M * N * 8;
    HIPA(hipMalloc(&a->d_samples, a->samples_cap));
    HIPA(hipMalloc(&a->d_metrics, 3 * P * D * 4));
    HIPA(hipMemsetAsync(a->d_metrics, 0, 3 * P * D * 4, a->stream));
    HIPA(hipMalloc(&a->d_worker_list, P * 4));
    if (a->Q == 1 && pl->split_slab && M >= 2) {
        // planes the tail split can ever use for THIS handle: every item cut (small grids: P*D items x M planes), or per XCD one
        // resident round of parts (64 slots; an item's M planes each) — never more than GM_CORR_SPLIT_MAX_SLABS
        // (a one-PRN AcquisitionWorker handle at N = 16368 takes 290 planes = 19 MB instead of 2560 = 251 MB)
        size_t planes = size_t(8) * ((P * D + 7) / 8) * M;      // (the launcher deals items to the XCDs in equal shares of ceil(P*D / 8))
        const size_t tail = size_t(8) * (64 + M);
        if (planes < tail) planes = tail;
        if (planes > size_t(gm::GM_CORR_SPLIT_MAX_SLABS)) planes = size_t(gm::GM_CORR_SPLIT_MAX_SLABS);
        a->split_planes = int(planes);
        HIPA(hipMalloc(&a->d_split_scratch, planes * pl->split_slab * sizeof(float)));
        HIPA(hipMalloc(&a->d_split_counter, gm::GM_CORR_SPLIT_MAX_ITEMS * sizeof(uint32_t)));
        HIPA(hipMemsetAsync(a->d_split_counter, 0, gm::GM_CORR_SPLIT_MAX_ITEMS * sizeof(uint32_t), a->stream));
    }
    HIPA(hipMemcpy(a->d_tables, tables.data(), D * N * 8, hipMemcpyHostToDevice));
    HIPA(hipMemcpy(a->d_table_freq, a->table_freq.data(), D * 4, hipMemcpyHostToDevice));
    HIPA(hipMemcpy(a->d_tw_fwd, twf.data(), twf.size() * 8, hipMemcpyHostToDevice));
    HIPA(hipMemcpy(a->d_tw_inv, twi.data(), twi.size() * 8, hipMemcpyHostToDevice));
    HIPA(hipMemcpy(a->d_tw_mix, twm.data(), twm.size() * 8, hipMemcpyHostToDevice));
    HIPA(hipMemcpy(a->d_code_samples, code_samples.data(), P * N, hipMemcpyHostToDevice));
    if ((rc = acq_reserve_results(a, uint32_t(P)))) return fail(rc);
    HIPA(hipMemcpy(a->d_prn_ids, a->prn_ids.data(), P, hipMemcpyHostToDevice));
    a->dev_prn_ids = a->prn_ids;
    // replica spectra: forward FFT of the resampled code (:136-138)
    int (*fill_order)(uint16_t*) = a->Q == 1 ? pl->fill_order : comp->fill_order;
    if (const int n_order = fill_order(nullptr)) {      // this (base) size's correlation plan reads a permuted spectrum
        std::vector<uint16_t> order(static_cast<size_t>(n_order), 0);
        fill_order(order.data());
        HIPA(hipMalloc(&a->d_order, order.size() * sizeof(uint16_t)));
        HIPA(hipMemcpy(a->d_order, order.data(), order.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    if (a->Q == 1) {
        pl->code_fft(a->stream, a->d_code_samples, a->d_tw_fwd, a->d_code_fft, int(P));
        HIPA(hipMalloc(&a->d_code_fft_paired, P * N * 8));
        pl->pair_codes(a->stream, a->d_code_fft, a->d_code_fft_paired, int(P));
    } else {     // the signal's two forward steps on the chips: natural order (the API's view), then every block paired
        const size_t tmp_items = D * M > P ? D * M : P;
        HIPA(hipMalloc(&a->d_comp_tmp, tmp_items * N * 8));
        if (cfg->strict_sum_order) HIPA(hipMalloc(&a->d_planes, P * D * N * sizeof(float)));
        HIPA(hipMalloc(&a->d_comp_twn, N * 8));
        HIPA(hipMalloc(&a->d_code_fft_paired, P * N * 8));
        std::vector<gm::cf> twn(N);
        comp->fill_twn(twn.data());
        HIPA(hipMemcpy(a->d_comp_twn, twn.data(), N * 8, hipMemcpyHostToDevice));
        comp->fwd_sub(a->stream, nullptr, GM_FMT_C32, nullptr, a->d_code_samples, a->d_tw_mix, a->d_comp_tmp, uint32_t(P), 1, nullptr);
        comp->fwd_post(a->stream, a->d_comp_tmp, a->d_code_fft, uint32_t(P), 0, nullptr);
        comp->relayout(a->stream, a->d_code_fft, a->d_code_fft_paired, int(P * a->Q));
        HIPA(hipMalloc(&a->d_code_comb, P * a->Q * N * 8));      // [code][n1][k1][pos]: the whole code-side factor per sub-transform
        comp->comb(a->stream, a->d_code_fft_paired, a->d_comp_twn, a->d_code_comb, uint32_t(P));
        if (a->d_order) {     // sub-transforms leave in storage order: the signal's forward step 2 goes into the table as well (CompOps::fold_post)
            gm::cf* comb2 = nullptr;
            HIPA(hipMalloc(&comb2, P * a->Q * N * 8));
            comp->fold_post(a->stream, a->d_code_comb, a->d_order, comb2, uint32_t(P));
            HIPA(hipStreamSynchronize(a->stream));
            hipFree(a->d_code_comb);
            a->d_code_comb = comb2;
            a->comp_post_folded = true;
        }
    }
    HIPA(hipGetLastError());
    HIPA(hipStreamSynchronize(a->stream));
    a->worker_list.clear();
    a->n_workers = 0xffffffffu;   // force the first upload
    if ((rc = acq_set_mask(a, ~0ull))) return fail(rc);
#undef HIPA
    *out = a;
    return GM_OK;
}

int gm_acq_set_stream(gm_acq* a, void* s) {
    if (!a) return set_err(GM_ERR_INVALID_ARG, "null handle");
    if (a->dec_deferred) { if (int rc = ensure_device(a->device)) return rc; if (int rc = acq_flush_decision(a)) return rc; }
    if (a->own_stream && a->stream) { hipStreamSynchronize(a->stream); hipStreamDestroy(a->stream); }
    a->stream = reinterpret_cast<hipStream_t>(s);
    a->own_stream = false;
    return GM_OK;
}

int gm_acq_set_prn_mask(gm_acq* a, uint64_t mask) {
    if (!a) return set_err(GM_ERR_INVALID_ARG, "null handle");
    if (int rc = ensure_device(a->device)) return rc;
    return acq_set_mask(a, mask);
}

// stage F (unless `prepared`: then the spectra wait in the second buffer) + stage C of one dwell on the handle's stream
static int acq_search_common(gm_acq* a, const void* d_samples, int fmt, void* d_metrics, bool prepared) {
    uint32_t* met = d_metrics ? static_cast<uint32_t*>(d_metrics) : a->d_metrics;
    const size_t PD = size_t(a->P) * a->D;
    const bool t = a->tm.on && (a->tm.calls++ % a->tm.stride == 0);
    a->tm.this_call = t;
    // a timed dwell measures its own kernels only: a decision still pending from the dwell before runs now, in front of the first
    // event, not as trailing workgroups of the timed stage F
    if (t) { if (int rc = acq_flush_decision(a)) return rc; }
    hipEvent_t* ev = t ? &a->tm.ev[size_t(a->tm.count % Timing::CAP) * 4] : nullptr;
    if (t) HIPC(hipEventRecord(ev[0], a->stream));
    gm_acq::Ahead& ah = a->ahead;
    bool tickets_cleared = a->d_split_counter != nullptr;
    if (prepared) {
        // the spectra were prepared (gm_acq_prepare_dev): take that buffer once its stage F is through; the tail split's tickets,
        // which the in-stream stage F clears on its way, are left to corr() (a memset, if it cuts the tail)
        if (int rc = acq_flush_decision(a)) return rc;
        ah.valid = false;
        std::swap(a->d_spectra, ah.d_spectra_alt);
        HIPC(hipStreamWaitEvent(a->stream, ah.ev_f, 0));
        tickets_cleared = false;
    } else if (a->Q == 1) {
        a->plan->mix_fft(a->stream, d_samples, fmt, a->d_tables, a->d_tw_mix, a->d_spectra, int(a->D), int(a->M), a->d_split_counter, a->d_order,
                         a->dec_deferred ? &a->dec_args : nullptr);
        a->dec_deferred = false;
    } else {
        if (a->comp_post_folded) {     // the sub-transforms ARE what the correlation kernel reads
            a->comp->fwd_sub(a->stream, d_samples, fmt, a->d_tables, nullptr, a->d_tw_mix, a->d_spectra, a->D * a->M, a->M, a->d_order);
        } else {
            a->comp->fwd_sub(a->stream, d_samples, fmt, a->d_tables, nullptr, a->d_tw_mix, a->d_comp_tmp, a->D * a->M, a->M, a->d_order);
            a->comp->fwd_post(a->stream, a->d_comp_tmp, a->d_spectra, a->D * a->M, 1, a->d_order);
        }
    }
    if (t) HIPC(hipEventRecord(ev[1], a->stream));
    if (ah.side) HIPC(hipEventRecord(ah.ev_s, a->stream));
    if (a->Q == 1) {
        a->plan->corr(a->stream, a->d_spectra, a->plan->code_paired ? a->d_code_fft_paired : a->d_code_fft, a->d_tw_inv, reinterpret_cast<float*>(met), met + PD,
                      reinterpret_cast<float*>(met + 2 * PD), a->d_worker_list, int(a->n_workers), int(a->D), int(a->M),
                      a->d_split_scratch, a->split_planes, a->d_split_counter, a->cfg.strict_sum_order ? 1 : 0, tickets_cleared ? 1 : 0,
                      a->cfg.reference_products ? 1 : 0);
    } else if (a->n_workers) {
        a->comp->corr(a->stream, a->d_spectra, a->d_code_comb, a->d_comp_twn, a->d_tw_inv, reinterpret_cast<float*>(met),
                      met + PD, reinterpret_cast<float*>(met + 2 * PD), a->d_worker_list, int(a->n_workers), int(a->D), int(a->M), a->d_planes);
        if (a->d_planes)      // strict_sum_order: the plane sums once more, in is_good_satellite's own order (:229-235)
            gm::launch_plane_strict_sum(a->stream, a->d_planes, reinterpret_cast<float*>(met + 2 * PD), a->d_worker_list, int(a->n_workers), int(a->D), uint32_t(a->N));
    }
    if (t) { HIPC(hipEventRecord(ev[2], a->stream)); a->tm.count++; a->tm.decide_valid = false; }
    HIPC(hipGetLastError());
    a->last_metrics = met;
    a->last_samples = d_samples; a->last_fmt = fmt;
    return GM_OK;
}

int gm_acq_search_dev(gm_acq* a, const void* d_samples, int fmt, void* d_metrics) {
    if (!a || !d_samples) return set_err(GM_ERR_INVALID_ARG, "null handle/samples");
    if (fmt < GM_FMT_C32 || fmt > GM_FMT_I8_REAL) return set_err(GM_ERR_INVALID_ARG, "bad sample format");
    if (int rc = ensure_device(a->device)) return rc;
    // always from the samples as they are NOW: a preparation (gm_acq_prepare_dev) is never matched by address
    return acq_search_common(a, d_samples, fmt, d_metrics, false);
}

int gm_acq_search_prepared_dev(gm_acq* a, uint64_t token, void* d_metrics) {
    if (!a) return set_err(GM_ERR_INVALID_ARG, "null handle");
    gm_acq::Ahead& ah = a->ahead;
    if (!token || !ah.valid || ah.token != token)
        return set_err(GM_ERR_INVALID_ARG, "no such preparation (token stale, consumed, replaced or dropped)");
    if (int rc = ensure_device(a->device)) return rc;
    if (a->Q != 1) {                        // composite sizes prepare nothing: the whole search runs now, from the samples named then,
        ah.valid = false;                   // behind what `ready_stream` held at the time of gm_acq_prepare_dev
        if (ah.comp_wait) { HIPC(hipStreamWaitEvent(a->stream, ah.ev_in, 0)); ah.comp_wait = false; }
        return acq_search_common(a, ah.samples, ah.fmt, d_metrics, false);
    }
    return acq_search_common(a, ah.samples, ah.fmt, d_metrics, true);
}

int gm_acq_drop_prepared(gm_acq* a) {
    if (!a) return set_err(GM_ERR_INVALID_ARG, "null handle");
    a->ahead.valid = false;                 // (a stage F still running on the side stream finishes into the spare buffer: harmless)
    return GM_OK;
}

int gm_acq_prepare_dev(gm_acq* a, const void* d_samples, int fmt, void* ready_stream, uint64_t* token) {
    if (!a || !d_samples || !token) return set_err(GM_ERR_INVALID_ARG, "null handle/samples/token");
    if (fmt < GM_FMT_C32 || fmt > GM_FMT_I8_REAL) return set_err(GM_ERR_INVALID_ARG, "bad sample format");
    *token = 0;
    gm_acq::Ahead& ah = a->ahead;
    if (int rc = ensure_device(a->device)) return rc;
    if (a->Q != 1) {                        // composite sizes: nothing is prepared, gm_acq_search_prepared_dev does all of it —
        // but the ordering promise is the same: the samples are complete once the work queued on `ready_stream` so far has run, so
        // the event is recorded NOW and the search waits for it on the handle's stream (the token used to be handed
        // out before `ready_stream` was looked at)
        ah.valid = false; ah.comp_wait = false;
        if (ready_stream) {
            if (!ah.ev_in) HIPC(hipEventCreateWithFlags(&ah.ev_in, hipEventDisableTiming));
            HIPC(hipEventRecord(ah.ev_in, reinterpret_cast<hipStream_t>(ready_stream)));
            ah.comp_wait = true;
        }
        ah.valid = true; ah.samples = d_samples; ah.fmt = fmt; ah.token = *token = ah.next_token++;
        return GM_OK;
    }
    if (!ah.side) {
        // all four resources or none: a failure half-way must leave the handle as it was (a later call starts over)
        cf* buf = nullptr; hipStream_t side = nullptr; hipEvent_t e[3] = {nullptr, nullptr, nullptr};
        // lowest priority: its workgroups are wanted where stage C has none left to place (the idle CUs of its last round), not
        // beside stage C's first round on every CU — the dispatcher honours that only in part (N = 16368: 307.3 us per dwell at
        // normal priority, 304.7 at the lowest, before the ticket memset went: DESIGN_HISTORY R4 has the table)
        int least = 0, greatest = 0;
        hipError_t er = hipMalloc(&buf, size_t(a->D) * a->M * a->N * 8);
        if (er == hipSuccess) er = hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (er == hipSuccess) er = hipStreamCreateWithPriority(&side, hipStreamNonBlocking, gm::diag_int("GM_PREPARE_PRIORITY", least));
        for (int i = 0; i < 3 && er == hipSuccess; ++i) er = hipEventCreateWithFlags(&e[i], hipEventDisableTiming);
        if (er == hipSuccess) er = hipEventRecord(e[0], a->stream);      // first time: behind whatever the handle's stream holds so far
        if (er != hipSuccess) {
            for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x);
            if (side) (void)hipStreamDestroy(side);
            if (buf) (void)hipFree(buf);
            return set_err(GM_ERR_HIP, hipGetErrorString(er));
        }
        ah.d_spectra_alt = buf; ah.side = side; ah.ev_s = e[0]; ah.ev_f = e[1]; ah.ev_in = e[2];
    }
    // the buffer to fill was last read by a stage C in front of the most recent one: free at ev_s (an earlier, unclaimed
    // preparation is overwritten in stream order)
    HIPC(hipStreamWaitEvent(ah.side, ah.ev_s, 0));
    if (ready_stream) {                     // ... and the samples are complete once the work queued on `ready_stream` so far has run
        HIPC(hipEventRecord(ah.ev_in, reinterpret_cast<hipStream_t>(ready_stream)));
        HIPC(hipStreamWaitEvent(ah.side, ah.ev_in, 0));
    }
    ah.valid = false;
    a->plan->mix_fft(ah.side, d_samples, fmt, a->d_tables, a->d_tw_mix, ah.d_spectra_alt, int(a->D), int(a->M), nullptr, a->d_order, nullptr);
    HIPC(hipEventRecord(ah.ev_f, ah.side));
    HIPC(hipGetLastError());
    ah.valid = true; ah.samples = d_samples; ah.fmt = fmt; ah.token = *token = ah.next_token++;
    return GM_OK;
}

int gm_acq_decide_dev(gm_acq* a, const void* d_metrics, uint32_t n_prn, const uint8_t* prn_ids, uint64_t local_tail) {
    if (!a || !n_prn) return set_err(GM_ERR_INVALID_ARG, "null handle / n_prn == 0");
    if (int rc = ensure_device(a->device)) return rc;
    if (int rc = acq_flush_decision(a)) return rc;
    if (int rc = acq_reserve_results(a, n_prn)) return rc;
    const uint32_t* met = d_metrics ? static_cast<const uint32_t*>(d_metrics) : a->d_metrics;
    const uint8_t* ids = prn_ids ? prn_ids : (n_prn == a->P ? a->prn_ids.data() : nullptr);
    if (!ids) return set_err(GM_ERR_INVALID_ARG, "prn_ids required when n_prn differs from the handle's");
    if (a->dev_prn_ids.size() != n_prn || memcmp(a->dev_prn_ids.data(), ids, n_prn) != 0) {   // upload only on change
        a->dev_prn_ids.assign(ids, ids + n_prn);
        HIPC(hipMemcpyAsync(a->d_prn_ids, a->dev_prn_ids.data(), n_prn, hipMemcpyHostToDevice, a->stream));
        HIPC(hipStreamSynchronize(a->stream));
    }
    const size_t PD = size_t(n_prn) * a->D;
    gm::DecideArgs da;
    da.mmax = reinterpret_cast<const float*>(met);
    da.margmax = met + PD;
    da.msum = reinterpret_cast<const float*>(met + 2 * PD);
    da.table_freq = a->d_table_freq;
    da.prn_ids = a->d_prn_ids;
    da.mask_lo = (d_metrics && n_prn != a->P) ? ~0ull : a->mask;
    da.n_prn = int(n_prn); da.n_bins = int(a->D); da.fft_size = int(a->N);
    da.fs = a->cfg.fs; da.threshold = a->cfg.threshold; da.code_rate = a->code_rate;
    da.best_bin_mode = a->cfg.decision_mode == GM_DECIDE_BEST_BIN ? 1 : 0;
    da.local_tail = local_tail;
    da.results = a->d_results; da.found = a->d_found;
    if (a->defer_decisions && a->Q == 1 && !(a->tm.on && a->tm.this_call) && met == a->last_metrics && a->D <= 64) {
        a->dec_args = da;          // runs with the next gm_acq_search_dev's stage F, or at the next flush point
        a->dec_deferred = true;
        return GM_OK;
    }
    gm::launch_decide(a->stream, da);
    if (a->tm.on && a->tm.this_call && a->tm.count > 0) {
        a->tm.this_call = false;
        HIPC(hipEventRecord(a->tm.ev[size_t((a->tm.count - 1) % Timing::CAP) * 4 + 3], a->stream));
        a->tm.decide_valid = true;
    }
    HIPC(hipGetLastError());
    return GM_OK;
}

int gm_acq_synchronize(gm_acq* a) {
    if (!a) return set_err(GM_ERR_INVALID_ARG, "null handle");
    if (int rc = ensure_device(a->device)) return rc;
    if (int rc = acq_flush_decision(a)) return rc;
    HIPC(hipStreamSynchronize(a->stream));
    return GM_OK;
}

int gm_acq_set_deferred_decision(gm_acq* a, int on) {
    if (!a) return set_err(GM_ERR_INVALID_ARG, "null handle");
    if (!on && a->dec_deferred) { if (int rc = ensure_device(a->device)) return rc; if (int rc = acq_flush_decision(a)) return rc; }
    a->defer_decisions = on != 0;
    return GM_OK;
}

int gm_acq_fetch_results(gm_acq* a, uint32_t n_prn, gm_acq_result* results, uint8_t* found) {
    if (!a || !results || !found) return set_err(GM_ERR_INVALID_ARG, "null pointer");
    if (n_prn > a->results_cap) return set_err(GM_ERR_INVALID_ARG, "n_prn exceeds the last decide call");
    if (int rc = ensure_device(a->device)) return rc;
    if (int rc = acq_flush_decision(a)) return rc;
    HIPC(hipStreamSynchronize(a->stream));          // decide_kernel's stores to the pinned block are visible once its stream has drained
    memcpy(results, a->d_results, sizeof(gm_acq_result) * n_prn);
    memcpy(found, a->d_found, n_prn);
    return GM_OK;
}

int gm_acq_search(gm_acq* a, const void* samples, size_t n_samples, int fmt, uint64_t local_tail, uint64_t prn_mask,
                  gm_acq_result* results, uint8_t* found) {
    if (!a || !samples || !results || !found) return set_err(GM_ERR_INVALID_ARG, "null pointer");
    if (fmt < GM_FMT_C32 || fmt > GM_FMT_I8_REAL) return set_err(GM_ERR_INVALID_ARG, "bad sample format");
    const size_t need = size_t(a->M) * a->N;
    if (n_samples < need) return set_err(GM_ERR_OUT_OF_RANGE, "samples_chunk shorter than num_integrations*fft_size");
    if (int rc = ensure_device(a->device)) return rc;
    const size_t bps = fmt == GM_FMT_C32 ? 8 : (fmt == GM_FMT_I8_IQ ? 2 : 1);
    if (int rc = acq_set_mask(a, prn_mask)) return rc;
    HIPC(hipMemcpyAsync(a->d_samples, samples, need * bps, hipMemcpyHostToDevice, a->stream));   // (a pinned staging block of the handle's own: measured, no gain)
    if (int rc = gm_acq_search_dev(a, a->d_samples, fmt, nullptr)) return rc;
    if (int rc = gm_acq_decide_dev(a, nullptr, a->P, nullptr, local_tail)) return rc;
    return gm_acq_fetch_results(a, a->P, results, found);
}

// run()'s snapshot + search (do_acquisition.rs:297-313) against the DEVICE ring: the n_integrations*fft_size
// samples ending at `head` (local_tail = head - M*N) are copied device-to-device (wrap-aware, like
// copy_to_slice :107-129) and searched; no host round trip of the samples.
int gm_acq_search_ring(gm_acq* a, gm_ring* ring, uint64_t prn_mask, gm_acq_result* results, uint8_t* found,
                       uint64_t* local_tail_out) {
    if (!a || !ring || !results || !found) return set_err(GM_ERR_INVALID_ARG, "null pointer");
    if (a->device != ring->device) return set_err(GM_ERR_INVALID_ARG, "ring lives on another device");
    const size_t need = size_t(a->M) * a->N;
    if (need > ring->size) return set_err(GM_ERR_OUT_OF_RANGE, "ring smaller than num_integrations*fft_size");
    if (int rc = ensure_device(ring->device)) return rc;
    ring_refresh_head(ring);
    const uint64_t head = ring->head.load(std::memory_order_acquire);
    if ((int64_t)(head - need) < 0) return set_err(GM_ERR_OUT_OF_RANGE, "not enough samples yet (head < M*N)");   // :299
    if (int rc = ensure_device(a->device)) return rc;
    const uint64_t local_tail = head - need;
    const size_t ps = size_t(local_tail & ring->mask);
    cf* dst = static_cast<cf*>(a->d_samples);
    if (ps + need <= ring->size) {
        HIPC(hipMemcpyAsync(dst, ring->d_buf + ps, need * 8, hipMemcpyDeviceToDevice, a->stream));
    } else {
        const size_t first = ring->size - ps;
        HIPC(hipMemcpyAsync(dst, ring->d_buf + ps, first * 8, hipMemcpyDeviceToDevice, a->stream));
        HIPC(hipMemcpyAsync(dst + first, ring->d_buf, (need - first) * 8, hipMemcpyDeviceToDevice, a->stream));
    }
    if (int rc = acq_set_mask(a, prn_mask)) return rc;
    if (int rc = gm_acq_search_dev(a, a->d_samples, GM_FMT_C32, nullptr)) return rc;
    if (int rc = gm_acq_decide_dev(a, nullptr, a->P, nullptr, local_tail)) return rc;
    if (local_tail_out) *local_tail_out = local_tail;
    return gm_acq_fetch_results(a, a->P, results, found);
}

// Fine Doppler: finer_doppler (acquisition_bk.rs:215-302, legacy) for every found result of the last
// search, on the snapshot that search used (still resident in HBM).  fft_size = 8 * next_pow2((M-1)*N) as N1 x N2.
int gm_acq_finer_doppler(gm_acq* a, const gm_acq_result* results, const uint8_t* found, uint32_t n_prn, float* fine_freq_hz,
                         uint64_t* peak_index, float* peak_mag, uint64_t* fft_size_out) {
    if (!a || !results || !found) return set_err(GM_ERR_INVALID_ARG, "null pointer");
    if (n_prn > a->P) return set_err(GM_ERR_INVALID_ARG, "n_prn exceeds the handle's workers");
    if (a->M < 2) return set_err(GM_ERR_INVALID_ARG, "fine Doppler needs num_integrations >= 2 ((M-1)*N samples after the code phase)");
    if (!a->last_samples) return set_err(GM_ERR_INVALID_ARG, "no search has run on this handle yet");
    if (int rc = ensure_device(a->device)) return rc;
    gm_acq::Fine& f = a->fine;
    const uint32_t size_use = (a->M - 1) * a->N;                         // :240
    if (!f.p1) {
        uint64_t p2 = 1;
        while (p2 < size_use) p2 <<= 1;                                  // next_power_of_two :249
        const uint64_t n = 8 * p2;
        if (n > (1ull << 24)) return set_err(GM_ERR_UNSUPPORTED_N, "fine-Doppler FFT longer than 2^24");
        // N = N1 * N2 with both factors among the power-of-two in-LDS plans, as square as possible
        int lg = 0;
        while ((1ull << lg) < n) ++lg;
        const gm::PlanOps *b1 = nullptr, *b2 = nullptr;
        for (int l2 = lg / 2; l2 >= 8 && !b1; --l2) {
            const gm::PlanOps *q1 = gm::find_plan(1 << (lg - l2)), *q2 = gm::find_plan(1 << l2);
            if (q1 && q2 && q1->fine_cols && q2->fine_rows) { b1 = q1; b2 = q2; }
        }
        if (!b1) return set_err(GM_ERR_UNSUPPORTED_N, "no pair of in-LDS plans factors the fine-Doppler FFT size");
        std::vector<cf> t1(size_t(b1->tw_total) + 1), t2(size_t(b2->tw_total) + 1);
        b1->fill_tw(t1.data(), false);
        b2->fill_tw(t2.data(), false);
        HIPC(hipMalloc(&f.d_tw1, t1.size() * 8));
        HIPC(hipMalloc(&f.d_tw2, t2.size() * 8));
        HIPC(hipMemcpy(f.d_tw1, t1.data(), t1.size() * 8, hipMemcpyHostToDevice));
        HIPC(hipMemcpy(f.d_tw2, t2.data(), t2.size() * 8, hipMemcpyHostToDevice));
        HIPC(hipMalloc(&f.d_chips, a->chips.size()));
        HIPC(hipMemcpy(f.d_chips, a->chips.data(), a->chips.size(), hipMemcpyHostToDevice));
        HIPC(hipMalloc(&f.d_mean, 2 * sizeof(float)));
        f.p1 = b1; f.p2 = b2; f.size_use = size_use;
    }
    const uint32_t N1 = uint32_t(f.p1->n), N2 = uint32_t(f.p2->n);
    std::vector<uint32_t> workers, cps, slot(n_prn, 0xFFFFFFFFu);
    for (uint32_t p = 0; p < n_prn; ++p) {
        if (!found[p]) continue;
        if (results[p].code_phase_samples + size_use > uint64_t(a->M) * a->N)
            return set_err(GM_ERR_OUT_OF_RANGE, "code_phase + (M-1)*N exceeds the snapshot (:260 would panic)");
        slot[p] = uint32_t(workers.size());
        workers.push_back(p);
        cps.push_back(uint32_t(results[p].code_phase_samples));
    }
    const uint32_t S = uint32_t(workers.size());
    if (fft_size_out) *fft_size_out = uint64_t(N1) * N2;
    if (!S) return GM_OK;
    if (S > f.sats_cap) {
        hipFree(f.d_B); hipFree(f.d_rowmax); hipFree(f.d_rowarg); hipFree(f.d_sat_worker); hipFree(f.d_sat_cp);
        hipFree(f.d_peak_pow); hipFree(f.d_peak_idx);
        f.d_B = nullptr; f.sats_cap = 0;
        HIPC(hipMalloc(&f.d_B, size_t(S) * N1 * N2 * 8));
        HIPC(hipMalloc(&f.d_rowmax, size_t(S) * N1 * 4));
        HIPC(hipMalloc(&f.d_rowarg, size_t(S) * N1 * 4));
        HIPC(hipMalloc(&f.d_sat_worker, S * 4));
        HIPC(hipMalloc(&f.d_sat_cp, S * 4));
        HIPC(hipMalloc(&f.d_peak_pow, S * 4));
        HIPC(hipMalloc(&f.d_peak_idx, S * 4));
        f.sats_cap = S;
    }
    HIPC(hipMemcpyAsync(f.d_sat_worker, workers.data(), S * 4, hipMemcpyHostToDevice, a->stream));
    HIPC(hipMemcpyAsync(f.d_sat_cp, cps.data(), S * 4, hipMemcpyHostToDevice, a->stream));
    gm::FineArgs fa{};
    fa.samples = a->last_samples; fa.fmt = a->last_fmt; fa.mean = f.d_mean;
    fa.chips = f.d_chips; fa.code_len = a->code_len; fa.code_rate = a->code_rate; fa.fs = a->cfg.fs;
    fa.sat_worker = f.d_sat_worker; fa.sat_code_phase = f.d_sat_cp;
    fa.size_use = size_use; fa.N1 = N1; fa.N2 = N2; fa.B = f.d_B; fa.tw1 = f.d_tw1; fa.tw2 = f.d_tw2;
    fa.rowmax = f.d_rowmax; fa.rowarg = f.d_rowarg;
    gm::launch_fine_mean(a->stream, a->last_samples, a->last_fmt, a->M * a->N, f.d_mean);
    f.p1->fine_cols(a->stream, fa, int(S));
    f.p2->fine_rows(a->stream, fa, int(S));
    gm::launch_fine_final(a->stream, f.d_rowmax, f.d_rowarg, N1 / uint32_t(f.p2->fine_rows_per_wg), int(S), f.d_peak_pow, f.d_peak_idx);
    HIPC(hipGetLastError());
    std::vector<float> pw(S);
    std::vector<uint32_t> pi(S);
    HIPC(hipMemcpyAsync(pw.data(), f.d_peak_pow, S * 4, hipMemcpyDeviceToHost, a->stream));
    HIPC(hipMemcpyAsync(pi.data(), f.d_peak_idx, S * 4, hipMemcpyDeviceToHost, a->stream));
    HIPC(hipStreamSynchronize(a->stream));
    const uint64_t fft_size = uint64_t(N1) * N2;
    const uint64_t one_side = uint64_t(ceilf((float(fft_size) + 1.0f) / 2.0f));       // :250
    for (uint32_t p = 0; p < n_prn; ++p) {
        if (slot[p] == 0xFFFFFFFFu) continue;
        const uint64_t idx = pi[slot[p]];
        if (peak_index) peak_index[p] = idx;
        if (peak_mag) peak_mag[p] = sqrtf(pw[slot[p]]);
        if (fine_freq_hz)   // :251-253; the upper half is where the legacy indexes out of bounds (:285-288): negative frequency
            fine_freq_hz[p] = idx > one_side ? -((float(fft_size - idx) * a->cfg.fs) / float(fft_size))
                                             : (float(idx) * a->cfg.fs) / float(fft_size);
    }
    return GM_OK;
}

int gm_acq_search_c32(gm_acq* a, const gm_c32* s, size_t n, uint64_t tail, uint64_t mask, gm_acq_result* r, uint8_t* f) {
    return gm_acq_search(a, s, n, GM_FMT_C32, tail, mask, r, f);
}
int gm_acq_search_i8(gm_acq* a, const int8_t* s, size_t n, uint64_t tail, uint64_t mask, gm_acq_result* r, uint8_t* f) {
    return gm_acq_search(a, s, n, GM_FMT_I8_IQ, tail, mask, r, f);
}

int gm_acq_metrics(gm_acq* a, float* mx, uint32_t* am, float* sm) {
    if (!a) return set_err(GM_ERR_INVALID_ARG, "null handle");
    if (int rc = ensure_device(a->device)) return rc;
    HIPC(hipStreamSynchronize(a->stream));
    const size_t PD = size_t(a->P) * a->D;
    const uint32_t* met = a->last_metrics ? static_cast<const uint32_t*>(a->last_metrics) : a->d_metrics;
    if (mx) HIPC(hipMemcpy(mx, met, PD * 4, hipMemcpyDeviceToHost));
    if (am) HIPC(hipMemcpy(am, met + PD, PD * 4, hipMemcpyDeviceToHost));
    if (sm) HIPC(hipMemcpy(sm, met + 2 * PD, PD * 4, hipMemcpyDeviceToHost));
    return GM_OK;
}

int gm_acq_code_fft(gm_acq* a, uint32_t worker, gm_c32* out) {
    if (!a || !out || worker >= a->P) return set_err(GM_ERR_INVALID_ARG, "bad worker index");
    if (int rc = ensure_device(a->device)) return rc;
    // natural order at every size (the composite path keeps blocks k1*Nb + k2 in natural order too)
    HIPC(hipMemcpy(out, a->d_code_fft + size_t(worker) * a->N, size_t(a->N) * 8, hipMemcpyDeviceToHost));
    return GM_OK;
}

int gm_acq_tables(gm_acq* a, gm_c32* tables, float* freq) {
    if (!a) return set_err(GM_ERR_INVALID_ARG, "null handle");
    if (int rc = ensure_device(a->device)) return rc;
    if (tables) HIPC(hipMemcpy(tables, a->d_tables, size_t(a->D) * a->N * 8, hipMemcpyDeviceToHost));
    if (freq) memcpy(freq, a->table_freq.data(), a->D * sizeof(float));
    return GM_OK;
}

int gm_acq_enable_timing(gm_acq* a, int on) {
    if (!a) return set_err(GM_ERR_INVALID_ARG, "null handle");
    if (int rc = ensure_device(a->device)) return rc;
    if (on && a->tm.ev.empty()) {
        a->tm.ev.assign(size_t(Timing::CAP) * 4, nullptr);
        for (auto& e : a->tm.ev) HIPC(hipEventCreate(&e));
    }
    a->tm.on = on != 0;
    a->tm.stride = on > 1 ? on : 1;      // on = k > 1: every k-th search is timed
    a->tm.calls = 0;
    a->tm.this_call = false;
    a->tm.count = 0;
    a->tm.decide_valid = false;
    return GM_OK;
}

int gm_acq_last_timing(gm_acq* a, float* ms_mix, float* ms_corr, float* ms_decide) {
    if (!a || !a->tm.on || a->tm.count == 0) return set_err(GM_ERR_INVALID_ARG, "timing not enabled / nothing recorded");
    if (int rc = ensure_device(a->device)) return rc;
    HIPC(hipStreamSynchronize(a->stream));
    hipEvent_t* ev = &a->tm.ev[size_t((a->tm.count - 1) % Timing::CAP) * 4];
    float t = 0;
    if (ms_mix) { HIPC(hipEventElapsedTime(&t, ev[0], ev[1])); *ms_mix = t; }
    if (ms_corr) { HIPC(hipEventElapsedTime(&t, ev[1], ev[2])); *ms_corr = t; }
    if (ms_decide) {
        *ms_decide = 0;
        if (a->tm.decide_valid) { HIPC(hipEventElapsedTime(&t, ev[2], ev[3])); *ms_decide = t; }
    }
    return GM_OK;
}

int gm_acq_timing_summary(gm_acq* a, uint32_t* launches, float* avg_ms_mix, float* avg_ms_corr) {
    if (!a || !a->tm.on) return set_err(GM_ERR_INVALID_ARG, "timing not enabled");
    if (int rc = ensure_device(a->device)) return rc;
    HIPC(hipStreamSynchronize(a->stream));
    const int n = a->tm.count < Timing::CAP ? a->tm.count : Timing::CAP;
    double sm = 0, sc = 0;
    for (int i = 0; i < n; ++i) {
        hipEvent_t* ev = &a->tm.ev[size_t(i) * 4];
        float t = 0;
        HIPC(hipEventElapsedTime(&t, ev[0], ev[1])); sm += t;
        HIPC(hipEventElapsedTime(&t, ev[1], ev[2])); sc += t;
    }
    if (launches) *launches = uint32_t(n);
    if (avg_ms_mix) *avg_ms_mix = n ? float(sm / n) : 0.f;
    if (avg_ms_corr) *avg_ms_corr = n ? float(sc / n) : 0.f;
    return GM_OK;
}

// Host replay of the decision (same arithmetic as decide_kernel) for callers that hold the metrics on the
// host, e.g. after a gloo/MPI all-gather.  No device needed.
int gm_acq_decide_host(const float* mmax, const uint32_t* margmax, const float* msum, const float* table_freq,
                       uint32_t n_prn, uint32_t n_bins, const uint8_t* prn_ids, uint32_t fft_size, float fs,
                       float code_rate, float threshold, uint64_t local_tail, gm_acq_result* results, uint8_t* found) {
    if (!mmax || !margmax || !msum || !table_freq || !prn_ids || !results || !found || fft_size < 2)
        return set_err(GM_ERR_INVALID_ARG, "null pointer");
    if (threshold == 0.0f) threshold = 7.0f;
    if (!(code_rate > 0.0f)) code_rate = CA_RATE;
    const float nm1 = float(fft_size - 1);
    for (uint32_t p = 0; p < n_prn; ++p) {
        gm_acq_result r;
        memset(&r, 0, sizeof(r));
        r.prn = prn_ids[p]; r.doppler_bin = -1;
        found[p] = 0;
        float gmax = 0.0f, bfreq = 0.0f, bsum = 0.0f;
        uint32_t bphase = 0;
        int bbin = -1;
        for (uint32_t d = 0; d < n_bins; ++d) {
            const size_t o = size_t(p) * n_bins + d;
            if (mmax[o] > gmax) { gmax = mmax[o]; bfreq = table_freq[d]; bphase = margmax[o]; bsum = msum[o]; bbin = int(d); }
            const float avg = (bsum - gmax) / nm1;
            if (gmax / avg > threshold) {
                r.code_phase_samples = bphase;
                r.code_phase_chips = float(bphase) * code_rate / fs;
                r.carrier_freq = bfreq; r.fs = fs; r.mag_relative = gmax;
                r.sample_global_index = local_tail + bphase; r.doppler_bin = bbin;
                found[p] = 1;
                break;
            }
        }
        results[p] = r;
    }
    return GM_OK;
}

// Diagnostic (not in the reference): arm (out == NULL) / read back phase stamps of workgroup 0 of acq_corr_kernel:
// out = [n_integrations][8 waves][8 phases] shader-clock values of the last search.
int gm_acq_debug_stamps(gm_acq* a, long long* out) {
    if (!a) return set_err(GM_ERR_INVALID_ARG, "null handle");
    if (!gm::corr_stamps_built())
        return set_err(GM_ERR_UNSUPPORTED, "the stamped correlation kernels are in a diagnostic build only (GM_EXTRA_FLAGS=-DGM_DIAG_STAMPS, tools/README.md)");
    if (int rc = ensure_device(a->device)) return rc;
    static long long* d_st = nullptr;
    const size_t bytes = size_t(a->M) * 64 * sizeof(long long);
    if (!out) {
        if (d_st) { gm::set_corr_stamps(nullptr); hipFree(d_st); d_st = nullptr; }
        HIPC(hipMalloc(&d_st, bytes));
        HIPC(hipMemset(d_st, 0, bytes));
        HIPC(hipStreamSynchronize(nullptr));
        gm::set_corr_stamps(d_st);
        return GM_OK;
    }
    if (!d_st) return set_err(GM_ERR_INVALID_ARG, "stamps not armed");
    HIPC(hipStreamSynchronize(a->stream));
    HIPC(hipMemcpy(out, d_st, bytes, hipMemcpyDeviceToHost));
    gm::set_corr_stamps(nullptr);
    hipFree(d_st); d_st = nullptr;
    return GM_OK;
}

int gm_acq_manager_mode_for(size_t n) { return n == 0 ? 0 : (n <= 4 ? 1 : 2); }   // update_mode :50-56

int gm_acq_manager_pacing_and_list(int mode, uint32_t active, uint64_t* interval_ms, uint32_t* mask) {
    if (!interval_ms || !mask) return set_err(GM_ERR_INVALID_ARG, "null pointer");
    uint64_t interval; uint32_t size;                         // get_pacing_and_list :58-73
    switch (mode) {
        case 0: interval = 500; size = 32; break;
        case 1: interval = 1000; size = 8; break;
        case 2: interval = 2000; size = 5; break;
        default: return set_err(GM_ERR_INVALID_ARG, "mode must be 0..2");
    }
    uint32_t m = 0, taken = 0;
    for (uint32_t prn = 1; prn <= 32 && taken < size; ++prn)
        if (!((active >> (prn - 1)) & 1u)) { m |= 1u << (prn - 1); ++taken; }
    *interval_ms = interval; *mask = m;
    return GM_OK;
}

}  // extern "C"

// ====================================================================== ring mirror (struct gm_ring is defined above)

extern "C" {

int gm_ring_create(size_t buf_size, gm_ring** out) {
    if (!out) return set_err(GM_ERR_INVALID_ARG, "null out");
    *out = nullptr;
    if (!buf_size || (buf_size & (buf_size - 1))) return set_err(GM_ERR_INVALID_ARG, "Buffer size must be a power of two");
    if (int rc = ensure_device(g_device)) return rc;
    gm_ring* r = new gm_ring();
    r->device = g_device; r->size = buf_size; r->mask = buf_size - 1;
    r->slot_samples = buf_size < gm_ring::SLOT_SAMPLES_MAX ? buf_size : gm_ring::SLOT_SAMPLES_MAX;
    hipError_t e = hipMalloc(&r->d_buf, buf_size * 8);
    if (e == hipSuccess) e = hipMemset(r->d_buf, 0, buf_size * 8);
    // hipMemset of device memory returns before the fill has run and is ordered on the NULL stream only; the ring's writers work on
    // NON-BLOCKING streams, which that stream does not order: without this wait the fill can land on top of the first blocks written
    // (seen in round 6 as a tracking run that differed from its twin by a few zeroed samples, only in a long-lived process)
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) { hipFree(r->d_buf); delete r; return hip_fail(e, "hipMalloc(ring)"); }
    *out = r;
    return GM_OK;
}

int gm_ring_destroy(gm_ring* r) {
    if (!r) return GM_OK;
    hipSetDevice(r->device);
    if (r->copy_stream) { hipStreamSynchronize(r->copy_stream); hipStreamDestroy(r->copy_stream); }
    if (r->fe_stream && r->fe_stream != r->copy_stream) { hipStreamSynchronize(r->fe_stream); hipStreamDestroy(r->fe_stream); }
    for (hipEvent_t e : r->h2d_done) if (e) hipEventDestroy(e);
    if (r->ev_enq) hipEventDestroy(r->ev_enq);
    for (int i = 0; i < gm_ring::SLOTS; ++i) {
        if (r->staging[i]) hipHostFree(r->staging[i]);
        if (r->slot_done[i]) hipEventDestroy(r->slot_done[i]);
    }
    hipFree(r->d_buf);
    delete r;
    return GM_OK;
}

// write_samples without blocking the producer on the H2D copy: the samples are staged in pinned host
// memory, copied on the ring's own stream, and `head` advances (and the Condvar fires) only once they are in HBM, so a
// reader that sees the new head may launch kernels on the mirror without any stream dependency.  Single writer, like
// the reference ring; do not mix with gm_ring_write_samples without a gm_ring_flush in between.
int gm_ring_write_samples_async(gm_ring* r, const gm_c32* s, size_t n) {
    if (!r || (!s && n)) return set_err(GM_ERR_INVALID_ARG, "null pointer");
    if (n > r->size) return set_err(GM_ERR_OUT_OF_RANGE, "write larger than the ring");
    if (int rc = ensure_device(r->device)) return rc;
    if (int rc = ring_async_init(r)) return rc;
    const cf* src = reinterpret_cast<const cf*>(s);
    while (n) {
        const size_t chunk = n < r->slot_samples ? n : r->slot_samples;
        const int slot = int(r->slot_seq++ % gm_ring::SLOTS);
        if (int rc = ring_reclaim_slot(r, slot)) return rc;
        memcpy(r->staging[slot], src, chunk * 8);
        const size_t start = size_t(r->write_pos & r->mask);
        const size_t first = start + chunk <= r->size ? chunk : r->size - start;
        HIPC(hipMemcpyAsync(r->d_buf + start, r->staging[slot], first * 8, hipMemcpyHostToDevice, r->copy_stream));
        if (first < chunk)
            HIPC(hipMemcpyAsync(r->d_buf, r->staging[slot] + first, (chunk - first) * 8, hipMemcpyHostToDevice, r->copy_stream));
        HIPC(hipEventRecord(r->slot_done[slot], r->copy_stream));
        r->slot_used[slot] = true;
        r->write_pos += chunk;
        if (int rc = ring_enqueue_publish(r, slot, r->copy_stream)) return rc;
        src += chunk; n -= chunk;
    }
    return GM_OK;
}

int gm_ring_flush(gm_ring* r) {
    if (!r) return set_err(GM_ERR_INVALID_ARG, "null ring");
    if (!r->copy_stream) return GM_OK;
    if (int rc = ensure_device(r->device)) return rc;
    HIPC(hipStreamSynchronize(r->copy_stream));
    if (r->fe_stream) HIPC(hipStreamSynchronize(r->fe_stream));
    ring_refresh_head(r, true);
    return GM_OK;
}

// The Condvar wait of do_tracking::run (do_tracking.rs:392-406): sleep until head has reached `required_idx`
// (wrapping signed comparison like :393) or timeout_ms has passed.  *reached = 1 / 0.
int gm_ring_wait_head(gm_ring* r, uint64_t required_idx, uint32_t timeout_ms, int* reached) {
    if (!r) return set_err(GM_ERR_INVALID_ARG, "null ring");
    auto ok = [&] { return int64_t(r->head.load(std::memory_order_acquire) - required_idx) >= 0; };
    if (r->copy_stream) { if (int rc = ensure_device(r->device)) return rc; }
    // the asynchronous writer's chunks are retired by whoever looks (ring_refresh_head): look, then sleep on the Condvar in short
    // slices (a synchronous writer, or another reader's refresh, notifies it) until the head is there or the time is up
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
    bool got = false;
    for (;;) {
        ring_refresh_head(r);
        if ((got = ok())) break;
        const auto now = std::chrono::steady_clock::now();
        if (now >= deadline) break;
        std::unique_lock<std::mutex> g(r->notifier);
        auto slice = std::chrono::microseconds(100);
        if (deadline - now < slice) slice = std::chrono::duration_cast<std::chrono::microseconds>(deadline - now);
        if ((got = r->condvar.wait_for(g, slice, ok))) break;
    }
    if (reached) *reached = got ? 1 : 0;
    return GM_OK;
}

int gm_ring_write_samples(gm_ring* r, const gm_c32* s, size_t n) {
    if (!r || (!s && n)) return set_err(GM_ERR_INVALID_ARG, "null pointer");
    if (n > r->size) return set_err(GM_ERR_OUT_OF_RANGE, "write larger than the ring");
    if (int rc = ensure_device(r->device)) return rc;
    if (r->copy_stream) {       // chunks of the asynchronous writer still in flight: land and retire them first, so that `head` is the
        HIPC(hipStreamSynchronize(r->copy_stream));      // writer's position (a synchronous write placed at a stale head would overwrite them)
        if (r->fe_stream) HIPC(hipStreamSynchronize(r->fe_stream));
        ring_refresh_head(r, true);
    }
    const uint64_t cur = r->head.load(std::memory_order_relaxed);
    const size_t start = size_t(cur & r->mask);
    if (start + n <= r->size) {
        HIPC(hipMemcpy(r->d_buf + start, s, n * 8, hipMemcpyHostToDevice));
    } else {
        const size_t first = r->size - start;
        HIPC(hipMemcpy(r->d_buf + start, s, first * 8, hipMemcpyHostToDevice));
        HIPC(hipMemcpy(r->d_buf, s + first, (n - first) * 8, hipMemcpyHostToDevice));
    }
    r->write_pos = cur + n;
    ring_publish(r, cur + n);
    return GM_OK;
}

int gm_ring_get_head(gm_ring* r, uint64_t* head) {
    if (!r || !head) return set_err(GM_ERR_INVALID_ARG, "null pointer");
    if (r->copy_stream) { if (int rc = ensure_device(r->device)) return rc; ring_refresh_head(r); }
    *head = r->head.load(std::memory_order_acquire);
    return GM_OK;
}

int gm_ring_get_enqueued_head(gm_ring* r, uint64_t* head) {
    if (!r || !head) return set_err(GM_ERR_INVALID_ARG, "null pointer");
    uint64_t h = r->head.load(std::memory_order_acquire);
    {
        std::lock_guard<std::mutex> g(r->enq_mu);
        if (r->ev_enq_armed && int64_t(r->enqueued - h) > 0) h = r->enqueued;
    }
    *head = h;
    return GM_OK;
}

int gm_ring_copy_to_slice(gm_ring* r, uint64_t start, gm_c32* dest, size_t n) {
    if (!r || (!dest && n)) return set_err(GM_ERR_INVALID_ARG, "null pointer");
    if (n > r->size) return set_err(GM_ERR_OUT_OF_RANGE, "slice larger than the ring");
    if (int rc = ensure_device(r->device)) return rc;
    const size_t ps = size_t(start & r->mask);
    if (ps + n <= r->size) {
        HIPC(hipMemcpy(dest, r->d_buf + ps, n * 8, hipMemcpyDeviceToHost));
    } else {
        const size_t first = r->size - ps;
        HIPC(hipMemcpy(dest, r->d_buf + ps, first * 8, hipMemcpyDeviceToHost));
        HIPC(hipMemcpy(dest + first, r->d_buf, (n - first) * 8, hipMemcpyDeviceToHost));
    }
    return GM_OK;
}

}  // extern "C"

// ====================================================================== tracking handle
struct gm_trk {
    int device = -1;
    gm_trk_cfg cfg{};
    gm::TrkDevCfg dc{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t C = 0;
    int slices = 1;
    std::vector<int8_t> h_codes;
    int8_t* d_codes = nullptr;
    gm_trk_state* d_states = nullptr;
    float* d_partials = nullptr;
    uint8_t* d_ready = nullptr;
    cf* d_scratch = nullptr; size_t scratch_cap = 0;
    float* d_terms = nullptr; size_t terms_cap = 0;   // strict_sum_order: [C][2 * arms][terms_cap] per-sample products of one epoch
    // results of a call of e passes, n = e * C entries: ONE device block laid out for that n as [outs n | processed n | lost n |
    // lost_prn n] (trk_reserve_epochs), so that outs + processed + lost leave in ONE device-to-host copy; d_outs .. d_lostprn point into it
    uint8_t* d_res = nullptr; size_t res_cap = 0, res_n = 0;
    gm_trk_out* d_outs = nullptr; uint8_t *d_proc = nullptr, *d_lost = nullptr, *d_lostprn = nullptr;
    uint8_t* h_res = nullptr; size_t h_cap = 0;    // pinned landing area of gm_trk_update_all: [outs | processed | lost] of the call's passes
    uint32_t epochs_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing = false; uint32_t timed_launches = 0;
    // persistent multi-epoch kernel: G workgroups per channel, granule exchange buffer, launch counter
    int G = 1;
    bool packed = false;                // the persistent grid's packed layout (trk_kernels.hip): G workgroups per channel dealt over the XCDs as one run
    unsigned long long* d_xchg = nullptr;
    int* d_error = nullptr;             // host-pinned, device-visible: written by the kernel only when an exchange times out
    int* d_error_dev = nullptr;         // its device-memory twin, read by every later launch (trk_persistent_kernel)
    uint32_t launch_seq = 0;
    long long* d_stamps = nullptr; uint32_t stamps_cap = 0;   // diagnostic phase stamps (gm_trk_debug_stamps)
    // gm_trk_update_all_async: results of call k land in pinned slot k % TICKETS ([outs | processed | lost] of that call's epochs),
    // `done` recorded behind the copies; gm_trk_collect hands them over and frees the slot
    static constexpr int TICKETS = 8;
    struct Ticket { uint8_t* h = nullptr; hipEvent_t done = nullptr; bool in_flight = false; uint32_t epochs = 0; uint64_t id = 0; } tk[TICKETS];
    uint8_t* tk_block = nullptr;          // ONE pinned allocation cut into the TICKETS slots (a pinned allocation costs ~1 ms: not per slot, not in a receiver's loop)
    size_t tk_cap = 0;                    // bytes per slot
    uint64_t next_ticket = 1;
};

static int trk_check_error(gm_trk* t);
static int trk_reserve_tickets(gm_trk* t, size_t bytes);
// strict_sum_order: room for `samples` per-sample products per sum and channel
static int trk_reserve_terms(gm_trk* t, size_t samples) {
    samples = (samples + 3) & ~size_t(3);       // rows of the streams stay 16-byte aligned
    if (!t->dc.strict_sum_order || samples <= t->terms_cap) return GM_OK;
    HIPC(hipStreamSynchronize(t->stream));
    hipFree(t->d_terms); t->d_terms = nullptr; t->terms_cap = 0;
    HIPC(hipMalloc(&t->d_terms, size_t(t->C) * 2 * size_t(t->dc.n_arms) * samples * sizeof(float)));
    t->terms_cap = samples;
    return GM_OK;
}
static int trk_reserve_epochs(gm_trk* t, uint32_t e) {
    const size_t n = size_t(e) * t->C, bytes = (n * (sizeof(gm_trk_out) + 3) + 3) & ~size_t(3);
    if (bytes > t->res_cap) {
        HIPC(hipStreamSynchronize(t->stream));      // nothing in flight still writes the block about to be replaced
        hipFree(t->d_res); t->d_res = nullptr; t->res_cap = 0; t->res_n = 0;
        HIPC(hipMalloc(&t->d_res, bytes));
        t->res_cap = bytes;
    }
    if (n * (sizeof(gm_trk_out) + 2) > t->h_cap) {
        HIPC(hipStreamSynchronize(t->stream));
        if (t->h_res) { hipHostFree(t->h_res); t->h_res = nullptr; t->h_cap = 0; }
        HIPC(hipHostMalloc(reinterpret_cast<void**>(&t->h_res), n * (sizeof(gm_trk_out) + 2), hipHostMallocDefault));
        t->h_cap = n * (sizeof(gm_trk_out) + 2);
    }
    if (n != t->res_n) {
        // a new layout: with three arms the persistent kernel writes the six live sums only, ive..qvl must read 0 — and bytes that were
        // flags under the previous layout may now lie inside `outs`.  Ordered on the handle's stream (a non-blocking stream does not
        // synchronise with the NULL stream a plain hipMemset runs on).  A loop that keeps its pass count pays this once.
        HIPC(hipMemsetAsync(t->d_res, 0, bytes, t->stream));
        t->res_n = n;
    }
    t->d_outs = reinterpret_cast<gm_trk_out*>(t->d_res);
    t->d_proc = t->d_res + n * sizeof(gm_trk_out);
    t->d_lost = t->d_proc + n;
    t->d_lostprn = t->d_lost + n;
    if (e > t->epochs_cap) t->epochs_cap = e;
    return GM_OK;
}

extern "C" {

int gm_loop_filter_new(float bw, float zeta, float gain, float* tau1, float* tau2) {
    if (!tau1 || !tau2) return set_err(GM_ERR_INVALID_ARG, "null pointer");
    loop_filter_new(bw, zeta, gain, tau1, tau2);
    return GM_OK;
}
float gm_loop_filter_update(float tau1, float tau2, float d_err, float err, float dt) {   // :68-70
    return d_err * (dt / tau1) + (d_err - err) * (tau2 / tau1);
}

int gm_trk_destroy(gm_trk* t) {
    if (!t) return GM_OK;
    if (t->device >= 0) hipSetDevice(t->device);
    hipFree(t->d_codes); hipFree(t->d_states); hipFree(t->d_partials); hipFree(t->d_ready); hipFree(t->d_scratch); hipFree(t->d_terms);
    hipFree(t->d_res);
    if (t->h_res) hipHostFree(t->h_res);
    for (auto& k : t->tk) if (k.done) hipEventDestroy(k.done);
    if (t->tk_block) hipHostFree(t->tk_block);
    hipFree(t->d_xchg); if (t->d_error) hipHostFree(t->d_error); hipFree(t->d_error_dev); hipFree(t->d_stamps);
    if (t->ev0) hipEventDestroy(t->ev0);
    if (t->ev1) hipEventDestroy(t->ev1);
    if (t->own_stream && t->stream) hipStreamDestroy(t->stream);
    delete t;
    return GM_OK;
}

int gm_trk_create(const gm_trk_cfg* cfg, gm_trk** out) {
    if (!cfg || !out) return set_err(GM_ERR_INVALID_ARG, "null cfg/out");
    *out = nullptr;
    if (!(cfg->fs > 0) || !cfg->n_channels) return set_err(GM_ERR_INVALID_ARG, "fs and n_channels are required");
    const uint32_t arms = cfg->n_arms ? cfg->n_arms : 3;
    if (arms != 3 && arms != 5) return set_err(GM_ERR_INVALID_ARG, "n_arms must be 3 or 5");
    if (cfg->codes && (!cfg->n_codes || !cfg->code_len)) return set_err(GM_ERR_INVALID_ARG, "n_codes/code_len required");
    {   // the kernels replace `% code_len` by one conditional subtraction: arm spacings must stay below a code period
        const float len = cfg->codes ? float(cfg->code_len) : 1023.0f;
        if (cfg->early_late_space >= len || cfg->very_early_late_space >= len || cfg->early_late_space < 0 ||
            cfg->very_early_late_space < 0)
            return set_err(GM_ERR_INVALID_ARG, "arm spacing must be in [0, code_len)");
    }
    if (int rc = ensure_device(g_device)) return rc;
    gm_trk* t = new gm_trk();
    t->device = g_device; t->cfg = *cfg; t->C = cfg->n_channels;
    gm::TrkDevCfg& d = t->dc;
    d.fs = cfg->fs; d.n_channels = int(cfg->n_channels); d.n_arms = int(arms);
    d.el_space = cfg->early_late_space > 0 ? cfg->early_late_space : 0.5f;
    d.vel_space = cfg->very_early_late_space > 0 ? cfg->very_early_late_space : 1.0f;
    d.code_index_mode = cfg->code_index_mode; d.boc11 = cfg->boc11;
    d.strict_libm = cfg->strict_libm ? 1 : 0;
    d.strict_sum_order = cfg->strict_sum_order ? 1 : 0;
    d.gps_ca = cfg->codes ? 0 : 1;
    d.code_len = cfg->codes ? int(cfg->code_len) : 1023;
    d.code_len_f = float(d.code_len);
    d.n_codes = cfg->codes ? int(cfg->n_codes) : 32;
    d.lock_threshold = cfg->lock_threshold > 0 ? cfg->lock_threshold : 15.0f;
    d.max_lost_epochs = cfg->max_lost_epochs ? cfg->max_lost_epochs : 20u;
    loop_filter_new(cfg->pll_bw > 0 ? cfg->pll_bw : 25.0f, cfg->pll_zeta > 0 ? cfg->pll_zeta : 0.7f,
                    cfg->pll_gain > 0 ? cfg->pll_gain : 0.25f, &d.pll_tau1, &d.pll_tau2);
    loop_filter_new(cfg->dll_bw > 0 ? cfg->dll_bw : 2.0f, cfg->dll_zeta > 0 ? cfg->dll_zeta : 0.7f,
                    cfg->dll_gain > 0 ? cfg->dll_gain : 1.0f, &d.dll_tau1, &d.dll_tau2);
    d.pll_dt = cfg->pll_dt > 0 ? cfg->pll_dt : 0.001f;
    d.dll_dt = cfg->dll_dt > 0 ? cfg->dll_dt : 0.001f;
    d.nominal_code_rate = cfg->nominal_code_rate > 0 ? cfg->nominal_code_rate : CA_RATE;
    gm::fill_trk_derived(d);
    if (cfg->codes) t->h_codes.assign(cfg->codes, cfg->codes + size_t(cfg->n_codes) * cfg->code_len);
    else t->h_codes.assign(&ca_table().rows[0][0], &ca_table().rows[0][0] + 32 * 1023);

    // slices per channel: fill the chip (>= ~1000 workgroups) without going under 256 samples each
    const size_t n_nom = num_samples_per_code(d.nominal_code_rate, d.fs, d.code_len_f);
    size_t target = (1024 + t->C - 1) / t->C;
    if (target < 1) target = 1;
    size_t per = ((n_nom + target - 1) / target + 255) / 256 * 256;
    if (per < 256) per = 256;
    size_t s = (n_nom + per - 1) / per;
    t->slices = int(s < 1 ? 1 : (s > 256 ? 256 : s));

    auto fail = [&](int code) { gm_trk_destroy(t); return code; };
#define HIPT(expr)                                                     \
    do {                                                               \
        hipError_t _e = (expr);                                        \
        if (_e != hipSuccess) return fail(hip_fail(_e, #expr));        \
    } while (0)
    {   // the tracking loop is the receiver's latency path (one short launch per block of samples): its own stream is the device's
        // most urgent one, so a launch is not queued behind a front-end block or an acquisition dwell in flight on another stream
        // (measured: 0.45 ms of waiting per call behind the front-end kernel of the same block, tools/trk_call_time.py)
        int least = 0, greatest = 0;
        HIPT(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPT(hipStreamCreateWithPriority(&t->stream, hipStreamNonBlocking, greatest));
    }
    t->own_stream = true;
    HIPT(hipMalloc(&t->d_codes, t->h_codes.size()));
    HIPT(hipMemcpy(t->d_codes, t->h_codes.data(), t->h_codes.size(), hipMemcpyHostToDevice));
    HIPT(hipMalloc(&t->d_states, sizeof(gm_trk_state) * t->C));
    std::vector<gm_trk_state> init(t->C);
    for (auto& st : init) {   // TrackingChannel::new :118-146
        memset(&st, 0, sizeof(st));
        st.num_samples_per_code = n_nom;
        st.code_rate = d.nominal_code_rate;
    }
    HIPT(hipMemcpy(t->d_states, init.data(), sizeof(gm_trk_state) * t->C, hipMemcpyHostToDevice));
    HIPT(hipMalloc(&t->d_partials, sizeof(float) * t->C * size_t(t->slices) * 10));
    HIPT(hipMalloc(&t->d_ready, t->C));
    HIPT(hipMemsetAsync(t->d_ready, 0, t->C, t->stream));      // (the handle's stream is non-blocking: a NULL-stream memset would not be ordered with it)
    HIPT(hipEventCreate(&t->ev0));
    HIPT(hipEventCreate(&t->ev1));
    {   // workgroups per channel for the persistent kernel: all n_channels*G must be co-resident (one per CU)
        hipDeviceProp_t prop;
        HIPT(hipGetDeviceProperties(&prop, t->device));
        const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
        const int nv = 2 * d.n_arms;
        int g = 1;
        const int per_cu = gm::trk_persistent_blocks_per_cu(d);      // occupancy of this instantiation, capped at the design's 2
        const size_t slots = size_t(gm::trk_persistent_slots(t->C));   // the grid is slots * G workgroups (empty ones leave at once)
        // the most workgroups per channel the chip holds at once (any count, not only powers of two: 36 channels take 12 each,
        // 480 of the 512 places, where 8 left a third of the chip idle and two-workgroup CUs beside one-workgroup CUs);
        // 17..31 fall back to 16, the width of the DPP totals path
        {
            // share_device: a quarter of the places (gm_trk_cfg: the other stages' kernels run beside a tracking launch)
            const size_t fit = (size_t(cus) * per_cu) / (cfg->share_device ? 4 : 1) / slots;
            g = int(fit < 1 ? 1 : fit > 32 ? 32 : fit);
            while (g > 1 && g * nv > 256) --g;
            if (g > 16 && g < 32) g = 16;
        }
        {   // Packed layout: a channel count that does not fill the eight XCDs evenly (36 -> 5, 5, 5, 5, 4, 4, 4, 4 slots of 12
            // workgroups: 432 of 512 places, and the launch lasts as long as the 5-channel XCDs) is dealt as ONE run of C * G
            // workgroups, 14 each at 36 channels (504 places); a channel at the edge of an XCD's piece exchanges through the fabric.
            // Only where correlation fills the epoch (>= 8 samples per lane): a latency-chain launch wants its channels on one XCD.
            const size_t places = (size_t(cus) * per_cu) / (cfg->share_device ? 4 : 1);
            int gp = int(places / t->C < 16 ? places / t->C : 16);
            while (gp > 1 && gp * nv > 256) --gp;
            { const int fp = gm::diag_int("GM_TRK_PACKED_G", 0); if (fp >= 1 && fp < gp) gp = fp; }      // diagnostic: fewer workgroups per channel in the packed layout
            const float nn = roundf(d.fs / (d.nominal_code_rate / d.code_len_f));
            const bool shaped = nn > 0 && gp >= 1 && nn / float(gp) / 512.0f >= 8.0f;
            if ((t->C % 8) != 0 && gp > g && shaped && gm::diag_int("GM_TRK_PACKED", 1) != 0) { g = gp; t->packed = true; }
        }
        {   // diagnostic override (GM_DIAGNOSTICS=1 only; must keep n_channels * G resident)
            const int f = gm::diag_int("GM_TRK_G", 0);
            if (f >= 1 && f <= 32 && size_t(f) * slots <= size_t(cus) * per_cu && f * nv <= 256) { g = f; t->packed = false; }
        }
        t->G = g;
        const size_t xb = (size_t(2) * t->C * gm::trk_persistent_granule_stride(g) * nv + size_t(t->C) * g) * sizeof(unsigned long long);   // partials + XCC_ID granules
        HIPT(hipMalloc(&t->d_xchg, xb));
        HIPT(hipMemsetAsync(t->d_xchg, 0, xb, t->stream));
        HIPT(hipHostMalloc(reinterpret_cast<void**>(&t->d_error), sizeof(int), hipHostMallocDefault));   // read on the host after a stream sync: no copy
        *t->d_error = 0;
        HIPT(hipMalloc(&t->d_error_dev, sizeof(int)));
        HIPT(hipMemsetAsync(t->d_error_dev, 0, sizeof(int), t->stream));
    }
    if (int rc = trk_reserve_epochs(t, 1)) return fail(rc);
    if (int rc = trk_reserve_tickets(t, size_t(128) * t->C * (sizeof(gm_trk_out) + 2))) return fail(rc);     // gm_trk_update_all_async's result slots
    HIPT(hipStreamSynchronize(t->stream));      // the clears above have landed before any caller-supplied stream (gm_trk_set_stream) can run a kernel
#undef HIPT
    *out = t;
    return GM_OK;
}

int gm_trk_set_stream(gm_trk* t, void* s) {
    if (!t) return set_err(GM_ERR_INVALID_ARG, "null handle");
    if (t->own_stream && t->stream) { hipStreamSynchronize(t->stream); hipStreamDestroy(t->stream); }
    t->stream = reinterpret_cast<hipStream_t>(s);
    t->own_stream = false;
    return GM_OK;
}

int gm_trk_get_state(gm_trk* t, uint32_t ch, gm_trk_state* out) {
    if (!t || !out || ch >= t->C) return set_err(GM_ERR_INVALID_ARG, "bad channel");
    if (int rc = ensure_device(t->device)) return rc;
    HIPC(hipStreamSynchronize(t->stream));
    HIPC(hipMemcpy(out, t->d_states + ch, sizeof(*out), hipMemcpyDeviceToHost));
    return GM_OK;
}

int gm_trk_set_state(gm_trk* t, uint32_t ch, const gm_trk_state* in) {
    if (!t || !in || ch >= t->C) return set_err(GM_ERR_INVALID_ARG, "bad channel");
    if (int rc = ensure_device(t->device)) return rc;
    HIPC(hipStreamSynchronize(t->stream));
    HIPC(hipMemcpy(t->d_states + ch, in, sizeof(*in), hipMemcpyHostToDevice));
    return GM_OK;
}

int gm_trk_get_states(gm_trk* t, gm_trk_state* out) {
    if (!t || !out) return set_err(GM_ERR_INVALID_ARG, "null pointer");
    if (int rc = ensure_device(t->device)) return rc;
    HIPC(hipStreamSynchronize(t->stream));
    HIPC(hipMemcpy(out, t->d_states, size_t(t->C) * sizeof(*out), hipMemcpyDeviceToHost));
    return GM_OK;
}

int gm_trk_set_states(gm_trk* t, const gm_trk_state* in, const uint8_t* which) {
    if (!t || !in) return set_err(GM_ERR_INVALID_ARG, "null pointer");
    if (int rc = ensure_device(t->device)) return rc;
    HIPC(hipStreamSynchronize(t->stream));
    if (!which) {
        HIPC(hipMemcpy(t->d_states, in, size_t(t->C) * sizeof(*in), hipMemcpyHostToDevice));
        return GM_OK;
    }
    for (uint32_t c = 0; c < t->C; ) {          // runs of flagged channels, one copy each
        if (!which[c]) { ++c; continue; }
        uint32_t e = c;
        while (e < t->C && which[e]) ++e;
        HIPC(hipMemcpy(t->d_states + c, in + c, size_t(e - c) * sizeof(*in), hipMemcpyHostToDevice));
        c = e;
    }
    return GM_OK;
}

int gm_trk_start(gm_trk* t, uint32_t ch, const gm_acq_result* r) {   // :148-154
    if (!t || !r || ch >= t->C) return set_err(GM_ERR_INVALID_ARG, "bad channel");
    gm_trk_state s;
    if (int rc = gm_trk_get_state(t, ch, &s)) return rc;
    s.prn = r->prn;
    s.carrier_freq = r->carrier_freq;
    s.code_phase = r->code_phase_chips;
    s.next_sample_index = r->sample_global_index;
    s.active = 1;
    // FIXED: sample_global_index already points at the code start, so the replica starts at chip 0 (the reference
    // passes the acquisition delay in chips AND the aligned sample index)
    if (t->cfg.code_index_mode == GM_CODE_INDEX_FIXED) s.code_phase = 0.0f;
    if (t->cfg.code_index_mode == GM_CODE_INDEX_FIXED && s.code_rate == 0.0f)
        s.code_rate = t->dc.nominal_code_rate;   // FIXED: undo reset()'s code_rate = 0 (reference bug)
    return gm_trk_set_state(t, ch, &s);
}

int gm_trk_reset(gm_trk* t, uint32_t ch) {   // :311-327
    if (!t || ch >= t->C) return set_err(GM_ERR_INVALID_ARG, "bad channel");
    gm_trk_state s;
    if (int rc = gm_trk_get_state(t, ch, &s)) return rc;
    const uint64_t n = s.num_samples_per_code;
    memset(&s, 0, sizeof(s));
    s.num_samples_per_code = n;   // reset() leaves num_samples_per_code untouched
    return gm_trk_set_state(t, ch, &s);
}

int gm_trk_get_ca_chip(gm_trk* t, uint32_t ch, float phase, float* chip) {   // :274-277
    if (!t || !chip || ch >= t->C) return set_err(GM_ERR_INVALID_ARG, "bad channel");
    gm_trk_state s;
    if (int rc = gm_trk_get_state(t, ch, &s)) return rc;
    const int len = t->dc.code_len;
    int row, idx;
    const float f = floorf(phase);
    if (t->dc.code_index_mode == GM_CODE_INDEX_FAITHFUL) {
        row = t->dc.gps_ca ? int(s.prn) : int(s.prn) - 1;
        if (!(f > 0.0f)) idx = 0;
        else if (f >= 18446744073709551616.0f) idx = int(18446744073709551615ull % (unsigned long long)len);
        else idx = int((unsigned long long)f % (unsigned long long)len);
    } else {
        row = int(s.prn) - 1;
        long li = long(f) % len;
        idx = int(li < 0 ? li + len : li);
    }
    if (row < 0 || row >= t->dc.n_codes) return set_err(GM_ERR_OUT_OF_RANGE, "code table row out of bounds (the reference panics)");
    *chip = float(t->h_codes[size_t(row) * len + idx]);
    return GM_OK;
}

static int trk_unit(gm_trk* t, uint32_t ch, const gm_c32* samples, size_t n, int mode, gm_trk_out* out, uint8_t* lost,
                    uint8_t* lost_prn) {
    if (!t || !samples || !out || ch >= t->C) return set_err(GM_ERR_INVALID_ARG, "bad argument");
    if (int rc = ensure_device(t->device)) return rc;
    gm_trk_state s;
    if (int rc = gm_trk_get_state(t, ch, &s)) return rc;
    if (n < s.num_samples_per_code || s.num_samples_per_code == 0)
        return set_err(GM_ERR_OUT_OF_RANGE, "fewer samples than num_samples_per_code (the reference panics)");
    const int row = t->dc.gps_ca ? (t->dc.code_index_mode == GM_CODE_INDEX_FAITHFUL ? int(s.prn) : int(s.prn) - 1)
                                 : int(s.prn) - 1;
    if (row < 0 || row >= t->dc.n_codes) return set_err(GM_ERR_OUT_OF_RANGE, "code table row out of bounds (the reference panics)");
    const size_t need = s.num_samples_per_code;
    if (need > t->scratch_cap) {
        hipFree(t->d_scratch); t->d_scratch = nullptr; t->scratch_cap = 0;
        HIPC(hipMalloc(&t->d_scratch, need * 8 * 2));
        t->scratch_cap = need * 2;
    }
    if (int rc = trk_reserve_terms(t, need)) return rc;
    if (int rc = trk_reserve_epochs(t, 1)) return rc;          // the result block laid out for one pass
    HIPC(hipMemcpyAsync(t->d_scratch, samples, need * 8, hipMemcpyHostToDevice, t->stream));
    // the unit entries run regardless of ChannelState (the reference's early_late_correlation/do_work do not test it)
    const uint8_t was_active = s.active;
    if (!was_active) { s.active = 1; HIPC(hipMemcpyAsync(t->d_states + ch, &s, sizeof(s), hipMemcpyHostToDevice, t->stream)); HIPC(hipStreamSynchronize(t->stream)); }
    gm::TrkSrc src;
    src.base = t->d_scratch; src.mask = ~0ull; src.head = 0; src.linear = 1; src.only_channel = int(ch);
    gm::launch_trk_epoch(t->stream, t->dc, t->d_codes, t->d_states, src, t->slices, t->d_partials, t->d_ready, mode,
                         t->d_outs, t->d_proc, t->d_lost, t->d_lostprn, t->d_terms, t->terms_cap, t->d_error);
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(t->stream));
    HIPC(hipMemcpy(out, t->d_outs + ch, sizeof(*out), hipMemcpyDeviceToHost));
    uint8_t l = 0, lp = 0;
    HIPC(hipMemcpy(&l, t->d_lost + ch, 1, hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(&lp, t->d_lostprn + ch, 1, hipMemcpyDeviceToHost));
    if (lost) *lost = l;
    if (lost_prn) *lost_prn = lp;
    if (!was_active && !l) {   // restore the caller's ChannelState
        gm_trk_state s2;
        HIPC(hipMemcpy(&s2, t->d_states + ch, sizeof(s2), hipMemcpyDeviceToHost));
        s2.active = 0;
        HIPC(hipMemcpy(t->d_states + ch, &s2, sizeof(s2), hipMemcpyHostToDevice));
    }
    return GM_OK;
}

int gm_trk_correlate(gm_trk* t, uint32_t ch, const gm_c32* samples, size_t n, gm_trk_out* out) {
    return trk_unit(t, ch, samples, n, gm::TRK_MODE_CORRELATE, out, nullptr, nullptr);
}
int gm_trk_do_work(gm_trk* t, uint32_t ch, const gm_c32* samples, size_t n, gm_trk_out* out, uint8_t* lost, uint8_t* lost_prn) {
    return trk_unit(t, ch, samples, n, gm::TRK_MODE_DO_WORK, out, lost, lost_prn);
}

// The persistent kernel's workgroups wait for one another, so its whole grid must be resident at once: ONE persistent
// launch per device at a time.  Launches of different handles (different streams) are chained through an event per device:
// each waits, on the GPU, for the previous persistent launch of that device and records the event behind itself — no host
// blocking, and two tracking managers can no longer strand half of each other's grids until the 0.2 s time-out.
namespace {
struct PersistChain { std::mutex mu; hipEvent_t ev = nullptr; bool armed = false; };
PersistChain g_persist_chain[16];
}

// `epochs` passes of process_channels on the handle's stream; the data gate of every pass (:170-172) uses `head`
static int trk_launch_all(gm_trk* t, gm_ring* ring, uint32_t epochs, uint64_t head) {
    if (int rc = trk_reserve_epochs(t, epochs)) return rc;
    if (t->dc.strict_sum_order) {
        // the reference's sequential sums: three launches per pass (products, one serial wave per channel, scalar epilogue)
        // instead of the persistent kernel; the data gate of every pass reads the head as of this call, like the persistent form
        const float nn = roundf(t->dc.fs / (t->dc.nominal_code_rate / t->dc.code_len_f));
        if (int rc = trk_reserve_terms(t, (nn > 0 ? size_t(nn * 1.01f) : 0) + 64)) return rc;
        // the strict launches (256 threads, 40-48 KB of LDS each) must not take CU slots away from another handle's persistent
        // grid on this device either: same chain — wait for the previous launch of the device, record behind the last one
        PersistChain& schain = g_persist_chain[t->device & 15];
        std::lock_guard<std::mutex> schain_lock(schain.mu);
        if (!schain.ev) HIPC(hipEventCreateWithFlags(&schain.ev, hipEventDisableTiming));
        if (schain.armed) HIPC(hipStreamWaitEvent(t->stream, schain.ev, 0));
        if (t->timing) HIPC(hipEventRecord(t->ev0, t->stream));
        gm::TrkSrc src;
        src.base = ring->d_buf; src.mask = ring->mask; src.head = head; src.linear = 0; src.only_channel = -1;
        for (uint32_t e = 0; e < epochs; ++e) {
            const size_t o = size_t(e) * t->C;
            gm::launch_trk_epoch(t->stream, t->dc, t->d_codes, t->d_states, src, t->slices, t->d_partials, t->d_ready, gm::TRK_MODE_DO_WORK,
                                 t->d_outs + o, t->d_proc + o, t->d_lost + o, t->d_lostprn + o, t->d_terms, t->terms_cap, t->d_error);
        }
        if (t->timing) { HIPC(hipEventRecord(t->ev1, t->stream)); t->timed_launches = epochs; }
        HIPC(hipEventRecord(schain.ev, t->stream));
        schain.armed = true;
        HIPC(hipGetLastError());
        return GM_OK;
    }
    PersistChain& chain = g_persist_chain[t->device & 15];
    std::lock_guard<std::mutex> chain_lock(chain.mu);
    if (!chain.ev) HIPC(hipEventCreateWithFlags(&chain.ev, hipEventDisableTiming));
    if (chain.armed) HIPC(hipStreamWaitEvent(t->stream, chain.ev, 0));
    if (t->timing) HIPC(hipEventRecord(t->ev0, t->stream));
    // one persistent launch per <= 4095 epochs (the epoch index lives in the low 12 bits of the granule tag)
    for (uint32_t e0 = 0; e0 < epochs; e0 += 4095) {
        const uint32_t ne = epochs - e0 < 4095 ? epochs - e0 : 4095;
        const size_t o = size_t(e0) * t->C;
        t->launch_seq = (t->launch_seq + 1) & 0xfffffu;
        if (t->launch_seq == 0) t->launch_seq = 1;
        gm::launch_trk_persistent(t->stream, t->dc, t->d_codes, t->d_states, ring->d_buf, ring->mask, head, t->G, t->packed ? 1 : 0,
                                  int(ne), t->launch_seq << 12, t->d_xchg, t->d_outs + o, t->d_proc + o, t->d_lost + o,
                                  t->d_lostprn + o, t->d_error, t->d_error_dev, (t->d_stamps && e0 == 0 && ne <= t->stamps_cap) ? t->d_stamps : nullptr);
    }
    if (t->timing) { HIPC(hipEventRecord(t->ev1, t->stream)); t->timed_launches = epochs; }
    HIPC(hipEventRecord(chain.ev, t->stream));
    chain.armed = true;
    HIPC(hipGetLastError());
    return GM_OK;
}

int gm_trk_update_all_dev(gm_trk* t, gm_ring* ring, uint32_t epochs) {
    if (!t || !ring || !epochs) return set_err(GM_ERR_INVALID_ARG, "bad argument");
    if (t->device != ring->device) return set_err(GM_ERR_INVALID_ARG, "ring lives on another device");
    if (int rc = ensure_device(t->device)) return rc;
    ring_refresh_head(ring);
    return trk_launch_all(t, ring, epochs, ring->head.load(std::memory_order_acquire));      // the PUBLISHED head: the samples are in HBM
}

// the pinned result slots of gm_trk_update_all_async: one block, at least `bytes` per slot (grown only while no ticket is in flight)
static int trk_reserve_tickets(gm_trk* t, size_t bytes) {
    if (bytes <= t->tk_cap) return GM_OK;
    for (const auto& k : t->tk)
        if (k.in_flight) return set_err(GM_ERR_OUT_OF_RANGE, "a larger max_epochs than the result slots hold: collect the tickets in flight first");
    if (t->tk_block) { hipHostFree(t->tk_block); t->tk_block = nullptr; t->tk_cap = 0; }
    const size_t per = (bytes + 255) & ~size_t(255);
    HIPC(hipHostMalloc(reinterpret_cast<void**>(&t->tk_block), per * gm_trk::TICKETS, hipHostMallocDefault));
    t->tk_cap = per;
    for (int i = 0; i < gm_trk::TICKETS; ++i) {
        t->tk[i].h = t->tk_block + size_t(i) * per;
        if (!t->tk[i].done) HIPC(hipEventCreateWithFlags(&t->tk[i].done, hipEventDisableTiming));
    }
    return GM_OK;
}

static uint32_t trk_epochs_done(const uint8_t* proc, uint32_t max_epochs, uint32_t C) {
    uint32_t done = 0;
    for (uint32_t e = 0; e < max_epochs; ++e) {
        bool any = false;
        for (uint32_t c = 0; c < C; ++c) any |= proc[size_t(e) * C + c] != 0;
        if (any) done = e + 1;
    }
    return done;
}

// The same passes without a host wait: ordered on the DEVICE behind whatever the ring's writer has enqueued so far
// — an event on the ring's copy stream, the counterpart of the Condvar wait at do_tracking.rs:392-406 — with the data gate on that
// enqueued head; results into a pinned slot, handed over by gm_trk_collect.
int gm_trk_update_all_async(gm_trk* t, gm_ring* ring, uint32_t max_epochs, uint64_t* ticket) {
    if (!t || !ring || !max_epochs || !ticket) return set_err(GM_ERR_INVALID_ARG, "bad argument");
    if (t->device != ring->device) return set_err(GM_ERR_INVALID_ARG, "ring lives on another device");
    *ticket = 0;
    gm_trk::Ticket& k = t->tk[t->next_ticket % gm_trk::TICKETS];
    if (k.in_flight) return set_err(GM_ERR_OUT_OF_RANGE, "all result slots are in flight: gm_trk_collect the oldest ticket first");
    if (int rc = ensure_device(t->device)) return rc;
    // slot: [outs n | processed n | lost n | pad to 8 | states C] — the states as they stand behind THIS call's passes
    const size_t n = size_t(max_epochs) * t->C, bytes = n * (sizeof(gm_trk_out) + 2);
    const size_t st_off = (bytes + 7) & ~size_t(7), st_bytes = size_t(t->C) * sizeof(gm_trk_state);
    if (int rc = trk_reserve_tickets(t, st_off + st_bytes)) return rc;
    static const int trace_slow = gm::diag_int("GM_TRK_TRACE_SLOW", 0);      // diagnostic: which runtime call of this entry takes milliseconds
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
    const auto t0 = now();
    ring_refresh_head(ring);
    const auto t1 = now();
    uint64_t head = ring->head.load(std::memory_order_acquire);
    {
        std::lock_guard<std::mutex> g(ring->enq_mu);
        if (ring->ev_enq_armed) {
            HIPC(hipStreamWaitEvent(t->stream, ring->ev_enq, 0));
            if (int64_t(ring->enqueued - head) > 0) head = ring->enqueued;
        }
    }
    const auto t2 = now();
    if (int rc = trk_launch_all(t, ring, max_epochs, head)) return rc;
    const auto t3 = now();
    gm::launch_trk_results_to_host(t->stream, t->d_res, k.h, bytes, t->d_states, k.h + st_off, st_bytes);   // -> the pinned slot (trk_kernels.hip)
    const auto t4 = now();
    HIPC(hipEventRecord(k.done, t->stream));
    const auto t5 = now();
    if (trace_slow && us(t0, t5) > 1000)
        fprintf(stderr, "gm_trk_update_all_async slow: refresh %ld us, wait-event %ld, launch %ld, copy %ld, record %ld\n", us(t0, t1), us(t1, t2), us(t2, t3), us(t3, t4), us(t4, t5));
    k.in_flight = true; k.epochs = max_epochs; k.id = *ticket = t->next_ticket++;
    return GM_OK;
}

int gm_trk_collect(gm_trk* t, uint64_t ticket, int wait, gm_trk_out* outs, uint8_t* processed, uint8_t* lost, gm_trk_state* states,
                   uint32_t* epochs_done, int* ready) {
    if (!t || !ticket || !ready) return set_err(GM_ERR_INVALID_ARG, "bad argument");
    gm_trk::Ticket& k = t->tk[ticket % gm_trk::TICKETS];
    if (!k.in_flight || k.id != ticket) return set_err(GM_ERR_INVALID_ARG, "no such ticket (collected already, or never issued)");
    if (int rc = ensure_device(t->device)) return rc;
    *ready = 0;
    // contract (include/gnss_mi355x.h): a collect that returns an ERROR has consumed the ticket — the slot is released on hard HIP
    // errors too, or it would be lost for good and every TICKETS-th asynchronous call refused from then on
    if (wait) {
        const hipError_t w = wait_event_polling(k.done);
        if (w != hipSuccess) { k.in_flight = false; return hip_fail(w, "wait(ticket)"); }
    } else {
        const hipError_t q = hipEventQuery(k.done);
        if (q == hipErrorNotReady) { (void)hipGetLastError(); return GM_OK; }
        if (q != hipSuccess) { k.in_flight = false; return hip_fail(q, "hipEventQuery(ticket)"); }
    }
    k.in_flight = false;
    if (int rc = trk_check_error(t)) return rc;
    const size_t n = size_t(k.epochs) * t->C;
    if (outs) memcpy(outs, k.h, n * sizeof(gm_trk_out));
    if (processed) memcpy(processed, k.h + n * sizeof(gm_trk_out), n);
    if (lost) memcpy(lost, k.h + n * sizeof(gm_trk_out) + n, n);
    if (states) memcpy(states, k.h + ((n * (sizeof(gm_trk_out) + 2) + 7) & ~size_t(7)), size_t(t->C) * sizeof(gm_trk_state));
    if (epochs_done) *epochs_done = trk_epochs_done(k.h + n * sizeof(gm_trk_out), k.epochs, t->C);
    *ready = 1;
    return GM_OK;
}

int gm_trk_update_all(gm_trk* t, gm_ring* ring, uint32_t max_epochs, gm_trk_out* outs, uint8_t* processed, uint8_t* lost,
                      uint32_t* epochs_done) {
    if (int rc = gm_trk_update_all_dev(t, ring, max_epochs)) return rc;
    // the three result arrays land in one pinned area behind the launch, on its stream: one synchronisation per call
    // (three blocking hipMemcpy calls from pageable memory cost more than the 16 epochs of a 16 ms block)
    const size_t n = size_t(max_epochs) * t->C;
    uint8_t* h_outs = t->h_res;
    uint8_t* h_proc = h_outs + n * sizeof(gm_trk_out);
    uint8_t* h_lost = h_proc + n;
    if (outs) HIPC(hipMemcpyAsync(h_outs, t->d_res, n * (sizeof(gm_trk_out) + 2), hipMemcpyDeviceToHost, t->stream));      // [outs | processed | lost]: one copy
    else HIPC(hipMemcpyAsync(h_proc, t->d_proc, 2 * n, hipMemcpyDeviceToHost, t->stream));
    HIPC(hipStreamSynchronize(t->stream));
    if (int rc = trk_check_error(t)) return rc;
    const uint8_t* proc = h_proc;
    if (outs) memcpy(outs, h_outs, n * sizeof(gm_trk_out));
    if (processed) memcpy(processed, h_proc, n);
    if (lost) memcpy(lost, h_lost, n);
    if (epochs_done) *epochs_done = trk_epochs_done(proc, max_epochs, t->C);
    return GM_OK;
}

static int trk_check_error(gm_trk* t) {
    const int err = *static_cast<volatile int*>(t->d_error);      // the stream has been synchronised by the caller
    if (err) {
        *t->d_error = 0;
        (void)hipMemsetAsync(t->d_error_dev, 0, sizeof(int), t->stream);
        (void)hipStreamSynchronize(t->stream);
        if (err == 2) return set_err(GM_ERR_OUT_OF_RANGE, "tracking (strict_sum_order): a code period longer than the product streams were sized for (nominal + 1 %)");
        return set_err(GM_ERR_HIP, "tracking: inter-workgroup exchange timed out (workgroups of a channel not co-resident?)");
    }
    return GM_OK;
}

int gm_trk_synchronize(gm_trk* t) {
    if (!t) return set_err(GM_ERR_INVALID_ARG, "null handle");
    if (int rc = ensure_device(t->device)) return rc;
    HIPC(hipStreamSynchronize(t->stream));
    return trk_check_error(t);
}

// Diagnostic: s_memtime stamps of workgroup 0 at 8 phase boundaries of each epoch of the next launches
// (epochs <= cap).  out = [cap][8] after gm_trk_synchronize.  Not part of the reference API.
int gm_trk_debug_stamps(gm_trk* t, uint32_t cap, long long* out) {
    if (!t) return set_err(GM_ERR_INVALID_ARG, "null handle");
    if (int rc = ensure_device(t->device)) return rc;
    if (!out) {   // arm
        hipFree(t->d_stamps); t->d_stamps = nullptr; t->stamps_cap = 0;
        if (cap) { HIPC(hipMalloc(&t->d_stamps, size_t(cap) * 48 * sizeof(long long))); HIPC(hipMemset(t->d_stamps, 0, size_t(cap) * 384)); HIPC(hipStreamSynchronize(nullptr)); t->stamps_cap = cap; }
        return GM_OK;
    }
    if (!t->d_stamps || cap > t->stamps_cap) return set_err(GM_ERR_INVALID_ARG, "stamps not armed");
    HIPC(hipStreamSynchronize(t->stream));
    HIPC(hipMemcpy(out, t->d_stamps, size_t(cap) * 384, hipMemcpyDeviceToHost));
    return GM_OK;
}

int gm_trk_enable_timing(gm_trk* t, int on) {
    if (!t) return set_err(GM_ERR_INVALID_ARG, "null handle");
    t->timing = on != 0;
    return GM_OK;
}

int gm_trk_last_timing(gm_trk* t, float* ms_total, uint32_t* launches) {
    if (!t || !t->timing) return set_err(GM_ERR_INVALID_ARG, "timing not enabled");
    if (int rc = ensure_device(t->device)) return rc;
    HIPC(hipStreamSynchronize(t->stream));
    float ms = 0;
    HIPC(hipEventElapsedTime(&ms, t->ev0, t->ev1));
    if (ms_total) *ms_total = ms;
    if (launches) *launches = t->timed_launches;
    return GM_OK;
}

}  // extern "C"

// ====================================================================== digital front-end
struct gm_frontend {
    int device = -1;
    float f_if = 0, fs_in = 0, fs_out = 0, phase_step = 0, alpha = 0.001f, con = 0;
    std::vector<float> lut;                 // [2][2048]
    float* d_lut = nullptr;
    gm::FeState* d_state = nullptr;
    hipStream_t stream = nullptr;
    void* d_io = nullptr;                   // scratch for the host-buffer entry
    size_t io_cap = 0;
    void* d_raw[gm_ring::SLOTS] = {nullptr, nullptr, nullptr, nullptr};   // raw-format landing zones of the ring writer
    gm::FrontendArgs::Stream* d_batch = nullptr;   // stream descriptors of gm_frontend_process_dev_batch
    uint32_t batch_cap = 0;
    // tabulated NCO phase orbit (fe_kernels.hip, fast form): built on first use
    bool tab_tried = false, on_orbit = true;       // on_orbit: the device phase_accumulator is entry `pos` of the orbit from 0
    std::vector<float> tab;                        // |phase| / 2048 before step k, k < mu + lambda'
    float* d_tab = nullptr;
    uint32_t tab_mu = 0, tab_lambda = 0;           // lambda' = the period repeated until >= FE_FAST_SEG
    uint32_t pos = 0;                              // table index of the next sample
    float* d_spec = nullptr;                       // [FE_SPEC_K][32] + 1 floats: the speculative form's run records (fe_kernels.hip)
};
static constexpr int FE_SPEC_K = gm::FE_SPEC_K_MAX;
static constexpr size_t FE_SPEC_MIN = size_t(gm::FE_FAST_SEG) * 48;     // blocks shorter than 48 pipeline segments stay on one workgroup

// r -> fract(fl(r + s)): the fast form of `(phase + step) % 2048` in revolutions (fe_kernels.hip nco_segment_fast)
static inline float fe_orbit_step(float r, float s) { const float t = r + s; return t - floorf(t); }

// The orbit of the NCO phase from 0: transient mu and period lambda by Brent's algorithm, then the table.  Only for the
// settings the sequential kernel's fast form covers (|step| < 2048, not denormal-small); false otherwise or when the
// orbit is longer than 2^24 + 2^16 steps (never seen: periods are <= 2^23).
static bool frontend_build_table(gm_frontend* f) {
    if (f->tab_tried) return !f->tab.empty();
    f->tab_tried = true;
    const float step = f->phase_step;
    if (!(fabsf(step) < 2048.0f) || !(step == 0.0f || fabsf(step) > 1.0e-20f)) return false;
    const float s = fabsf(step) * (1.0f / 2048.0f);
    const uint64_t cap = (1ull << 24) + (1ull << 16);
    uint64_t power = 1, lam = 1;
    float tort = 0.0f, hare = fe_orbit_step(0.0f, s);
    while (tort != hare) {
        if (power == lam) { tort = hare; power *= 2; lam = 0; }
        hare = fe_orbit_step(hare, s);
        if (++lam > cap) return false;
    }
    uint64_t mu = 0;
    tort = hare = 0.0f;
    for (uint64_t i = 0; i < lam; ++i) hare = fe_orbit_step(hare, s);
    while (tort != hare) { tort = fe_orbit_step(tort, s); hare = fe_orbit_step(hare, s); if (++mu > cap) return false; }
    const uint64_t reps = (uint64_t(gm::FE_FAST_SEG) + lam - 1) / lam;
    const uint64_t lam2 = lam * reps;
    if (mu + lam2 > cap) return false;
    f->tab.resize(size_t(mu + lam2));
    float r = 0.0f;
    for (size_t k = 0; k < f->tab.size(); ++k) { f->tab[k] = r; r = fe_orbit_step(r, s); }
    if (hipMalloc(&f->d_tab, f->tab.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(f->d_tab, f->tab.data(), f->tab.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        hipFree(f->d_tab); f->d_tab = nullptr; f->tab.clear();
        return false;
    }
    f->tab_mu = uint32_t(mu); f->tab_lambda = uint32_t(lam2);
    return true;
}
static inline uint32_t frontend_fold(const gm_frontend* f, uint64_t p) {
    const uint64_t len = uint64_t(f->tab_mu) + f->tab_lambda;
    return uint32_t(p < len ? p : f->tab_mu + (p - f->tab_mu) % f->tab_lambda);
}

static gm::FrontendArgs::Stream frontend_stream(gm_frontend* f, const void* d_in, void* d_out, uint64_t out_start,
                                                 uint64_t out_mask, size_t n) {
    gm::FrontendArgs::Stream s{};
    s.in = d_in; s.out = d_out; s.out_start = out_start; s.out_mask = out_mask; s.n_samples = n; s.state = f->d_state;
    s.phase_step = f->phase_step;
    s.fast_fmod = fabsf(f->phase_step) < 2048.0f ? 1 : 0;     // false for NaN too
    s.ph_table = nullptr;
    if (f->on_orbit && frontend_build_table(f)) {             // fast form: phases looked up, position tracked on the host
        const size_t n8 = n & ~size_t(7);
        s.ph_table = f->d_tab;
        s.tab_len = f->tab_mu + f->tab_lambda; s.tab_lambda = f->tab_lambda;
        s.tab_pos = f->pos;
        s.tab_pos_end = frontend_fold(f, uint64_t(f->pos) + n8);
        s.tab_scale = f->phase_step < 0.0f ? -2048.0f : 2048.0f;
        f->pos = s.tab_pos_end;
    }
    return s;
}

static int frontend_launch(gm_frontend* f, hipStream_t st, const void* d_in, int fmt, void* d_out, uint64_t out_start,
                           uint64_t out_mask, size_t n) {
    gm::FrontendArgs a{};
    a.streams = nullptr;
    a.one = frontend_stream(f, d_in, d_out, out_start, out_mask, n);
    a.lut = f->d_lut; a.alpha = f->alpha; a.con = f->con;
    // the speculative form: a long block whose output does not alias its input (the runs' warm-ups re-read earlier samples)
    const int spec = gm::diag_int("GM_FE_SPEC", 1);
    if (a.one.ph_table && spec != 0 && n >= FE_SPEC_MIN && d_in != d_out) {
        if (!f->d_spec) {
            HIPC(hipMalloc(&f->d_spec, (size_t(FE_SPEC_K) * 32 + 1) * sizeof(float)));
            HIPC(hipMemsetAsync(f->d_spec, 0, (size_t(FE_SPEC_K) * 32 + 1) * sizeof(float), st));
        }
        {
            const int k = gm::diag_int("GM_FE_SPEC_K", 32);      // 32 runs: 126 -> 113 us per 2^19-sample block against 16 (48, 64: 107 us, more runs to repair; DESIGN_HISTORY R6.5)
            a.spec_k = k < 2 ? 2 : (k > FE_SPEC_K ? FE_SPEC_K : k);
            a.spec_warm = gm::diag_int("GM_FE_SPEC_WARM", 0);
        }
        a.spec_buf = f->d_spec; a.spec_poison = spec == 2 ? 1 : 0;
        a.spec_repairs = reinterpret_cast<unsigned int*>(f->d_spec + size_t(FE_SPEC_K) * 32);
        gm::launch_frontend_spec(st, a, fmt);
    } else if (a.one.ph_table) gm::launch_frontend_fast(st, a, 1, fmt);
    else gm::launch_frontend(st, a, 1, fmt);
    HIPC(hipGetLastError());
    return GM_OK;
}

extern "C" {

int gm_frontend_create(float f_if, float fs_in, float fs_out, gm_frontend** out) {   // DigitalFrontend::new frontend.rs:19-30
    if (!out) return set_err(GM_ERR_INVALID_ARG, "null out");
    *out = nullptr;
    if (int rc = ensure_device(g_device)) return rc;
    gm_frontend* f = new gm_frontend;
    f->device = g_device; f->f_if = f_if; f->fs_in = fs_in; f->fs_out = fs_out;
    f->lut.resize(2 * 2048);
    for (int i = 0; i < 2048; ++i) {                          // NcoLut::new nco_lut.rs:28-32
        const float angle = ((2.0f * PI_F) * float(i)) / 2048.0f;
        f->lut[i] = cosf(angle);
        f->lut[2048 + i] = -sinf(angle);
    }
    f->phase_step = (f_if / fs_in) * 2048.0f;                 // :34
    f->alpha = 0.001f;                                        // DcRemoverSimd::new(0.001), frontend.rs:21
    f->con = 1.0f - f->alpha;                                 // dc_remove.rs:12
    hipError_t e = hipMalloc(&f->d_lut, 2 * 2048 * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(f->d_lut, f->lut.data(), 2 * 2048 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&f->d_state, sizeof(gm::FeState));
    if (e == hipSuccess) e = hipMemset(f->d_state, 0, sizeof(gm::FeState));
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);    // the fill has run before a kernel on a non-blocking stream can touch the state (see gm_ring_create)
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { gm_frontend_destroy(f); return hip_fail(e, "gm_frontend_create"); }
    *out = f;
    return GM_OK;
}

int gm_frontend_destroy(gm_frontend* f) {
    if (!f) return GM_OK;
    hipSetDevice(f->device);
    if (f->stream) { hipStreamSynchronize(f->stream); hipStreamDestroy(f->stream); }
    hipFree(f->d_lut); hipFree(f->d_state); hipFree(f->d_io); hipFree(f->d_batch); hipFree(f->d_tab); hipFree(f->d_spec);
    for (void* p : f->d_raw) hipFree(p);
    delete f;
    return GM_OK;
}

int gm_frontend_debug_repairs(gm_frontend* f, uint32_t* runs) {
    if (!f || !runs) return set_err(GM_ERR_INVALID_ARG, "null pointer");
    *runs = 0;
    if (!f->d_spec) return GM_OK;
    if (int rc = ensure_device(f->device)) return rc;
    HIPC(hipDeviceSynchronize());
    HIPC(hipMemcpy(runs, f->d_spec + size_t(FE_SPEC_K) * 32, sizeof(uint32_t), hipMemcpyDeviceToHost));
    return GM_OK;
}

int gm_frontend_lut(gm_frontend* f, float* lut_re, float* lut_im, float* phase_step) {
    if (!f) return set_err(GM_ERR_INVALID_ARG, "null handle");
    if (lut_re) memcpy(lut_re, f->lut.data(), 2048 * sizeof(float));
    if (lut_im) memcpy(lut_im, f->lut.data() + 2048, 2048 * sizeof(float));
    if (phase_step) *phase_step = f->phase_step;
    return GM_OK;
}

int gm_frontend_get_state(gm_frontend* f, float* phase_accumulator, float bias_re[8], float bias_im[8]) {
    if (!f) return set_err(GM_ERR_INVALID_ARG, "null handle");
    if (int rc = ensure_device(f->device)) return rc;
    gm::FeState s;
    HIPC(hipStreamSynchronize(f->stream));
    HIPC(hipMemcpy(&s, f->d_state, sizeof(s), hipMemcpyDeviceToHost));
    if (phase_accumulator) *phase_accumulator = s.phase_accumulator;
    if (bias_re) memcpy(bias_re, s.bias_re, sizeof(s.bias_re));
    if (bias_im) memcpy(bias_im, s.bias_im, sizeof(s.bias_im));
    return GM_OK;
}

int gm_frontend_set_state(gm_frontend* f, float phase_accumulator, const float bias_re[8], const float bias_im[8]) {
    if (!f || !bias_re || !bias_im) return set_err(GM_ERR_INVALID_ARG, "null pointer");
    if (int rc = ensure_device(f->device)) return rc;
    gm::FeState s;
    s.phase_accumulator = phase_accumulator;
    memcpy(s.bias_re, bias_re, sizeof(s.bias_re));
    memcpy(s.bias_im, bias_im, sizeof(s.bias_im));
    HIPC(hipStreamSynchronize(f->stream));
    HIPC(hipMemcpy(f->d_state, &s, sizeof(s), hipMemcpyHostToDevice));
    // where on the tabulated orbit is this phase?  (0 = its start; any other value is looked up; a phase that is not on
    // the orbit from 0 sends this front-end down the sequential kernel from here on)
    f->on_orbit = false;
    if (phase_accumulator == 0.0f && !std::signbit(phase_accumulator)) { f->on_orbit = true; f->pos = 0; }
    else if (frontend_build_table(f)) {
        const bool neg = f->phase_step < 0.0f;
        const bool sign_ok = neg ? !(phase_accumulator > 0.0f) : !(phase_accumulator < 0.0f);
        if (sign_ok && fabsf(phase_accumulator) < 2048.0f) {
            const float r = fabsf(phase_accumulator) * (1.0f / 2048.0f);
            for (size_t k = 0; k < f->tab.size(); ++k)
                if (f->tab[k] == r && (r != 0.0f || std::signbit(phase_accumulator) == neg || k == 0)) { f->on_orbit = true; f->pos = uint32_t(k); break; }
        }
    }
    return GM_OK;
}

// DigitalFrontend::process_block (frontend.rs:33-62) on a host buffer, in place: H2D, kernel, D2H.
int gm_frontend_process_block(gm_frontend* f, float* raw_floats, size_t n_floats) {
    if (!f || (!raw_floats && n_floats)) return set_err(GM_ERR_INVALID_ARG, "null pointer");
    if (int rc = ensure_device(f->device)) return rc;
    const size_t n = n_floats / 2;                // whole samples; an odd trailing float is never inside a 16-float chunk
    if (!n) return GM_OK;
    if (f->io_cap < n) {
        hipFree(f->d_io); f->d_io = nullptr; f->io_cap = 0;
        HIPC(hipMalloc(&f->d_io, n * 8));
        f->io_cap = n;
    }
    HIPC(hipMemcpyAsync(f->d_io, raw_floats, n * 8, hipMemcpyHostToDevice, f->stream));
    if (int rc = frontend_launch(f, f->stream, f->d_io, GM_FMT_C32, f->d_io, 0, ~0ull, n)) return rc;
    HIPC(hipMemcpyAsync(raw_floats, f->d_io, n * 8, hipMemcpyDeviceToHost, f->stream));
    HIPC(hipStreamSynchronize(f->stream));
    return GM_OK;
}

// Device-resident form: d_in (c32 or int8 IQ) -> d_out (c32), asynchronous on `stream` (NULL: the handle's own).
int gm_frontend_process_dev(gm_frontend* f, const void* d_in, int fmt, void* d_out, size_t n_samples, void* stream) {
    if (!f || !d_in || !d_out) return set_err(GM_ERR_INVALID_ARG, "null pointer");
    if (fmt != GM_FMT_C32 && fmt != GM_FMT_I8_IQ) return set_err(GM_ERR_INVALID_ARG, "front-end input is c32 or int8 IQ");
    if (int rc = ensure_device(f->device)) return rc;
    return frontend_launch(f, stream ? static_cast<hipStream_t>(stream) : f->stream, d_in, fmt, d_out, 0, ~0ull, n_samples);
}

// Many independent streams (antennas, bands) in ONE launch: one workgroup per stream, each with its own front-end state
// and NCO step.  Descriptors travel in a small device array owned by fes[0]; asynchronous on `stream` (NULL: fes[0]'s).
int gm_frontend_process_dev_batch(gm_frontend* const* fes, uint32_t n_streams, const void* const* d_in, int fmt,
                                  void* const* d_out, size_t n_samples, void* stream) {
    if (!fes || !d_in || !d_out || !n_streams) return set_err(GM_ERR_INVALID_ARG, "null pointer / no streams");
    if (fmt != GM_FMT_C32 && fmt != GM_FMT_I8_IQ) return set_err(GM_ERR_INVALID_ARG, "front-end input is c32 or int8 IQ");
    gm_frontend* f0 = fes[0];
    if (!f0) return set_err(GM_ERR_INVALID_ARG, "null handle");
    if (int rc = ensure_device(f0->device)) return rc;
    std::vector<gm::FrontendArgs::Stream> h(n_streams);
    for (uint32_t i = 0; i < n_streams; ++i) {
        if (!fes[i] || !d_in[i] || !d_out[i] || fes[i]->device != f0->device) return set_err(GM_ERR_INVALID_ARG, "bad stream entry");
        for (uint32_t j = 0; j < i; ++j)
            if (fes[j] == fes[i]) return set_err(GM_ERR_INVALID_ARG, "a front-end may appear once per batch (its state is sequential)");
        h[i] = frontend_stream(fes[i], d_in[i], d_out[i], 0, ~0ull, n_samples);
    }
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : f0->stream;
    if (f0->batch_cap < n_streams) {
        HIPC(hipStreamSynchronize(st));
        hipFree(f0->d_batch); f0->d_batch = nullptr; f0->batch_cap = 0;
        HIPC(hipMalloc(&f0->d_batch, n_streams * sizeof(gm::FrontendArgs::Stream)));
        f0->batch_cap = n_streams;
    }
    HIPC(hipMemcpyAsync(f0->d_batch, h.data(), n_streams * sizeof(gm::FrontendArgs::Stream), hipMemcpyHostToDevice, st));
    HIPC(hipStreamSynchronize(st));     // the descriptors were read from pageable host memory
    gm::FrontendArgs a{};
    a.streams = f0->d_batch; a.lut = f0->d_lut; a.alpha = f0->alpha; a.con = f0->con;
    bool all_fast = true;
    for (uint32_t i = 0; i < n_streams; ++i) all_fast = all_fast && h[i].ph_table != nullptr;
    if (all_fast) gm::launch_frontend_fast(st, a, int(n_streams), fmt);
    else gm::launch_frontend(st, a, int(n_streams), fmt);
    HIPC(hipGetLastError());
    return GM_OK;
}

int gm_frontend_synchronize(gm_frontend* f) {
    if (!f) return set_err(GM_ERR_INVALID_ARG, "null handle");
    if (int rc = ensure_device(f->device)) return rc;
    HIPC(hipStreamSynchronize(f->stream));
    return GM_OK;
}

// rf_thread's block step (rf_thread.rs:43-48): process_block + shared_ring_buffer.write_samples, fused and
// non-blocking: raw samples (c32, or int8 IQ = 2 B/sample over PCIe) are staged in the ring's pinned slots, copied on
// the ring's stream, processed there straight into the ring mirror, and `head` is published when they have landed.
int gm_frontend_write_ring(gm_frontend* f, gm_ring* r, const void* samples, size_t n_samples, int fmt) {
    if (!f || !r || (!samples && n_samples)) return set_err(GM_ERR_INVALID_ARG, "null pointer");
    if (fmt != GM_FMT_C32 && fmt != GM_FMT_I8_IQ) return set_err(GM_ERR_INVALID_ARG, "front-end input is c32 or int8 IQ");
    if (f->device != r->device) return set_err(GM_ERR_INVALID_ARG, "front-end and ring live on different devices");
    if (n_samples > r->size) return set_err(GM_ERR_OUT_OF_RANGE, "write larger than the ring");
    if (int rc = ensure_device(r->device)) return rc;
    if (int rc = ring_async_init(r)) return rc;
    if (!r->fe_stream) {
        if (gm::diag_int("GM_RING_FE_STREAM", 1) == 0) r->fe_stream = r->copy_stream;      // diagnostic: kernels on the copy stream (round 4's form)
        else {      // (highest priority: see ring_async_init)
            int least = 0, greatest = 0;
            HIPC(hipDeviceGetStreamPriorityRange(&least, &greatest));
            const int pr = gm::diag_int("GM_RING_FE_PRIORITY", 99);
            HIPC(hipStreamCreateWithPriority(&r->fe_stream, hipStreamNonBlocking, pr == 99 ? greatest : pr));
        }
        for (int i = 0; i < gm_ring::SLOTS; ++i) HIPC(hipEventCreateWithFlags(&r->h2d_done[i], hipEventDisableTiming));
    }
    const size_t bps = fmt == GM_FMT_C32 ? 8 : 2;
    const uint8_t* src = static_cast<const uint8_t*>(samples);
    while (n_samples) {
        const size_t chunk = n_samples < r->slot_samples ? n_samples : r->slot_samples;
        const int slot = int(r->slot_seq++ % gm_ring::SLOTS);
        if (int rc = ring_reclaim_slot(r, slot)) return rc;
        if (!f->d_raw[slot]) HIPC(hipMalloc(&f->d_raw[slot], gm_ring::SLOT_SAMPLES_MAX * 8));
        memcpy(r->staging[slot], src, chunk * bps);
        // (the kernel reading the pinned staging slot itself instead of a device copy of it — no copy, no event — was measured:
        //  30 x real time against 52: the front-end's loads then wait on host memory)
        HIPC(hipMemcpyAsync(f->d_raw[slot], r->staging[slot], chunk * bps, hipMemcpyHostToDevice, r->copy_stream));
        HIPC(hipEventRecord(r->h2d_done[slot], r->copy_stream));
        HIPC(hipStreamWaitEvent(r->fe_stream, r->h2d_done[slot], 0));
        if (int rc = frontend_launch(f, r->fe_stream, f->d_raw[slot], fmt, r->d_buf, r->write_pos, r->mask, chunk)) return rc;
        HIPC(hipEventRecord(r->slot_done[slot], r->fe_stream));
        r->slot_used[slot] = true;
        r->write_pos += chunk;
        if (int rc = ring_enqueue_publish(r, slot, r->fe_stream)) return rc;
        src += chunk * bps; n_samples -= chunk;
    }
    return GM_OK;
}

}  // extern "C"

// ====================================================================== multi-GPU exchange
// The path's ONE exchange step: all-gather of the per-(worker, bin) metrics over RCCL, enqueued on the acquisition
// handle's stream, followed by the regroup to the [3][nranks*P][D] layout gm_acq_decide_dev replays.  RCCL is bound
// lazily (dlopen of librccl.so.1 — inside a PyTorch process that is the copy PyTorch already loaded), so single-GPU
// hosts never need it.
#include <rccl/rccl.h>

namespace {
struct RcclApi {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char* n : names)
            if ((api.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!api.so) return;
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.so, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.so, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.so, "ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(api.so, "ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.so, "ncclGetErrorString"));
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.GetErrorString;
    });
    return api;
}
int rccl_fail(ncclResult_t r, const char* where) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", where, rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error");
    g_last_error = buf;
    return GM_ERR_HIP;
}
#define RCCLC(expr)                                           \
    do {                                                      \
        ncclResult_t _r = (expr);                             \
        if (_r != ncclSuccess) return rccl_fail(_r, #expr);   \
    } while (0)

// gathered [R][3][PD] -> out [3][R][PD]  (== [3][R*P][D])
__global__ void regroup_metrics_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t R, uint32_t PD) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = R * 3u * PD;
    if (i >= total) return;
    const uint32_t e = i % PD, q = (i / PD) % 3u, r = i / (3u * PD);
    out[(size_t(q) * R + r) * PD + e] = in[i];
}
}  // namespace

struct gm_comm {
    int device = -1;
    int nranks = 0, rank = 0;
    ncclComm_t comm = nullptr;
    uint32_t* d_stage = nullptr;   // [nranks][3][PD] as the all-gather lays it out
    size_t stage_words = 0;
    // overlapped exchange (gm_acq_allgather_metrics_async): the collective runs on the communicator's own stream between
    // two events, so the next dwell's kernels on the handle's stream do not wait for it
    hipStream_t xs = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    hipEvent_t ev_sync = nullptr;  // behind the last SYNCHRONOUS exchange's regroup (it reads d_stage on the caller's stream)
    bool pending = false, async_used = false, sync_used = false;
};

extern "C" {

int gm_comm_get_unique_id(uint8_t id[GM_COMM_ID_BYTES]) {
    if (!id) return set_err(GM_ERR_INVALID_ARG, "null id");
    static_assert(GM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!rccl().ok) return set_err(GM_ERR_UNSUPPORTED, "librccl.so.1 not found");
    ncclUniqueId u;
    RCCLC(rccl().GetUniqueId(&u));
    memcpy(id, u.internal, GM_COMM_ID_BYTES);
    return GM_OK;
}

int gm_comm_init(int nranks, int rank, const uint8_t id[GM_COMM_ID_BYTES], gm_comm** out) {
    if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) return set_err(GM_ERR_INVALID_ARG, "bad nranks/rank/id");
    if (int rc = ensure_device(g_device)) return rc;
    if (!rccl().ok) return set_err(GM_ERR_UNSUPPORTED, "librccl.so.1 not found");
    ncclUniqueId u;
    memcpy(u.internal, id, GM_COMM_ID_BYTES);
    gm_comm* c = new gm_comm;
    c->device = g_device; c->nranks = nranks; c->rank = rank;
    ncclResult_t r = rccl().CommInitRank(&c->comm, nranks, u, rank);
    if (r != ncclSuccess) { delete c; return rccl_fail(r, "ncclCommInitRank"); }
    *out = c;
    return GM_OK;
}

int gm_comm_destroy(gm_comm* c) {
    if (!c) return GM_OK;
    if (c->device >= 0) hipSetDevice(c->device);
    if (c->xs) { hipStreamSynchronize(c->xs); hipStreamDestroy(c->xs); }
    if (c->ev_in) hipEventDestroy(c->ev_in);
    if (c->ev_out) hipEventDestroy(c->ev_out);
    if (c->ev_sync) hipEventDestroy(c->ev_sync);
    if (c->comm) rccl().CommDestroy(c->comm);
    hipFree(c->d_stage);
    delete c;
    return GM_OK;
}

int gm_comm_info(gm_comm* c, int* nranks, int* rank) {
    if (!c) return set_err(GM_ERR_INVALID_ARG, "null comm");
    if (nranks) *nranks = c->nranks;
    if (rank) *rank = c->rank;
    return GM_OK;
}

static int comm_gather_regroup(gm_acq* a, gm_comm* c, const void* d_local, void* d_all, hipStream_t st);
int gm_acq_allgather_metrics(gm_acq* a, gm_comm* c, const void* d_local, void* d_all) {
    if (!a || !c || !d_all) return set_err(GM_ERR_INVALID_ARG, "null handle/comm/output");
    if (a->device != c->device) return set_err(GM_ERR_INVALID_ARG, "handle and communicator live on different devices");
    if (int rc = ensure_device(a->device)) return rc;
    // d_stage is ONE buffer: an asynchronous exchange still running on the communicator's stream must have read it out before
    // this one overwrites it (whether or not gm_comm_wait has been called for it yet, and on whichever stream)
    if (c->async_used) HIPC(hipStreamWaitEvent(a->stream, c->ev_out, 0));
    if (int rc = comm_gather_regroup(a, c, d_local, d_all, a->stream)) return rc;
    if (!c->ev_sync) HIPC(hipEventCreateWithFlags(&c->ev_sync, hipEventDisableTiming));
    HIPC(hipEventRecord(c->ev_sync, a->stream));
    c->sync_used = true;
    return GM_OK;
}

static int comm_gather_regroup(gm_acq* a, gm_comm* c, const void* d_local, void* d_all, hipStream_t st) {
    const size_t PD = size_t(a->P) * a->D, words = 3 * PD;
    const uint32_t* src = d_local ? static_cast<const uint32_t*>(d_local) : a->d_metrics;
    if (c->stage_words < words * c->nranks) {
        if (c->xs) HIPC(hipStreamSynchronize(c->xs));
        hipFree(c->d_stage); c->d_stage = nullptr; c->stage_words = 0;
        HIPC(hipMalloc(&c->d_stage, words * c->nranks * sizeof(uint32_t)));
        c->stage_words = words * c->nranks;
    }
    RCCLC(rccl().AllGather(src, c->d_stage, words, ncclInt32, c->comm, st));
    const uint32_t total = uint32_t(words * c->nranks);
    regroup_metrics_kernel<<<(total + 255) / 256, 256, 0, st>>>(c->d_stage, static_cast<uint32_t*>(d_all),
                                                                 uint32_t(c->nranks), uint32_t(PD));
    HIPC(hipGetLastError());
    return GM_OK;
}

int gm_acq_allgather_metrics_async(gm_acq* a, gm_comm* c, const void* d_local, void* d_all) {
    if (!a || !c || !d_all) return set_err(GM_ERR_INVALID_ARG, "null handle/comm/output");
    if (a->device != c->device) return set_err(GM_ERR_INVALID_ARG, "handle and communicator live on different devices");
    if (int rc = ensure_device(a->device)) return rc;
    if (!c->xs) {
        HIPC(hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking));
        HIPC(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
        HIPC(hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
    }
    HIPC(hipEventRecord(c->ev_in, a->stream));          // everything enqueued so far on the handle's stream (the search)
    HIPC(hipStreamWaitEvent(c->xs, c->ev_in, 0));
    if (c->sync_used) HIPC(hipStreamWaitEvent(c->xs, c->ev_sync, 0));     // ... and a synchronous exchange of ANOTHER handle through this communicator
    if (int rc = comm_gather_regroup(a, c, d_local, d_all, c->xs)) return rc;
    HIPC(hipEventRecord(c->ev_out, c->xs));
    c->pending = true;
    c->async_used = true;
    return GM_OK;
}

int gm_comm_wait(gm_comm* c, void* hip_stream) {
    if (!c) return set_err(GM_ERR_INVALID_ARG, "null comm");
    if (int rc = ensure_device(c->device)) return rc;
    if (c->pending) {
        HIPC(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(hip_stream), c->ev_out, 0));
        c->pending = false;
    }
    return GM_OK;
}

int gm_comm_allgather_words(gm_comm* c, const void* d_local, void* d_all, size_t words, void* hip_stream) {
    if (!c || !d_local || !d_all || !words) return set_err(GM_ERR_INVALID_ARG, "null comm/buffer or zero words");
    if (int rc = ensure_device(c->device)) return rc;
    RCCLC(rccl().AllGather(d_local, d_all, words, ncclInt32, c->comm, reinterpret_cast<hipStream_t>(hip_stream)));
    return GM_OK;
}

}  // extern "C"

namespace {
// gathered [R][3][pmax][D] -> out [3][n_rows][D], row i taken from block row_map[i] = rank * pmax + row
__global__ void grid_assemble_kernel(const uint32_t* __restrict__ in, const uint32_t* __restrict__ row_map, uint32_t pmax,
                                     uint32_t D, uint32_t n_rows, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3u * n_rows * D) return;
    const uint32_t e = i % D, row = (i / D) % n_rows, q = i / (D * n_rows);
    const uint32_t src = row_map[row], r = src / pmax, k = src - r * pmax;
    out[i] = in[((size_t(r) * 3u + q) * pmax + k) * D + e];
}
}  // namespace

extern "C" {

int gm_grid_assemble_dev(const void* d_gathered, uint32_t nranks, uint32_t p_max, uint32_t n_bins, const uint32_t* d_row_map,
                         uint32_t n_rows, void* d_out, void* hip_stream) {
    if (!d_gathered || !d_row_map || !d_out || !nranks || !p_max || !n_bins || !n_rows)
        return set_err(GM_ERR_INVALID_ARG, "null buffer or zero dimension");
    if (n_rows > nranks * p_max) return set_err(GM_ERR_OUT_OF_RANGE, "more rows than the gathered blocks hold");
    if (int rc = ensure_device(g_device)) return rc;
    const uint32_t total = 3u * n_rows * n_bins;
    grid_assemble_kernel<<<(total + 255) / 256, 256, 0, reinterpret_cast<hipStream_t>(hip_stream)>>>(
        static_cast<const uint32_t*>(d_gathered), d_row_map, p_max, n_bins, n_rows, static_cast<uint32_t*>(d_out));
    HIPC(hipGetLastError());
    return GM_OK;
}

int gm_acq_decide_planes_dev(const float* d_max, const uint32_t* d_argmax, const float* d_sum, uint32_t n_prn, uint32_t n_bins,
                             const uint8_t* d_prn_ids, const float* d_table_freq, uint32_t fft_size, float fs, float code_rate,
                             float threshold, int decision_mode, uint64_t local_tail, gm_acq_result* d_results, uint8_t* d_found,
                             void* hip_stream) {
    if (!d_max || !d_argmax || !d_sum || !d_prn_ids || !d_table_freq || !d_results || !d_found || !n_prn || !n_bins || fft_size < 2)
        return set_err(GM_ERR_INVALID_ARG, "null buffer or zero dimension");
    if (int rc = ensure_device(g_device)) return rc;
    gm::DecideArgs da;
    da.mmax = d_max; da.margmax = d_argmax; da.msum = d_sum;
    da.table_freq = d_table_freq; da.prn_ids = d_prn_ids;
    da.mask_lo = ~0ull;
    da.n_prn = int(n_prn); da.n_bins = int(n_bins); da.fft_size = int(fft_size);
    da.fs = fs; da.threshold = threshold; da.code_rate = code_rate;
    da.best_bin_mode = decision_mode == GM_DECIDE_BEST_BIN ? 1 : 0;
    da.local_tail = local_tail;
    da.results = d_results; da.found = d_found;
    gm::launch_decide(reinterpret_cast<hipStream_t>(hip_stream), da);
    HIPC(hipGetLastError());
    return GM_OK;
}

}  // extern "C"

