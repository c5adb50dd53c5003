// gnss_sdr.hpp — C++ host-side mirror of the reference crate's acquisition / tracking API over the C ABI
// (include/gnss_mi355x.h).  Header-only; link with libgnss_mi355x.so.  Type and method names follow the
// reference (file:line in the comments) so call sites port one to one; errors the reference turns into panics
// become gnss::Panic exceptions HERE (above the ABI — nothing unwinds across it).
#pragma once
#include <atomic>
#include <chrono>
#include <complex>
#include <algorithm>
#include <deque>
#include <functional>
#include <memory>
#include <map>
#include <mutex>
#include <thread>
#include <cstdint>
#include <cmath>
#include <limits>
#include <optional>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/gnss_mi355x.h"

namespace gnss {

using Complex32 = std::complex<float>;   // num_complex::Complex32, layout-compatible with gm_c32

struct Panic : std::runtime_error {
    int status;
    Panic(int st, const std::string& where)
        : std::runtime_error(where + ": " + gm_status_string(st) + " (" + gm_last_error() + ")"), status(st) {}
};
inline void check(int st, const char* where) { if (st != GM_OK) throw Panic(st, where); }
inline void init(int device = 0) { check(gm_init(device), "gm_init"); }

// ---- utilities::ca_code (src/utilities/ca_code.rs:12-27)
inline std::vector<int8_t> generate_ca_code_samples(uint8_t prn, float code_rate, float f_sampling) {
    size_t n = 0;
    check(gm_generate_ca_code_samples(prn, code_rate, f_sampling, nullptr, 0, &n), "generate_ca_code_samples");
    std::vector<int8_t> v(n);
    check(gm_generate_ca_code_samples(prn, code_rate, f_sampling, v.data(), v.size(), &n), "generate_ca_code_samples");
    return v;
}

// ---- acquisition::doppler_shift (src/acquisition/doppler_shift.rs:5-58)
struct DopplerShiftTable {
    float doppler_freq_hz = 0.f;          // = f_if + doppler (:20)
    std::vector<Complex32> table;
    DopplerShiftTable(float f_if, float doppler_freq_hz_, float fs, size_t num_samples) : table(num_samples) {
        check(gm_doppler_table_new(f_if, doppler_freq_hz_, fs, num_samples, &doppler_freq_hz,
                                   reinterpret_cast<gm_c32*>(table.data())), "DopplerShiftTable::new");
    }
};
inline void apply_doppler_shift(const std::vector<Complex32>& samples, const DopplerShiftTable& t,
                                std::vector<Complex32>& output) {
    check(gm_apply_doppler_shift(reinterpret_cast<const gm_c32*>(samples.data()),
                                 reinterpret_cast<const gm_c32*>(t.table.data()),
                                 reinterpret_cast<gm_c32*>(output.data()), samples.size()), "apply_doppler_shift");
}

// ---- acquisition::do_acquisition (src/acquisition/do_acquisition.rs)
constexpr uint8_t PRN_SEARCH_ACQUISITION_TOTAL = 32;   // :22
using AcquisitionResult = gm_acq_result;                // :93-116 (+ doppler_bin)

enum class SearchMode { ColdStart = 0, WarmStart = 1, SteadyState = 2 };   // :33-37
class AcquisitionManager {                                                 // :39-74
    SearchMode mode_ = SearchMode::ColdStart;
public:
    SearchMode mode() const { return mode_; }
    void update_mode(size_t trked_acount) { mode_ = SearchMode(gm_acq_manager_mode_for(trked_acount)); }
    std::pair<uint64_t, uint32_t> get_pacing_and_list(const std::set<uint8_t>& active_prns) const {
        uint32_t am = 0;
        for (uint8_t p : active_prns) am |= 1u << (p - 1);
        uint64_t iv = 0; uint32_t mask = 0;
        check(gm_acq_manager_pacing_and_list(int(mode_), am, &iv, &mask), "get_pacing_and_list");
        return {iv, mask};
    }
};

// All workers of one acquisition stage in one handle: the batched replacement of
// `workers.par_iter_mut()...search_satellite(...)` (:268-271, :302-313).
class AcquisitionEngine {
    gm_acq* h_ = nullptr;
    uint32_t n_prn_ = 0;
public:
    AcquisitionEngine(float fs, float f_if, uint32_t fft_size, const std::vector<float>& doppler_hz,
                      const std::vector<uint8_t>& prn_ids, uint32_t n_integrations = 10, float threshold = 7.0f,
                      int decision_mode = GM_DECIDE_REFERENCE) {
        gm_acq_cfg c{};
        c.decision_mode = decision_mode;
        c.fs = fs; c.f_if = f_if; c.fft_size = fft_size; c.n_integrations = n_integrations;
        c.n_bins = uint32_t(doppler_hz.size()); c.doppler_hz = doppler_hz.data();
        c.n_prn = uint32_t(prn_ids.size()); c.prn_ids = prn_ids.data(); c.threshold = threshold;
        n_prn_ = c.n_prn;
        check(gm_acq_create(&c, &h_), "AcquisitionEngine::new");
    }
    // caller-built tables, as search_satellite receives them (:160-161)
    AcquisitionEngine(float fs, uint32_t fft_size, const std::vector<DopplerShiftTable>& tables,
                      const std::vector<uint8_t>& prn_ids, uint32_t n_integrations) {
        std::vector<gm_c32> flat(tables.size() * size_t(fft_size));
        std::vector<float> freq(tables.size());
        for (size_t d = 0; d < tables.size(); ++d) {
            if (tables[d].table.size() < fft_size) throw Panic(GM_ERR_OUT_OF_RANGE, "DopplerShiftTable shorter than fft_size");
            for (uint32_t i = 0; i < fft_size; ++i) flat[d * fft_size + i] = {tables[d].table[i].real(), tables[d].table[i].imag()};
            freq[d] = tables[d].doppler_freq_hz;
        }
        gm_acq_cfg c{};
        c.fs = fs; c.fft_size = fft_size; c.n_integrations = n_integrations; c.n_bins = uint32_t(tables.size());
        c.tables = flat.data(); c.table_freq = freq.data(); c.n_prn = uint32_t(prn_ids.size()); c.prn_ids = prn_ids.data();
        n_prn_ = c.n_prn;
        check(gm_acq_create(&c, &h_), "AcquisitionEngine::new");
    }
    ~AcquisitionEngine() { gm_acq_destroy(h_); }
    AcquisitionEngine(const AcquisitionEngine&) = delete;
    AcquisitionEngine& operator=(const AcquisitionEngine&) = delete;
    gm_acq* handle() const { return h_; }

    std::vector<std::optional<AcquisitionResult>> search(const std::vector<Complex32>& samples_chunk, uint64_t local_tail,
                                                         uint64_t prn_mask = ~0ull) {
        std::vector<gm_acq_result> r(n_prn_);
        std::vector<uint8_t> f(n_prn_);
        check(gm_acq_search_c32(h_, reinterpret_cast<const gm_c32*>(samples_chunk.data()), samples_chunk.size(), local_tail,
                                prn_mask, r.data(), f.data()), "search_satellite");
        std::vector<std::optional<AcquisitionResult>> out(n_prn_);
        for (uint32_t i = 0; i < n_prn_; ++i) if (f[i]) out[i] = r[i];
        return out;
    }
    // run()'s snapshot + fan-out against the device ring (:297-313); nullopt while head < M*N (:299)
    std::optional<std::vector<std::optional<AcquisitionResult>>> search_ring(gm_ring* ring, uint64_t prn_mask = ~0ull,
                                                                             uint64_t* local_tail = nullptr) {
        std::vector<gm_acq_result> r(n_prn_);
        std::vector<uint8_t> f(n_prn_);
        const int st = gm_acq_search_ring(h_, ring, prn_mask, r.data(), f.data(), local_tail);
        if (st == GM_ERR_OUT_OF_RANGE) return std::nullopt;
        check(st, "search_ring");
        std::vector<std::optional<AcquisitionResult>> out(n_prn_);
        for (uint32_t i = 0; i < n_prn_; ++i) if (f[i]) out[i] = r[i];
        return out;
    }
    // finer_doppler (acquisition_bk.rs:215-302) on the snapshot of the last search: refined carrier (IF + Doppler) of
    // every found result, to a fraction of the coarse bin; entries of not-found workers are left as NaN
    std::vector<float> finer_doppler(const std::vector<std::optional<AcquisitionResult>>& results) {
        std::vector<gm_acq_result> r(results.size());
        std::vector<uint8_t> f(results.size(), 0);
        for (size_t i = 0; i < results.size(); ++i) if (results[i]) { r[i] = *results[i]; f[i] = 1; }
        std::vector<float> freq(results.size(), std::numeric_limits<float>::quiet_NaN());
        check(gm_acq_finer_doppler(h_, r.data(), f.data(), uint32_t(results.size()), freq.data(), nullptr, nullptr, nullptr),
              "finer_doppler");
        return freq;
    }
    std::vector<std::optional<AcquisitionResult>> search_i8(const std::vector<int8_t>& iq_interleaved, uint64_t local_tail,
                                                            uint64_t prn_mask = ~0ull) {
        std::vector<gm_acq_result> r(n_prn_);
        std::vector<uint8_t> f(n_prn_);
        check(gm_acq_search_i8(h_, iq_interleaved.data(), iq_interleaved.size() / 2, local_tail, prn_mask, r.data(), f.data()),
              "search_satellite");
        std::vector<std::optional<AcquisitionResult>> out(n_prn_);
        for (uint32_t i = 0; i < n_prn_; ++i) if (f[i]) out[i] = r[i];
        return out;
    }
};

// AcquisitionWorker::new(prn, fft_size, freq_sampling_hz) / search_satellite (:130-226): one PRN per object.
class AcquisitionWorker {
    uint8_t prn_; uint32_t fft_size_; float fs_;
    std::unique_ptr<AcquisitionEngine> eng_;
    const void* key_ = nullptr; size_t key_n_ = 0, key_m_ = 0;
public:
    AcquisitionWorker(uint8_t prn, size_t fft_size, float freq_sampling_hz) : prn_(prn), fft_size_(uint32_t(fft_size)), fs_(freq_sampling_hz) {
        if (prn < 1 || prn > 32) throw Panic(GM_ERR_OUT_OF_RANGE, "GPS_CA_CODE_32_PRN[prn - 1]");
    }
    std::optional<AcquisitionResult> search_satellite(const std::vector<Complex32>& samples_chunk,
                                                      const std::vector<DopplerShiftTable>& doppler_table, size_t local_tail,
                                                      size_t num_integrations) {
        if (!eng_ || key_ != doppler_table.data() || key_n_ != doppler_table.size() || key_m_ != num_integrations) {
            eng_ = std::make_unique<AcquisitionEngine>(fs_, fft_size_, doppler_table, std::vector<uint8_t>{prn_}, uint32_t(num_integrations));
            key_ = doppler_table.data(); key_n_ = doppler_table.size(); key_m_ = num_integrations;
        }
        return eng_->search(samples_chunk, local_tail)[0];
    }
};

// ---- utilities::multicast_ring_buffer (device mirror, :36-130)
class MulticastRingBuffer {
    gm_ring* h_ = nullptr;
public:
    explicit MulticastRingBuffer(size_t buf_size) { check(gm_ring_create(buf_size, &h_), "MulticastRingBuffer::new"); }
    ~MulticastRingBuffer() { gm_ring_destroy(h_); }
    MulticastRingBuffer(const MulticastRingBuffer&) = delete;
    gm_ring* handle() const { return h_; }
    void write_samples(const std::vector<Complex32>& s) { check(gm_ring_write_samples(h_, reinterpret_cast<const gm_c32*>(s.data()), s.size()), "write_samples"); }
    // producer side that never waits for the H2D copy (pinned staging, head published when the data is in HBM)
    void write_samples_async(const Complex32* s, size_t n) { check(gm_ring_write_samples_async(h_, reinterpret_cast<const gm_c32*>(s), n), "write_samples_async"); }
    void flush() { check(gm_ring_flush(h_), "flush"); }
    // notifier / condvar (:42-43): true once head >= required_idx
    bool wait_head(uint64_t required_idx, uint32_t timeout_ms) const { int r = 0; check(gm_ring_wait_head(h_, required_idx, timeout_ms, &r), "wait_head"); return r != 0; }
    uint64_t get_head() const { uint64_t h = 0; check(gm_ring_get_head(h_, &h), "get_head"); return h; }
    // what the asynchronous writer has ENQUEUED (>= get_head()): the head process_channels_async's passes are gated on
    uint64_t get_enqueued_head() const { uint64_t h = 0; check(gm_ring_get_enqueued_head(h_, &h), "get_enqueued_head"); return h; }
    void copy_to_slice(uint64_t start, std::vector<Complex32>& dest) const {
        check(gm_ring_copy_to_slice(h_, start, reinterpret_cast<gm_c32*>(dest.data()), dest.size()), "copy_to_slice");
    }
};

// ---- rf::frontend::DigitalFrontend (src/rf/frontend.rs:6-62)
class DigitalFrontend {
    gm_frontend* h_ = nullptr;
public:
    DigitalFrontend(float f_if, float fs_in, float fs_out) { check(gm_frontend_create(f_if, fs_in, fs_out, &h_), "DigitalFrontend::new"); }   // :19-30
    ~DigitalFrontend() { gm_frontend_destroy(h_); }
    DigitalFrontend(const DigitalFrontend&) = delete;
    void process_block(std::vector<float>& raw_floats) { check(gm_frontend_process_block(h_, raw_floats.data(), raw_floats.size()), "process_block"); }   // :33-62
    // rf_thread's block step (rf_thread.rs:43-48): process_block + write_samples, fused on the GPU, non-blocking
    void write_ring(MulticastRingBuffer& ring, const Complex32* block, size_t n) { check(gm_frontend_write_ring(h_, ring.handle(), block, n, GM_FMT_C32), "write_ring"); }
    void write_ring_i8(MulticastRingBuffer& ring, const int8_t* iq, size_t n) { check(gm_frontend_write_ring(h_, ring.handle(), iq, n, GM_FMT_I8_IQ), "write_ring"); }
    uint32_t debug_repairs() const { uint32_t n = 0; check(gm_frontend_debug_repairs(h_, &n), "debug_repairs"); return n; }   // runs of the speculative form done again
};

// ---- decoding::NavSyncStatus + nav_decoding's per-epoch step up to frame sync (src/decoding.rs:40-227, legacy)
class NavSyncStatus {
    gm_nav_sync* h_ = nullptr;
public:
    explicit NavSyncStatus(int mode = GM_NAV_FAITHFUL) { check(gm_nav_sync_create(mode, &h_), "NavSyncStatus::new"); }
    ~NavSyncStatus() { gm_nav_sync_destroy(h_); }
    NavSyncStatus(const NavSyncStatus&) = delete;
    gm_nav_sync* handle() const { return h_; }
    gm_nav_status update(float old_i_prompt, float i_prompt, uint64_t cnt, uint64_t buff_loc = 0) {
        gm_nav_status st{};
        check(gm_nav_sync_update(h_, old_i_prompt, i_prompt, cnt, buff_loc, &st), "nav_decoding");
        return st;
    }
    std::vector<int8_t> frame_bits() const {
        size_t n = 0;
        check(gm_nav_sync_frame_bits(h_, nullptr, 0, &n), "frame_bits");
        std::vector<int8_t> b(n);
        if (n) check(gm_nav_sync_frame_bits(h_, b.data(), n, &n), "frame_bits");
        return b;
    }
};

// ---- tracking::do_tracking (src/tracking/do_tracking.rs)
struct LoopFilter {                                            // :52-71
    float tau1 = 0, tau2 = 0;
    LoopFilter(float noise_bw, float dumping_ratio, float gain) { check(gm_loop_filter_new(noise_bw, dumping_ratio, gain, &tau1, &tau2), "LoopFilter::new"); }
    float update(float d_err, float err, float dt) const { return gm_loop_filter_update(tau1, tau2, d_err, err, dt); }
};
enum class TrackingMessageKind { SatelliteLost, SatelliteLocked };   // :47-50
struct TrackingMessage { TrackingMessageKind kind; uint8_t prn; };
using CorrelatorOut = gm_trk_out;   // (i_p,q_p,i_e,q_e,i_l,q_l [,very early / very late])

class TrackingManager;              // TrackingManager::new (:336-348)
class TrackingChannel {             // a view of one channel of the manager's handle (:88-327)
    gm_trk* h_; uint32_t id_;
    friend class TrackingManager;
    TrackingChannel(gm_trk* h, uint32_t id) : h_(h), id_(id) {}
public:
    uint32_t id() const { return id_; }
    gm_trk_state state() const { gm_trk_state s; check(gm_trk_get_state(h_, id_, &s), "state"); return s; }
    void start(const AcquisitionResult& r) { check(gm_trk_start(h_, id_, &r), "TrackingChannel::start"); }      // :148-154
    bool is_active() const { return state().active != 0; }                                                       // :156-158
    void reset() { check(gm_trk_reset(h_, id_), "TrackingChannel::reset"); }                                     // :311-327
    float get_ca_chip(float phase) const { float c; check(gm_trk_get_ca_chip(h_, id_, phase, &c), "get_ca_chip"); return c; }   // :274-277
    CorrelatorOut early_late_correlation(const std::vector<Complex32>& data_samples) {                           // :231-272
        gm_trk_out o; check(gm_trk_correlate(h_, id_, reinterpret_cast<const gm_c32*>(data_samples.data()), data_samples.size(), &o), "early_late_correlation"); return o;
    }
    std::optional<TrackingMessage> do_work(const std::vector<Complex32>& data_samples, CorrelatorOut* out = nullptr) {   // :183-210
        gm_trk_out o; uint8_t lost = 0, prn = 0;
        check(gm_trk_do_work(h_, id_, reinterpret_cast<const gm_c32*>(data_samples.data()), data_samples.size(), &o, &lost, &prn), "do_work");
        if (out) *out = o;
        if (lost) return TrackingMessage{TrackingMessageKind::SatelliteLost, prn};
        return std::nullopt;
    }
};

class TrackingManager {
    gm_trk* h_ = nullptr; uint32_t n_ = 0;
public:
    std::vector<TrackingChannel> channels;
    TrackingManager(float fs, uint32_t n_channels = 15, int code_index_mode = GM_CODE_INDEX_FAITHFUL, uint32_t n_arms = 3,
                    bool strict_libm = false, bool strict_sum_order = false, bool share_device = false) : n_(n_channels) {
        gm_trk_cfg c{}; c.fs = fs; c.n_channels = n_channels; c.n_arms = n_arms; c.code_index_mode = code_index_mode;
        c.strict_libm = strict_libm ? 1 : 0; c.strict_sum_order = strict_sum_order ? 1 : 0; c.share_device = share_device ? 1 : 0;
        check(gm_trk_create(&c, &h_), "TrackingManager::new");
        for (uint32_t i = 0; i < n_channels; ++i) channels.push_back(TrackingChannel(h_, i));
    }
    ~TrackingManager() { gm_trk_destroy(h_); }
    TrackingManager(const TrackingManager&) = delete;
    // process_channels' fan-out (:364-371), up to max_epochs passes; returns passes in which a channel ran
    uint32_t process_channels(MulticastRingBuffer& ring, uint32_t max_epochs, std::vector<CorrelatorOut>* outs = nullptr,
                              std::vector<uint8_t>* processed = nullptr, std::vector<uint8_t>* lost = nullptr) {
        const size_t n = size_t(max_epochs) * n_;
        if (outs) outs->resize(n);
        if (processed) processed->resize(n);
        if (lost) lost->resize(n);
        uint32_t done = 0;
        check(gm_trk_update_all(h_, ring.handle(), max_epochs, outs ? outs->data() : nullptr, processed ? processed->data() : nullptr,
                                lost ? lost->data() : nullptr, &done), "process_channels");
        return done;
    }
    // the same passes WITHOUT a host wait (ABI 6): ordered on the device behind everything the ring's asynchronous writer has enqueued
    // (the Condvar wait of do_tracking.rs:392-406 as an event on the ring's stream); the results are collected later, by ticket
    uint64_t process_channels_async(MulticastRingBuffer& ring, uint32_t max_epochs) {
        uint64_t ticket = 0;
        check(gm_trk_update_all_async(h_, ring.handle(), max_epochs, &ticket), "process_channels_async");
        pending_[ticket] = max_epochs;
        return ticket;
    }
    // false while the call is still running (wait = false); else the passes in which a channel ran through *done
    // states: the channel records as they stood behind THAT call's passes (a device-side snapshot in stream order)
    bool collect(uint64_t ticket, bool wait, uint32_t* done = nullptr, std::vector<CorrelatorOut>* outs = nullptr,
                 std::vector<uint8_t>* processed = nullptr, std::vector<uint8_t>* lost = nullptr,
                 std::vector<gm_trk_state>* states = nullptr) {
        const auto it = pending_.find(ticket);
        if (it == pending_.end()) throw Panic(GM_ERR_INVALID_ARG, "collect: no such ticket");
        const size_t n = size_t(it->second) * n_;
        if (outs) outs->resize(n);
        if (processed) processed->resize(n);
        if (lost) lost->resize(n);
        if (states) states->resize(n_);
        int ready = 0; uint32_t d = 0;
        const int rc = gm_trk_collect(h_, ticket, wait ? 1 : 0, outs ? outs->data() : nullptr, processed ? processed->data() : nullptr,
                                      lost ? lost->data() : nullptr, states ? states->data() : nullptr, &d, &ready);
        if (rc != GM_OK) { pending_.erase(it); check(rc, "collect"); }      // a failed collect has consumed the ticket (header)
        if (!ready) return false;
        pending_.erase(it);
        if (done) *done = d;
        return true;
    }
    size_t tickets_in_flight() const { return pending_.size(); }
    uint32_t n_channels() const { return n_; }
    uint32_t pass_count(uint64_t ticket) const { const auto it = pending_.find(ticket); return it == pending_.end() ? 0 : it->second; }
    // every channel's record in one synchronisation + one copy (ABI 7)
    std::vector<gm_trk_state> states() const { std::vector<gm_trk_state> s(n_); check(gm_trk_get_states(h_, s.data()), "states"); return s; }
private:
    std::map<uint64_t, uint32_t> pending_;      // ticket -> its pass count
};

// ---- the two stage drivers: do_acquisition::run (do_acquisition.rs:241-327) and
// do_tracking::run (do_tracking.rs:384-415), talking through the reference's two unbounded channels
// (main.rs:183-184).  The reference loops forever; here a StageControl stops the loops and can scale the pacing.
template <class T> class Channel {           // crossbeam_channel::unbounded()
    std::deque<T> q_;
    mutable std::mutex m_;
public:
    bool send(T v) { std::lock_guard<std::mutex> g(m_); q_.push_back(std::move(v)); return true; }
    std::optional<T> try_recv() {
        std::lock_guard<std::mutex> g(m_);
        if (q_.empty()) return std::nullopt;
        T v = std::move(q_.front()); q_.pop_front();
        return v;
    }
    size_t len() const { std::lock_guard<std::mutex> g(m_); return q_.size(); }
};

struct StageControl {
    std::atomic<bool> stop{false};
    double pacing_scale = 1.0;            // 1.0 = the reference's 500 / 1000 / 2000 ms search intervals (:59-63)
    std::atomic<uint64_t> acq_rounds{0}, trk_passes{0};
    // round 6: what a feeder that replays faster than real time needs for flow control, and what a harness reports
    std::atomic<uint64_t> trk_collected_head{0};   // ring head up to which the tracking stage's passes have run AND been collected
    std::atomic<uint64_t> channel_epochs{0};       // channel x code-period updates collected so far
    std::atomic<uint64_t> acq_ns{0}, fine_ns{0}, trk_ns{0}, hook_ns{0};   // wall clock spent inside the stages' calls
    std::atomic<bool> trk_finished{false};         // the tracking stage has collected its last call (its handles are torn down after this)
    std::atomic<int> stages_ready{0};              // + 1 by each stage once its handles exist (tables, code spectra, result slots): a
                                                   // replay that outruns real time starts its feeder when both stages are listening
};
inline uint64_t stage_ns(std::chrono::steady_clock::time_point t0) {
    return uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
}

struct AcquisitionRunOptions {            // the reference's compile-time constants (:20-23) as run-time values
    float freq_search_hz = 14e3f;         // FREQ_SEARCH_ACQUISITION_HZ
    float freq_step_hz = 500.0f;          // FREQ_SEARCH_STEP_HZ
    uint32_t long_samples_length = 10;    // LONG_SAMPLES_LENGTH (ms)
    int decision_mode = GM_DECIDE_REFERENCE;
    // The reference paces its rounds on the wall clock (:292-295), which IS signal time for a live source.  A replay that runs
    // faster (or slower) than real time keeps the reference's cadence per second of SIGNAL with this switch: the interval is
    // counted in samples of the ring's head.  first_round_signal_ms < 0: the reference's behaviour (the first round after one
    // interval, `last_run = Instant::now()` at :274); >= 0: the first round once that much signal is in the ring.
    bool pace_on_signal_time = false;
    double first_round_signal_ms = -1.0;
    // refine every hit's carrier (finer_doppler, acquisition_bk.rs:215-302) before it goes to tracking
    bool fine_doppler = false;
    std::function<void(uint64_t head, const std::vector<std::optional<AcquisitionResult>>&)> on_round;   // after every round (may be empty)
};

inline void run_acquisition(MulticastRingBuffer& multi_buffer, float freq_sampling_hz, float f_if,
                            Channel<AcquisitionResult>& to_tracking, Channel<TrackingMessage>& from_tracking,
                            StageControl& ctl, const AcquisitionRunOptions& opt = {}) {
    const size_t capacity = size_t(uint16_t(opt.freq_search_hz) / uint16_t(opt.freq_step_hz)) + 1;        // :248
    const uint32_t fft_size = uint32_t(std::lround(freq_sampling_hz / (1.023e6f / 1023.0f)));              // :249-251
    std::vector<float> doppler(capacity);
    for (size_t i = 0; i < capacity; ++i) doppler[i] = -opt.freq_search_hz / 2.0f + float(i) * opt.freq_step_hz;   // :253-255
    std::vector<uint8_t> prns(PRN_SEARCH_ACQUISITION_TOTAL);
    for (uint8_t p = 0; p < PRN_SEARCH_ACQUISITION_TOTAL; ++p) prns[p] = uint8_t(p + 1);
    AcquisitionEngine workers(freq_sampling_hz, f_if, fft_size, doppler, prns, opt.long_samples_length, 7.0f,
                              opt.decision_mode);                                                            // :252-271
    std::set<uint8_t> active_prns;
    AcquisitionManager acq_manager;
    ctl.stages_ready++;
    auto last_run = std::chrono::steady_clock::now();
    const auto ms = [&](double v) { return std::chrono::duration<double, std::milli>(v * ctl.pacing_scale); };
    // signal-time pacing: the head (in samples) at which the next round is due
    const double samples_per_ms = double(freq_sampling_hz) * 1e-3;
    uint64_t next_due = opt.first_round_signal_ms >= 0.0 ? uint64_t(opt.first_round_signal_ms * samples_per_ms) : 0;
    bool first = true;
    while (!ctl.stop.load()) {
        while (auto msg = from_tracking.try_recv()) {                                                       // :278-287
            if (msg->kind == TrackingMessageKind::SatelliteLost) active_prns.erase(msg->prn);
            else active_prns.insert(msg->prn);
        }
        acq_manager.update_mode(active_prns.size());                                                        // :289
        auto [interval_ms, mask] = acq_manager.get_pacing_and_list(active_prns);                            // :290
        if (opt.pace_on_signal_time) {
            if (first && opt.first_round_signal_ms < 0.0) next_due = uint64_t(double(interval_ms) * ctl.pacing_scale * samples_per_ms);
            first = false;
            if (int64_t(multi_buffer.get_head() - next_due) < 0) {                                          // :292-295 in samples
                multi_buffer.wait_head(next_due, 2);         // the ring's Condvar, bounded so that `stop` is seen
                continue;
            }
        } else if (std::chrono::steady_clock::now() - last_run < ms(double(interval_ms))) {                 // :292-295
            std::this_thread::sleep_for(ms(50.0));
            continue;
        }
        uint64_t local_tail = 0;
        auto t0 = std::chrono::steady_clock::now();
        auto results = workers.search_ring(multi_buffer.handle(), uint64_t(mask), &local_tail);             // :297-313
        ctl.acq_ns += stage_ns(t0);
        if (!results) { std::this_thread::sleep_for(std::chrono::milliseconds(1)); continue; }              // :324-326
        const uint64_t head_of_round = local_tail + uint64_t(opt.long_samples_length) * fft_size;
        if (opt.fine_doppler) {
            // hits on satellites that are tracked already go nowhere (:315-320 would hand them over again; the mask
            // normally excludes them) — refine only what is about to start a channel
            bool any = false;
            for (auto& r : *results) { if (r && active_prns.count(r->prn)) r.reset(); any = any || bool(r); }
            if (any) {
                t0 = std::chrono::steady_clock::now();
                const std::vector<float> fine = workers.finer_doppler(*results);
                for (size_t i = 0; i < results->size(); ++i)
                    if ((*results)[i] && std::isfinite(fine[i])) (*results)[i]->carrier_freq = fine[i];
                ctl.fine_ns += stage_ns(t0);
            }
        }
        for (auto& r : *results)                                                                            // :315-320
            if (r && to_tracking.send(*r)) active_prns.insert(r->prn);
        if (opt.on_round) opt.on_round(head_of_round, *results);
        last_run = std::chrono::steady_clock::now();                                                        // :322
        // the NEXT interval is the one the manager gives for the satellites just handed over (as the loop head would compute it)
        acq_manager.update_mode(active_prns.size());
        next_due = head_of_round + uint64_t(double(acq_manager.get_pacing_and_list(active_prns).first) * ctl.pacing_scale * samples_per_ms);
        ctl.acq_rounds++;
    }
}

// What one collected call of the tracking stage hands to a consumer behind it (nav-bit accumulation, logging): the passes'
// correlator outputs and flags, [passes][n_channels], and the ring head the call was gated on.
struct EpochBlock {
    uint32_t passes = 0, n_channels = 0;
    uint64_t head = 0;
    const CorrelatorOut* outs = nullptr;
    const uint8_t* processed = nullptr;
    const uint8_t* lost = nullptr;
    const gm_trk_state* states = nullptr;       // [n_channels], as they stood behind this call's passes (NULL on the synchronous path)
    const uint8_t* channel_prn = nullptr;       // [n_channels] the PRN each channel was started with (0: never started)
};

struct TrackingRunOptions {
    int code_index_mode = GM_CODE_INDEX_FIXED;
    uint32_t n_channels = 15;                   // NUM_OF_CHANNELS (:18)
    // true (default): no host wait per block — process_channels is ENQUEUED behind whatever the ring's writer has enqueued
    // (gm_trk_update_all_async: the Condvar wait of :392-406 as an event on the ring's stream) and collected a block or two later;
    // false: the synchronous pass per loop turn (rounds 3-5)
    bool async_tickets = true;
    uint32_t max_in_flight = 6;                 // tickets not yet collected (the library holds 8 result slots)
    uint32_t max_passes_per_call = 256;         // the result slots are sized for this once, before the loop (they cannot grow under tickets in flight)
    bool share_device = true;                   // a receiver: leave CUs to the front-end and the acquisition dwell (gm_trk_cfg.share_device)
    bool drain_on_stop = true;                  // after `stop`: hand over pending acquisitions and run every whole code period the ring still holds
    std::function<void(const EpochBlock&)> on_epochs;      // every collected call (may be empty)
    std::vector<gm_trk_state>* final_states = nullptr;
};

namespace detail {
// passes needed so that every active channel reaches `head`: channels advance one code period per pass
inline uint32_t passes_to_head(const std::vector<gm_trk_state>& st, const std::vector<uint8_t>& busy, uint64_t head,
                               const std::vector<uint64_t>& covered) {
    uint64_t worst = 0;
    for (size_t c = 0; c < st.size(); ++c) {
        if (!busy[c] || !st[c].num_samples_per_code) continue;
        const uint64_t n = st[c].num_samples_per_code;
        uint64_t idx = st[c].next_sample_index;
        // passes already enqueued for this channel (not yet collected) take it to the last whole period before covered[c]
        if (int64_t(covered[c] - idx) > 0) idx += ((covered[c] - idx) / n) * n;
        if (int64_t(head - idx) > 0) worst = std::max<uint64_t>(worst, (head - idx) / n);
    }
    return uint32_t(std::min<uint64_t>(worst + 2, 4095));    // + 2: the period length moves by a sample now and then (code_rate)
}
}  // namespace detail

inline void run_tracking(MulticastRingBuffer& multi_ring_buf, Channel<AcquisitionResult>& acq_to_trk,
                         Channel<TrackingMessage>& trk_to_acq, float fs, StageControl& ctl, const TrackingRunOptions& opt) {
    const uint32_t n_channels = opt.n_channels;
    TrackingManager manager(fs, n_channels, opt.code_index_mode, 3, false, false, opt.share_device);        // :390
    std::vector<uint8_t> channel_prn(n_channels, 0), busy(n_channels, 0), lost, processed;
    std::vector<CorrelatorOut> outs;
    std::vector<gm_trk_state> states = manager.states(), snap;       // host view of the channels: exact whenever nothing is in flight
    struct Pending { uint64_t ticket, head, seq; };
    std::deque<Pending> tickets;
    uint64_t issued = 0;                                             // calls enqueued so far
    std::vector<uint64_t> start_seq(n_channels, 0);                  // `issued` when the channel was started: older snapshots predate it
    std::vector<uint64_t> covered(n_channels, 0);                    // the head up to which passes for this channel have been enqueued
    uint64_t planned_head = multi_ring_buf.get_enqueued_head();      // the head the most recent enqueued call was gated on
    ctl.trk_collected_head = planned_head;
    constexpr uint32_t LOOP_MS = 10;                                                                        // :29
    const bool want_outs = bool(opt.on_epochs);
    const uint32_t max_passes = std::max<uint32_t>(opt.max_passes_per_call, 3);
    if (opt.async_tickets)      // one empty call of the largest size: pinned result slots, copy path and events exist before the first block
        manager.collect(manager.process_channels_async(multi_ring_buf, max_passes), true);
    ctl.stages_ready++;

    // process_channels' first half (:352-362): hand new acquisitions to idle channels
    const auto take_acquisitions = [&] {
        while (auto msg = acq_to_trk.try_recv()) {
            for (uint32_t c = 0; c < n_channels; ++c) {
                if (busy[c]) continue;
                trk_to_acq.send(TrackingMessage{TrackingMessageKind::SatelliteLocked, msg->prn});
                manager.channels[c].start(*msg);       // (waits for the passes in flight: a hand-over, a few per minute)
                states[c] = manager.channels[c].state();
                busy[c] = 1; channel_prn[c] = msg->prn;
                start_seq[c] = issued; covered[c] = states[c].next_sample_index;
                break;
            }
        }
    };
    const auto report = [&](uint32_t passes, uint32_t done, uint64_t head, const gm_trk_state* st) {
        uint64_t ran = 0;
        for (uint32_t e = 0; e < passes; ++e)
            for (uint32_t c = 0; c < n_channels; ++c) {
                ran += processed[size_t(e) * n_channels + c];
                if (lost[size_t(e) * n_channels + c]) {
                    // the reference's message carries prn 0 (reset() runs first, :199-201); FIXED reports the real one
                    const uint8_t prn = opt.code_index_mode == GM_CODE_INDEX_FAITHFUL ? uint8_t(0) : channel_prn[c];
                    trk_to_acq.send(TrackingMessage{TrackingMessageKind::SatelliteLost, prn});
                    busy[c] = 0;
                }
            }
        ctl.trk_passes += done; ctl.channel_epochs += ran;
        if (opt.on_epochs) {
            const auto t0 = std::chrono::steady_clock::now();
            EpochBlock b; b.passes = passes; b.n_channels = n_channels; b.head = head; b.outs = outs.data();
            b.processed = processed.data(); b.lost = lost.data(); b.states = st; b.channel_prn = channel_prn.data();
            opt.on_epochs(b);
            ctl.hook_ns += stage_ns(t0);
        }
        ctl.trk_collected_head = head;
    };
    // collect what is ready (everything, waiting, when `all`); true if a call was collected
    const auto drain = [&](bool all) {
        bool got = false;
        while (!tickets.empty()) {
            const auto t0 = std::chrono::steady_clock::now();
            const uint32_t passes = manager.pass_count(tickets.front().ticket);
            uint32_t done = 0;
            const bool ready = manager.collect(tickets.front().ticket, all || tickets.size() >= opt.max_in_flight, &done,
                                               want_outs ? &outs : nullptr, &processed, &lost, &snap);
            ctl.trk_ns += stage_ns(t0);
            if (!ready) break;
            const uint64_t head = tickets.front().head, seq = tickets.front().seq;
            tickets.pop_front();
            // the collected snapshot is the truth for every channel that was not (re)started after that call was enqueued
            for (uint32_t c = 0; c < n_channels; ++c)
                if (seq >= start_seq[c]) states[c] = snap[c];
            report(passes, done, head, snap.data());
            got = true;
        }
        return got;
    };

    bool stopping = false;
    for (;;) {
        if (!stopping && ctl.stop.load()) { stopping = true; if (!opt.drain_on_stop) break; }
        take_acquisitions();
        bool any_active = false;
        for (uint32_t c = 0; c < n_channels; ++c) any_active = any_active || busy[c];
        bool progressed = false;
        if (opt.async_tickets) {
            const uint64_t head = multi_ring_buf.get_enqueued_head();
            if (any_active && tickets.size() < opt.max_in_flight) {
                const uint32_t passes = std::min(detail::passes_to_head(states, busy, head, covered), max_passes);
                if (passes > 2 || (stopping && tickets.empty())) {      // > 2: at least one whole code period is there for some channel
                    const auto t0 = std::chrono::steady_clock::now();
                    tickets.push_back(Pending{manager.process_channels_async(multi_ring_buf, passes), head, issued++});
                    ctl.trk_ns += stage_ns(t0);
                    planned_head = head;
                    for (uint32_t c = 0; c < n_channels; ++c) {         // how far this call takes each channel (all the way unless capped)
                        if (!busy[c] || !states[c].num_samples_per_code) continue;
                        const uint64_t n = states[c].num_samples_per_code;
                        uint64_t idx = states[c].next_sample_index;
                        if (int64_t(covered[c] - idx) > 0) idx += ((covered[c] - idx) / n) * n;
                        const uint64_t reach = idx + uint64_t(passes) * n;
                        covered[c] = int64_t(reach - head) < 0 ? reach : head;
                    }
                    progressed = true;
                }
            }
            const size_t before = tickets.size();
            const uint64_t passes_before = ctl.trk_passes.load();
            progressed = drain(stopping) || progressed;
            if (stopping) {
                // done when a drained call found nothing left to run and nothing new is waiting
                if (tickets.empty() && before && ctl.trk_passes.load() == passes_before && acq_to_trk.len() == 0) break;
                if (!any_active && acq_to_trk.len() == 0) break;
                continue;
            }
            if (!any_active) ctl.trk_collected_head = head;       // nothing to track: the feeder is not held back
        } else {
            // ... every active channel with a whole code period available runs update(), up to LOOP_MS passes, synchronously
            const auto t0 = std::chrono::steady_clock::now();
            const uint64_t head = multi_ring_buf.get_head();
            const uint32_t done = manager.process_channels(multi_ring_buf, LOOP_MS, want_outs ? &outs : nullptr, &processed, &lost);
            ctl.trk_ns += stage_ns(t0);
            report(LOOP_MS, done, head, nullptr);
            progressed = done != 0;
            if (stopping) { if (!done && acq_to_trk.len() == 0) break; continue; }
            if (!done) states = manager.states();                 // next_tracking_index (:373-381) needs the current records
        }
        if (!progressed && opt.async_tickets && !tickets.empty()) {
            // calls in flight: their completion and the writer's next ENQUEUED block are what this thread reacts to — neither moves
            // the published head's Condvar, so look again shortly (one ticket per writer block keeps the collected head a few
            // blocks behind the writer instead of a ticket's worth of blocks)
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        } else if (!progressed) {                                                       // Condvar wait (:392-406)
            uint64_t required_idx = 0; bool any = false;                                // next_tracking_index (:373-381)
            for (uint32_t c = 0; c < n_channels; ++c) {
                if (!busy[c] || !states[c].active) continue;
                uint64_t need = states[c].next_sample_index + states[c].num_samples_per_code;
                if (opt.async_tickets && int64_t(planned_head - need) >= 0) need = planned_head + states[c].num_samples_per_code;
                if (!any || int64_t(need - required_idx) < 0) required_idx = need;
                any = true;
            }
            if (any) multi_ring_buf.wait_head(required_idx, 1);     // bounded (1 ms) so that `stop` and new acquisitions are seen
            else std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    while (!tickets.empty()) drain(true);
    ctl.trk_finished = true;
    if (opt.final_states) *opt.final_states = manager.states();
}

// the signature of rounds 3-5 (tests/cpp/test_host_api.cpp): the default options, i.e. the ticket loop
inline void run_tracking(MulticastRingBuffer& multi_ring_buf, Channel<AcquisitionResult>& acq_to_trk,
                         Channel<TrackingMessage>& trk_to_acq, float fs, StageControl& ctl,
                         int code_index_mode = GM_CODE_INDEX_FIXED, uint32_t n_channels = 15,
                         std::vector<gm_trk_state>* final_states = nullptr) {
    TrackingRunOptions opt;
    opt.code_index_mode = code_index_mode; opt.n_channels = n_channels; opt.final_states = final_states;
    run_tracking(multi_ring_buf, acq_to_trk, trk_to_acq, fs, ctl, opt);
}

// ---- crate root FFT<T> / RealFFT<T> (src/fft.rs:5-56), T = f32
class FFT {
    size_t len_;
public:
    explicit FFT(size_t len) : len_(len) {}
    std::vector<Complex32> execute(std::vector<Complex32>& input) const {   // in place, returns a copy (:21-25)
        check(gm_fft_c2c_f32(len_, 0, reinterpret_cast<gm_c32*>(input.data()), input.size() / len_), "FFT::execute");
        return input;
    }
    std::vector<float> power_spectrum(std::vector<Complex32>& input) const {
        std::vector<float> p(len_);
        check(gm_fft_power_spectrum_f32(len_, reinterpret_cast<gm_c32*>(input.data()), p.data()), "FFT::power_spectrum");
        return p;
    }
};
class RealFFT {
    size_t len_;
public:
    explicit RealFFT(size_t len) : len_(len) {}
    std::vector<Complex32> execute(const std::vector<float>& input) const {
        std::vector<Complex32> out(len_ / 2 + 1);
        check(gm_rfft_f32(len_, input.data(), reinterpret_cast<gm_c32*>(out.data())), "RealFFT::execute");
        return out;
    }
};

}  // namespace gnss
