// gnss_sdr.hpp — C++ host-side mirror of the reference crate's acquisition / tracking API over the C ABI
// (include/gnss_mi355x.h).  Header-only; link with libgnss_mi355x.so.  Type and method names follow the
// reference (file:line in the comments) so call sites port one to one; errors the reference turns into panics
// become gnss::Panic exceptions HERE (above the ABI — nothing unwinds across it).
#pragma once
#include <atomic>
#include <chrono>
#include <complex>
#include <deque>
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
};

// ---- decoding::NavSyncStatus + nav_decoding's per-epoch step up to frame sync (src/decoding.rs:40-227, legacy)
class NavSyncStatus {
    gm_nav_sync* h_ = nullptr;
public:
    explicit NavSyncStatus(int mode = GM_NAV_FAITHFUL) { check(gm_nav_sync_create(mode, &h_), "NavSyncStatus::new"); }
    ~NavSyncStatus() { gm_nav_sync_destroy(h_); }
    NavSyncStatus(const NavSyncStatus&) = delete;
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
    bool collect(uint64_t ticket, bool wait, uint32_t* done = nullptr, std::vector<CorrelatorOut>* outs = nullptr,
                 std::vector<uint8_t>* processed = nullptr, std::vector<uint8_t>* lost = nullptr) {
        const auto it = pending_.find(ticket);
        if (it == pending_.end()) throw Panic(GM_ERR_INVALID_ARG, "collect: no such ticket");
        const size_t n = size_t(it->second) * n_;
        if (outs) outs->resize(n);
        if (processed) processed->resize(n);
        if (lost) lost->resize(n);
        int ready = 0; uint32_t d = 0;
        check(gm_trk_collect(h_, ticket, wait ? 1 : 0, outs ? outs->data() : nullptr, processed ? processed->data() : nullptr,
                             lost ? lost->data() : nullptr, &d, &ready), "collect");
        if (!ready) return false;
        pending_.erase(it);
        if (done) *done = d;
        return true;
    }
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
};

struct AcquisitionRunOptions {            // the reference's compile-time constants (:20-23) as run-time values
    float freq_search_hz = 14e3f;         // FREQ_SEARCH_ACQUISITION_HZ
    float freq_step_hz = 500.0f;          // FREQ_SEARCH_STEP_HZ
    uint32_t long_samples_length = 10;    // LONG_SAMPLES_LENGTH (ms)
    int decision_mode = GM_DECIDE_REFERENCE;
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
    auto last_run = std::chrono::steady_clock::now();
    const auto ms = [&](double v) { return std::chrono::duration<double, std::milli>(v * ctl.pacing_scale); };
    while (!ctl.stop.load()) {
        while (auto msg = from_tracking.try_recv()) {                                                       // :278-287
            if (msg->kind == TrackingMessageKind::SatelliteLost) active_prns.erase(msg->prn);
            else active_prns.insert(msg->prn);
        }
        acq_manager.update_mode(active_prns.size());                                                        // :289
        auto [interval_ms, mask] = acq_manager.get_pacing_and_list(active_prns);                            // :290
        if (std::chrono::steady_clock::now() - last_run < ms(double(interval_ms))) {                        // :292-295
            std::this_thread::sleep_for(ms(50.0));
            continue;
        }
        uint64_t local_tail = 0;
        auto results = workers.search_ring(multi_buffer.handle(), uint64_t(mask), &local_tail);             // :297-313
        if (!results) { std::this_thread::sleep_for(std::chrono::milliseconds(1)); continue; }              // :324-326
        for (auto& r : *results)                                                                            // :315-320
            if (r && to_tracking.send(*r)) active_prns.insert(r->prn);
        last_run = std::chrono::steady_clock::now();                                                        // :322
        ctl.acq_rounds++;
    }
}

inline void run_tracking(MulticastRingBuffer& multi_ring_buf, Channel<AcquisitionResult>& acq_to_trk,
                         Channel<TrackingMessage>& trk_to_acq, float fs, StageControl& ctl,
                         int code_index_mode = GM_CODE_INDEX_FIXED, uint32_t n_channels = 15,
                         std::vector<gm_trk_state>* final_states = nullptr) {
    TrackingManager manager(fs, n_channels, code_index_mode);                                               // :390
    std::vector<uint8_t> channel_prn(n_channels, 0), lost;
    constexpr uint32_t LOOP_MS = 10;                                                                        // :29
    while (!ctl.stop.load()) {
        // process_channels (:351-371): hand new acquisitions to idle channels ...
        while (auto msg = acq_to_trk.try_recv()) {
            for (uint32_t c = 0; c < n_channels; ++c) {
                if (!manager.channels[c].is_active()) {
                    trk_to_acq.send(TrackingMessage{TrackingMessageKind::SatelliteLocked, msg->prn});
                    manager.channels[c].start(*msg);
                    channel_prn[c] = msg->prn;
                    break;
                }
            }
        }
        // ... then every active channel with a whole code period available runs update(), up to LOOP_MS passes
        const uint32_t done = manager.process_channels(multi_ring_buf, LOOP_MS, nullptr, nullptr, &lost);
        for (uint32_t e = 0; e < LOOP_MS; ++e)
            for (uint32_t c = 0; c < n_channels; ++c)
                if (lost[size_t(e) * n_channels + c]) {
                    // the reference's message carries prn 0 (reset() runs first, :199-201); FIXED reports the real one
                    const uint8_t prn = code_index_mode == GM_CODE_INDEX_FAITHFUL ? uint8_t(0) : channel_prn[c];
                    trk_to_acq.send(TrackingMessage{TrackingMessageKind::SatelliteLost, prn});
                }
        ctl.trk_passes += done;
        if (done == 0) {                                                                // Condvar wait (:392-406)
            uint64_t required_idx = 0; bool any = false;                                // next_tracking_index (:373-381)
            for (uint32_t c = 0; c < n_channels; ++c) {
                const gm_trk_state st = manager.channels[c].state();
                if (!st.active) continue;
                const uint64_t need = st.next_sample_index + st.num_samples_per_code;
                if (!any || need < required_idx) required_idx = need;
                any = true;
            }
            if (any) multi_ring_buf.wait_head(required_idx, 2);     // bounded so that `stop` and new acquisitions are seen
            else std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    if (final_states) {
        final_states->resize(n_channels);
        for (uint32_t c = 0; c < n_channels; ++c) (*final_states)[c] = manager.channels[c].state();
    }
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
