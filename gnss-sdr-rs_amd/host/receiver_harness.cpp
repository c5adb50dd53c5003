// receiver_harness.cpp — the receiver main.rs:182-227 wires, built from the DELIVERABLE stage drivers of host/gnss_sdr.hpp
// (gnss::run_acquisition, gnss::run_tracking, DigitalFrontend::write_ring, NavSyncStatus) behind a small extern "C" surface, so
// that bench.py's `receiver` leg and the tests time and check the C++ drop-in itself instead of a Python re-statement of its
// loops.  Host code only: g++, links libgnss_mi355x.so through the C ABI; no kernels here.
//
//   gmrx_receiver_run   feeder (the caller's thread: rf_thread's block step, rf/rf_thread.rs:43-48) + do_acquisition::run
//                       (do_acquisition.rs:241-327) + do_tracking::run (do_tracking.rs:384-415) on threads of their own,
//                       talking through the two unbounded channels (main.rs:183-184); nav-bit accumulation behind tracking.
//   gmrx_tracking_ab    the tracking stage driver alone on a pre-loaded ring with given acquisition results, once with the
//                       synchronous loop and once with the ticket loop: final channel states of both (they must be equal).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "gnss_sdr.hpp"

using namespace gnss;

extern "C" {

typedef struct {
    int32_t device;                    // HIP device of every handle (gm_init)
    float fs, f_if;                    // sample rate, IF of the stream handed to the front-end
    float freq_search_hz, freq_step_hz;      // FREQ_SEARCH_ACQUISITION_HZ / FREQ_SEARCH_STEP_HZ (do_acquisition.rs:20-21)
    uint32_t n_integrations;           // LONG_SAMPLES_LENGTH
    uint32_t n_channels;               // NUM_OF_CHANNELS
    uint32_t block_samples;            // feeder block (<= 2^19: one staging slot of the ring's writer)
    uint32_t ring_log2;                // ring size = 2^ring_log2 samples
    int32_t decision_mode, code_index_mode, nav_mode;
    int32_t fine_doppler;              // 1: refine every hit's carrier before the hand-over
    int32_t async_tickets;             // 1: the ticket loop (no host wait per block); 0: the synchronous loop
    double first_round_signal_ms;      // < 0: the reference's (one interval first)
    uint32_t pre_samples;              // written (and flushed) before the clock starts: the ring's one-time start-up
    uint32_t warmup_calls;             // asynchronous tracking calls on a scratch handle before the clock starts
} gmrx_cfg;

typedef struct {
    uint8_t prn, active, bit_sync, frame_sync;
    uint32_t lost_counter;
    float carrier_freq;
    uint32_t frame_sync_ind;
    uint64_t epochs, n_frame_bits;
    uint64_t start_index;              // AcquisitionResult.sample_global_index of the hand-over: the channel's epoch 0 starts there
} gmrx_channel;

typedef struct {
    double wall_seconds, signal_seconds;
    double seconds_frontend, seconds_acquisition, seconds_fine_doppler, seconds_tracking, seconds_nav_bits;
    double fe_block_first_s, fe_block_median_s, fe_block_max_after_first_s, feeder_held_back_s;
    uint32_t dwells, channels_started, blocks, fe_runs_repaired;     // fe_runs_repaired: runs of the speculative front-end done again (gm_frontend_debug_repairs)
    uint64_t channel_epochs, tracking_passes;
    double first_handover_signal_ms, first_handover_wall_s;
    double first_bit_sync_signal_ms, first_bit_sync_wall_s;
    double first_frame_sync_signal_ms, first_frame_sync_wall_s;
    gmrx_channel channels[32];
} gmrx_report;

static thread_local std::string g_err;
const char* gmrx_last_error(void) { return g_err.c_str(); }
int gmrx_abi_version(void) { return 1; }

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int gmrx_receiver_run(const gmrx_cfg* cfg, const int8_t* iq, size_t n_samples, gmrx_report* rep) {
    if (!cfg || !iq || !rep || !n_samples) { g_err = "null argument"; return GM_ERR_INVALID_ARG; }
    if (cfg->n_channels == 0 || cfg->n_channels > 32) { g_err = "n_channels must be 1..32"; return GM_ERR_INVALID_ARG; }
    std::memset(rep, 0, sizeof(*rep));
    rep->first_handover_signal_ms = rep->first_bit_sync_signal_ms = rep->first_frame_sync_signal_ms = -1.0;
    try {
        init(cfg->device);
        const float fs = cfg->fs;
        const uint32_t C = cfg->n_channels;
        const size_t BLK = cfg->block_samples;
        MulticastRingBuffer ring(size_t(1) << cfg->ring_log2);
        DigitalFrontend fe(cfg->f_if, fs, fs);
        Channel<AcquisitionResult> acq_to_trk;
        Channel<TrackingMessage> trk_to_acq;
        StageControl ctl;
        std::vector<std::unique_ptr<NavSyncStatus>> navs;
        for (uint32_t c = 0; c < C; ++c) navs.push_back(std::make_unique<NavSyncStatus>(cfg->nav_mode));
        std::vector<float> nav_old(C, 0.0f);
        std::vector<uint64_t> nav_cnt(C, 0);
        std::vector<gm_nav_status> nav_st(C);
        std::vector<float> ip;
        std::vector<gmrx_channel> chan(C);
        std::atomic<double> t_start{0.0};
        std::mutex ev_mu;
        uint64_t start_of_prn[33] = {0};       // sample_global_index + 1 of the first hand-over of each PRN

        AcquisitionRunOptions aopt;
        aopt.freq_search_hz = cfg->freq_search_hz; aopt.freq_step_hz = cfg->freq_step_hz;
        aopt.long_samples_length = cfg->n_integrations; aopt.decision_mode = cfg->decision_mode;
        aopt.pace_on_signal_time = true; aopt.first_round_signal_ms = cfg->first_round_signal_ms;
        aopt.fine_doppler = cfg->fine_doppler != 0;
        aopt.on_round = [&](uint64_t head, const std::vector<std::optional<AcquisitionResult>>& res) {
            std::lock_guard<std::mutex> g(ev_mu);
            rep->dwells++;
            bool any = false;
            for (const auto& r : res) { any = any || bool(r); if (r && r->prn <= 32 && !start_of_prn[r->prn]) start_of_prn[r->prn] = r->sample_global_index + 1; }
            if (any && rep->first_handover_signal_ms < 0) {
                rep->first_handover_signal_ms = double(head) / fs * 1e3;
                rep->first_handover_wall_s = now_s() - t_start.load();
            }
        };
        TrackingRunOptions topt;
        topt.code_index_mode = cfg->code_index_mode; topt.n_channels = C; topt.async_tickets = cfg->async_tickets != 0;
        std::vector<gm_trk_state> finals;
        topt.final_states = &finals;
        // nav_decoding's per-epoch step on the prompt I (decoding.rs:102-145): one call per channel and collected block
        topt.on_epochs = [&](const EpochBlock& b) {
            for (uint32_t c = 0; c < b.n_channels; ++c) {
                ip.clear();
                for (uint32_t e = 0; e < b.passes; ++e)
                    if (b.processed[size_t(e) * b.n_channels + c]) ip.push_back(b.outs[size_t(e) * b.n_channels + c].ip);
                if (ip.empty()) continue;
                int64_t fb = -1, ff = -1;
                check(gm_nav_sync_update_many(navs[c]->handle(), nav_old[c], ip.data(), 1, ip.size(), nav_cnt[c], 0, &nav_st[c], &fb, &ff),
                      "nav_decoding");
                nav_old[c] = ip.back(); nav_cnt[c] += ip.size();
                chan[c].epochs += ip.size();
                if (fb >= 0 || ff >= 0) {
                    std::lock_guard<std::mutex> g(ev_mu);
                    const double sig_ms = double(b.head) / fs * 1e3, wall = now_s() - t_start.load();
                    if (fb >= 0 && rep->first_bit_sync_signal_ms < 0) { rep->first_bit_sync_signal_ms = sig_ms; rep->first_bit_sync_wall_s = wall; }
                    if (ff >= 0 && rep->first_frame_sync_signal_ms < 0) { rep->first_frame_sync_signal_ms = sig_ms; rep->first_frame_sync_wall_s = wall; }
                }
                chan[c].prn = b.channel_prn[c];
            }
        };

        // ---- before the clock starts: code objects, pinned staging, result slots (start-up, not throughput)
        if (cfg->warmup_calls) {
            MulticastRingBuffer wring(size_t(1) << 18);
            DigitalFrontend wfe(cfg->f_if, fs, fs);
            const uint32_t N = uint32_t(std::lround(fs / 1000.0f));
            const size_t wn = std::min<size_t>(n_samples, size_t(cfg->n_integrations + 2) * N);
            wfe.write_ring_i8(wring, iq, wn); wring.flush();
            {   // one dwell + fine Doppler on a scratch engine of the same geometry
                const size_t capacity = size_t(uint16_t(cfg->freq_search_hz) / uint16_t(cfg->freq_step_hz)) + 1;
                std::vector<float> doppler(capacity);
                for (size_t i = 0; i < capacity; ++i) doppler[i] = -cfg->freq_search_hz / 2.0f + float(i) * cfg->freq_step_hz;
                std::vector<uint8_t> prns(32);
                for (uint8_t p = 0; p < 32; ++p) prns[p] = uint8_t(p + 1);
                AcquisitionEngine weng(fs, 0.0f, N, doppler, prns, cfg->n_integrations, 7.0f, cfg->decision_mode);
                auto r = weng.search_ring(wring.handle());
                if (r && cfg->fine_doppler) { bool any = false; for (auto& x : *r) any = any || bool(x); if (any) (void)weng.finer_doppler(*r); }
            }
            TrackingManager wm(fs, C, cfg->code_index_mode, 3, false, false, true);
            AcquisitionResult r0{}; r0.prn = 1; r0.fs = fs; r0.mag_relative = 1.0f;
            wm.channels[0].start(r0);
            for (uint32_t i = 0; i < cfg->warmup_calls; ++i) { uint64_t t = wm.process_channels_async(wring, 4); wm.collect(t, true); }
        }
        size_t off = std::min<size_t>(cfg->pre_samples, n_samples);
        if (off) { fe.write_ring_i8(ring, iq, off); ring.flush(); }

        std::atomic<bool> stage_failed{false};
        std::thread t_acq([&] { try { run_acquisition(ring, fs, 0.0f, acq_to_trk, trk_to_acq, ctl, aopt); }
                                catch (const std::exception& e) { std::fprintf(stderr, "acquisition stage: %s\n", e.what()); stage_failed = true; ctl.stop = true; } });
        std::thread t_trk([&] { try { run_tracking(ring, acq_to_trk, trk_to_acq, fs, ctl, topt); }
                                catch (const std::exception& e) { std::fprintf(stderr, "tracking stage: %s\n", e.what()); stage_failed = true; ctl.stop = true; }
                                ctl.trk_finished = true; });

        // ---- the feeder: rf_thread's block step, never waiting for the front-end (copy + kernel enqueued on the ring's streams);
        // held back only when it is more than three quarters of a ring ahead of what tracking has collected (a replay outruns real time)
        for (int i = 0; i < 20000 && ctl.stages_ready.load() < 2 && !ctl.stop.load(); ++i) std::this_thread::sleep_for(std::chrono::microseconds(500));
        std::vector<double> fe_blocks;
        const uint64_t half_ring = (uint64_t(1) << cfg->ring_log2) / 4 * 3;       // (three quarters: the writer may lead what tracking has collected by that much)
        t_start = now_s();
        const double t0 = t_start.load();
        double held = 0.0;
        const bool trace = std::getenv("GMRX_TRACE") != nullptr;      // diagnostics of the HARNESS (the product library reads no environment)
        for (; off < n_samples && !ctl.stop.load(); off += BLK) {
            const double tb = now_s();
            if (trace) std::fprintf(stderr, "blk %4zu t %8.3f ms  feeder %7.1f  published %7.1f  collected %7.1f ms  epochs %llu dwells %u\n", off / BLK, (tb - t0) * 1e3,
                                    off / double(fs) * 1e3, ring.get_head() / double(fs) * 1e3, ctl.trk_collected_head.load() / double(fs) * 1e3,
                                    (unsigned long long)ctl.channel_epochs.load(), rep->dwells);
            while (!ctl.stop.load() && uint64_t(off) > ctl.trk_collected_head.load() + half_ring) std::this_thread::sleep_for(std::chrono::microseconds(50));
            const double tw = now_s();
            held += tw - tb;
            fe.write_ring_i8(ring, iq + 2 * off, std::min(BLK, n_samples - off));
            fe_blocks.push_back(now_s() - tw);
        }
        const double tf = now_s();
        ring.flush();
        if (trace) std::fprintf(stderr, "end: feeder done %8.3f ms, flushed %8.3f ms\n", (tf - t0) * 1e3, (now_s() - t0) * 1e3);
        rep->seconds_frontend = (now_s() - tf);
        for (double b : fe_blocks) rep->seconds_frontend += b;
        // everything the ring holds is tracked: the tracking stage drains on `stop`; acquisition just ends
        // (give a round that is due at the final head the chance to run first: signal-time pacing has no wall clock to wait for)
        for (int i = 0; i < 2000 && ctl.trk_collected_head.load() < ring.get_head() && !ctl.stop.load(); ++i)
            std::this_thread::sleep_for(std::chrono::microseconds(100));
        if (trace) std::fprintf(stderr, "end: tracking collected the final head %8.3f ms\n", (now_s() - t0) * 1e3);
        ctl.stop = true;
        while (!ctl.trk_finished.load()) std::this_thread::yield();       // the last call collected: the chain is through (tearing the
        rep->wall_seconds = now_s() - t0;                                 // stage's handles down — pinned slots, streams — is not throughput)
        t_trk.join();
        if (trace) std::fprintf(stderr, "end: tracking stage finished %8.3f ms, joined %8.3f ms\n", rep->wall_seconds * 1e3, (now_s() - t0) * 1e3);
        t_acq.join();

        if (stage_failed.load()) { g_err = "a stage driver threw (see stderr)"; return GM_ERR_HIP; }
        rep->signal_seconds = double(n_samples) / fs;
        rep->seconds_acquisition = double(ctl.acq_ns.load()) * 1e-9;
        rep->seconds_fine_doppler = double(ctl.fine_ns.load()) * 1e-9;
        rep->seconds_tracking = double(ctl.trk_ns.load()) * 1e-9;
        rep->seconds_nav_bits = double(ctl.hook_ns.load()) * 1e-9;
        rep->feeder_held_back_s = held;
        rep->blocks = uint32_t(fe_blocks.size());
        if (!fe_blocks.empty()) {
            rep->fe_block_first_s = fe_blocks[0];
            std::vector<double> s(fe_blocks.begin() + (fe_blocks.size() > 1 ? 1 : 0), fe_blocks.end());
            std::sort(s.begin(), s.end());
            rep->fe_block_median_s = s[s.size() / 2];
            rep->fe_block_max_after_first_s = s.back();
        }
        rep->fe_runs_repaired = fe.debug_repairs();
        rep->channel_epochs = ctl.channel_epochs.load();
        rep->tracking_passes = ctl.trk_passes.load();
        for (uint32_t c = 0; c < C && c < finals.size(); ++c) {
            gmrx_channel& o = rep->channels[c];
            o = chan[c];
            o.active = finals[c].active; o.lost_counter = finals[c].lost_counter; o.carrier_freq = finals[c].carrier_freq;
            if (o.prn) rep->channels_started++;
            if (o.prn && o.prn <= 32 && start_of_prn[o.prn]) o.start_index = start_of_prn[o.prn] - 1;
            if (nav_cnt[c]) {
                o.bit_sync = nav_st[c].flag_bit_sync; o.frame_sync = nav_st[c].flag_frame_sync;
                o.frame_sync_ind = nav_st[c].frame_sync_ind; o.n_frame_bits = nav_st[c].n_frame_bits;
            }
        }
        return GM_OK;
    } catch (const Panic& p) {
        g_err = p.what();
        return p.status;
    } catch (const std::exception& e) {
        g_err = e.what();
        return GM_ERR_HIP;
    }
}

// The tracking stage driver alone, deterministically: `samples` (c32, n <= ring size) are written to a fresh ring and published,
// `results` are queued on the acquisition channel, and gnss::run_tracking is called with `stop` already set — its drain hands the
// acquisitions to channels and runs every whole code period the ring holds.  Twice: the synchronous loop -> states_sync, the
// ticket loop -> states_async ([n_channels] each), with the messages each sent back (Locked / Lost counts) and the channel epochs.
int gmrx_tracking_ab(int32_t device, float fs, uint32_t n_channels, int32_t code_index_mode, uint32_t ring_log2, const gm_c32* samples, size_t n,
                     uint32_t write_block, const gm_acq_result* results, uint32_t n_results, gm_trk_state* states_sync,
                     gm_trk_state* states_async, uint64_t epochs[2], uint32_t locked[2], uint32_t lost[2], double seconds[2]) {
    if (!samples || !results || !states_sync || !states_async || !n_channels) { g_err = "null argument"; return GM_ERR_INVALID_ARG; }
    try {
        init(device);
        for (int mode = 0; mode < 2; ++mode) {
            MulticastRingBuffer ring(size_t(1) << ring_log2);
            const size_t wb = write_block ? write_block : n;
            for (size_t off = 0; off < n; off += wb)
                ring.write_samples_async(reinterpret_cast<const Complex32*>(samples) + off, std::min(wb, n - off));
            if (mode == 0) ring.flush();          // the synchronous loop reads the PUBLISHED head; the ticket loop orders itself on the device
            Channel<AcquisitionResult> acq_to_trk;
            Channel<TrackingMessage> trk_to_acq;
            for (uint32_t i = 0; i < n_results; ++i) acq_to_trk.send(results[i]);
            StageControl ctl;
            ctl.stop = true;
            TrackingRunOptions opt;
            opt.code_index_mode = code_index_mode; opt.n_channels = n_channels; opt.async_tickets = mode == 1;
            std::vector<gm_trk_state> finals;
            opt.final_states = &finals;
            const double t0 = now_s();
            run_tracking(ring, acq_to_trk, trk_to_acq, fs, ctl, opt);
            if (seconds) seconds[mode] = now_s() - t0;
            ring.flush();
            std::memcpy(mode ? states_async : states_sync, finals.data(), size_t(n_channels) * sizeof(gm_trk_state));
            if (epochs) epochs[mode] = ctl.channel_epochs.load();
            uint32_t nl = 0, nx = 0;
            while (auto m = trk_to_acq.try_recv()) (m->kind == TrackingMessageKind::SatelliteLocked ? nl : nx)++;
            if (locked) locked[mode] = nl;
            if (lost) lost[mode] = nx;
        }
        return GM_OK;
    } catch (const Panic& p) {
        g_err = p.what();
        return p.status;
    } catch (const std::exception& e) {
        g_err = e.what();
        return GM_ERR_HIP;
    }
}

}  // extern "C"
