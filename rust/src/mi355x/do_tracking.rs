// do_tracking.rs — DESTINATION: src/mi355x/do_tracking.rs (module crate::mi355x::do_tracking, a new sibling of
// crate::tracking::do_tracking, which stays in the crate untouched).  The items of src/tracking/do_tracking.rs that sit on the
// hot path, on the MI355X library.  Imported from the reference's module because they do not change: LoopFilter (:52-71),
// TrackingMessage (:47-50), TrackingError (:31-45).  The reference's private constants (:16-29) are restated below.
//
//   * TrackingChannel keeps its 22 pub fields and every method signature (:88-327).  Each channel owns a ONE-channel
//     handle; the evolving fields are written to the handle before a call and read back after it, so code that pokes the
//     fields between calls (the reference's tests do) sees the same behaviour.  `update` sizes `data_samples` before it
//     slices it (as committed the reference slices an empty Vec there and panics, :174-177).
//   * TrackingManager::process_channels (:351-371) is the fast path: all channels in ONE handle on a device mirror of the
//     ring, `update` for every active channel and up to LOOP_MS code periods in one persistent launch.  NOTE the one
//     behavioural difference: the reference runs ONE `update` per active channel per call (:364-371); this runs up to
//     LOOP_MS = 10 per call (as many as the ring holds samples for).  Under run()'s loop (:391-414) the two are the same
//     sequence of updates; a caller that interleaves its own work between calls sees channels advance by up to 10 code
//     periods per call.  `gm_trk_update_all(.., 1, ..)` is the one-epoch form (PROCESS_EPOCHS below).
//   * Fields poked between calls are honoured on the batched path too: a channel whose pub fields differ from what the handle
//     was last seen to hold is written before the launch — a dirty test, ONE gm_trk_set_states call for all of them, none when
//     nothing was touched — and all records come back in ONE gm_trk_get_states call (ABI 7; rounds 3-5 made 30 calls per pass).
//     The host ring reaches the device mirror straight out of its own storage (gm_ring_write_samples_async on <= 2 pieces):
//     no Vec per call.
//   * run() drives the ticket path (process_channels_async / collect_ready): passes ENQUEUED behind the mirror's copies, results
//     adopted a wake-up or two later; the thread never waits for a pass (the receiver gain of round 5 inside the deliverable).
//   * run — the tracking stage's thread body with the reference's signature and control flow (:384-415): main.rs:216-227 calls
//     it through `use gnss_sdr_rs::mi355x::do_tracking;` instead of `...::tracking::do_tracking;`.
use crate::acquisition::do_acquisition::{AcquisitionResult, ChannelState};
use crate::mi355x::*;
use crate::tracking::do_tracking::{LoopFilter, TrackingError, TrackingMessage};
use crate::utilities::multicast_ring_buffer::MulticastRingBuffer;
use crossbeam_channel::{Receiver, Sender};   // the crate's channels (do_tracking.rs:8, main.rs:183-184)
use num_complex::Complex32;
use std::collections::VecDeque;
use std::sync::Arc;

const NUM_OF_CHANNELS: usize = 15;                  // :18
const PLL_SUM_CARR: f32 = 0.001;                    // :26
const DLL_SUM_CODE: f32 = 0.001;                    // :27
const LOOP_MS: usize = 10;                          // :29
const PROCESS_EPOCHS: usize = LOOP_MS;              // code periods per process_channels call; 1 = the reference's call granularity
const CODE_INDEX_FAITHFUL: i32 = 0;                 // GPS_CA_CODE_32_PRN[prn] and the saturating late-arm index, as written (:275-276)

fn trk_cfg(fs: f32, n_channels: u32) -> GmTrkCfg {  // zero = the reference's constants (:16-28)
    GmTrkCfg { fs, n_channels, n_arms: 3, early_late_space: 0.5, very_early_late_space: 0.0, code_index_mode: CODE_INDEX_FAITHFUL,
               boc11: 0, codes: std::ptr::null(), n_codes: 0, code_len: 0, nominal_code_rate: 0.0, pll_bw: 0.0, pll_zeta: 0.0,
               pll_gain: 0.0, dll_bw: 0.0, dll_zeta: 0.0, dll_gain: 0.0, pll_dt: 0.0, dll_dt: 0.0, lock_threshold: 0.0,
               max_lost_epochs: 0, strict_libm: 0, strict_sum_order: 0, share_device: 1 }
}
pub struct TrackingChannel {
    pub id: u8,
    pub prn: u8,
    pub state: ChannelState,
    pub lost_counter: u32,
    pub fs: f32,
    pub next_sample_index: usize,
    pub num_samples_per_code: usize,
    pub ca_code_samples: Vec<i8>,
    pub data_samples: Vec<Complex32>,

    pub carrier_freq: f32,
    pub carrier_phase: f32,
    pub carrier_error: f32,
    pub carrier_nco: f32,
    pub code_phase: f32,
    pub code_error: f32,
    pub code_nco: f32,
    pub code_rate: f32,
    pub cos_p: Vec<f32>,
    pub sin_p: Vec<f32>,

    pub i_prompt: f32,
    pub q_prompt: f32,

    pub pll_filter: LoopFilter,
    pub dll_filter: LoopFilter,

    h: *mut GmTrk,
}
unsafe impl Send for TrackingChannel {}

impl TrackingChannel {
    pub fn new(id: u8, fs: f32) -> Self {
        let mut h = std::ptr::null_mut();
        let st = unsafe { gm_trk_create(&trk_cfg(fs, 1), &mut h) };
        assert_eq!(st, 0, "gm_trk_create: {}", last_error());
        let mut s = GmTrkState::default();
        unsafe { gm_trk_get_state(h, 0, &mut s); }       // TrackingChannel::new's initial values (:118-146)
        let (mut p1, mut p2, mut d1, mut d2) = (0.0f32, 0.0f32, 0.0f32, 0.0f32);
        unsafe { gm_loop_filter_new(25.0, 0.7, 0.25, &mut p1, &mut p2); gm_loop_filter_new(2.0, 0.7, 1.0, &mut d1, &mut d2); }
        Self { id, prn: 0, state: ChannelState::Idle, lost_counter: 0, fs, next_sample_index: 0,
               num_samples_per_code: s.num_samples_per_code as usize, ca_code_samples: Vec::new(), data_samples: Vec::new(),
               carrier_freq: 0.0, carrier_phase: 0.0, carrier_error: 0.0, carrier_nco: 0.0, code_phase: 0.0, code_error: 0.0,
               code_nco: 0.0, code_rate: s.code_rate, cos_p: Vec::new(), sin_p: Vec::new(), i_prompt: 0.0, q_prompt: 0.0,
               pll_filter: LoopFilter { tau1: p1, tau2: p2 }, dll_filter: LoopFilter { tau1: d1, tau2: d2 }, h }
    }

    fn raw_state(&self) -> GmTrkState {      // the pub fields as the library's state record
        GmTrkState { prn: self.prn, active: self.is_active() as u8, reserved: [0; 2], lost_counter: self.lost_counter,
            next_sample_index: self.next_sample_index as u64, num_samples_per_code: self.num_samples_per_code as u64,
            carrier_freq: self.carrier_freq, carrier_phase: self.carrier_phase, carrier_error: self.carrier_error,
            carrier_nco: self.carrier_nco, code_phase: self.code_phase, code_error: self.code_error, code_nco: self.code_nco,
            code_rate: self.code_rate, i_prompt: self.i_prompt, q_prompt: self.q_prompt }
    }
    fn push(&self) {       // fields -> handle
        let s = self.raw_state();
        let st = unsafe { gm_trk_set_state(self.h, 0, &s) };
        assert_eq!(st, 0, "gm_trk_set_state: {}", last_error());
    }
    fn pull(&mut self) {   // handle -> fields
        let mut s = GmTrkState::default();
        let st = unsafe { gm_trk_get_state(self.h, 0, &mut s) };
        assert_eq!(st, 0, "gm_trk_get_state: {}", last_error());
        self.prn = s.prn; self.lost_counter = s.lost_counter;
        if s.active == 0 { self.state = ChannelState::Idle; }
        self.next_sample_index = s.next_sample_index as usize; self.num_samples_per_code = s.num_samples_per_code as usize;
        self.carrier_freq = s.carrier_freq; self.carrier_phase = s.carrier_phase; self.carrier_error = s.carrier_error;
        self.carrier_nco = s.carrier_nco; self.code_phase = s.code_phase; self.code_error = s.code_error;
        self.code_nco = s.code_nco; self.code_rate = s.code_rate; self.i_prompt = s.i_prompt; self.q_prompt = s.q_prompt;
    }

    pub fn start(&mut self, result: AcquisitionResult) {
        self.prn = result.prn;
        self.carrier_freq = result.carrier_freq;
        self.code_phase = result.code_phase_chips;
        self.next_sample_index = result.sample_global_index;
        self.state = ChannelState::Tracking(result.prn);
    }

    pub fn is_active(&self) -> bool {
        self.state == ChannelState::Tracking(self.prn)
    }

    pub fn update(&mut self, buff: Arc<MulticastRingBuffer>) -> Option<TrackingMessage> {
        if self.state != ChannelState::Tracking(self.prn) {
            return None;
        }
        // generate_ca_code_samples(..).len() (:165-166): only the length is used; the library's closed form of it
        self.num_samples_per_code = (self.fs / (self.code_rate / 1023.0)).round() as usize;
        let head = buff.get_head();
        if (head.wrapping_sub(self.next_sample_index + self.num_samples_per_code) as isize) < 0 {
            return None;
        }
        self.data_samples.resize(self.num_samples_per_code, Complex32::new(0.0, 0.0));     // sized, then sliced (:174-177)
        buff.copy_to_slice(self.next_sample_index, &mut self.data_samples[0..self.num_samples_per_code]);
        self.do_work()
    }

    fn do_work(&mut self) -> Option<TrackingMessage> {      // :183-210 on the device
        self.push();
        let (mut out, mut lost, mut lost_prn) = (GmTrkOut::default(), 0u8, 0u8);
        let st = unsafe { gm_trk_do_work(self.h, 0, self.data_samples.as_ptr(), self.num_samples_per_code, &mut out,
                                         &mut lost, &mut lost_prn) };
        assert_eq!(st, 0, "gm_trk_do_work: {}", last_error());          // row GPS_CA_CODE_32_PRN[32]: the reference panics (:276)
        self.pull();
        self.data_samples.clear();                                       // free_data (:304-309)
        if lost != 0 { Some(TrackingMessage::SatelliteLost(lost_prn)) } else { None }   // prn 0: built after reset() (:199-201)
    }

    pub fn early_late_correlation(&mut self) -> (f32, f32, f32, f32, f32, f32) {
        self.push();
        let mut out = GmTrkOut::default();
        let st = unsafe { gm_trk_correlate(self.h, 0, self.data_samples.as_ptr(), self.num_samples_per_code, &mut out) };
        assert_eq!(st, 0, "gm_trk_correlate: {}", last_error());
        self.pull();                                                     // carrier_phase, code_phase, i/q_prompt advanced (:240-270)
        (out.ip, out.qp, out.ie, out.qe, out.il, out.ql)
    }

    pub fn get_ca_chip(&self, phase: f32) -> f32 {
        self.push();
        let mut chip = 0.0f32;
        let st = unsafe { gm_trk_get_ca_chip(self.h, 0, phase, &mut chip) };
        assert_eq!(st, 0, "index out of bounds: GPS_CA_CODE_32_PRN[{}]", self.prn);     // the reference's panic (:276)
        chip
    }

    pub fn run_loop_filters(&mut self, i_p: f32, q_p: f32, i_e: f32, q_e: f32, i_l: f32, q_l: f32) {
        // scalar host arithmetic in the reference too (:279-302); LoopFilter::update through the library's twin
        let pll_err = (q_p / i_p).atan() / (2.0 * std::f32::consts::PI);
        self.carrier_nco = unsafe { gm_loop_filter_update(self.pll_filter.tau1, self.pll_filter.tau2, pll_err, self.carrier_error, PLL_SUM_CARR) };
        self.carrier_error = pll_err;
        self.carrier_freq += self.carrier_nco;
        let pow_e = (i_e.powi(2) + q_e.powi(2)).sqrt();
        let pow_l = (i_l.powi(2) + q_l.powi(2)).sqrt();
        let dll_err = if (pow_e + pow_l) != 0.0 { (pow_e - pow_l) / (pow_e + pow_l) } else { 0.0 };
        self.code_nco = unsafe { gm_loop_filter_update(self.dll_filter.tau1, self.dll_filter.tau2, dll_err, self.code_error, DLL_SUM_CODE) };
        self.code_error = dll_err;
        self.code_rate += self.code_nco;
    }

    pub fn reset(&mut self) {
        let st = unsafe { gm_trk_reset(self.h, 0) };
        assert_eq!(st, 0, "gm_trk_reset: {}", last_error());
        self.state = ChannelState::Idle;
        self.pull();                                                     // every field zeroed, code_rate = 0.0 like :311-327
    }
}
impl Drop for TrackingChannel { fn drop(&mut self) { unsafe { gm_trk_destroy(self.h); } } }

pub struct TrackingManager {
    pub channels: Vec<TrackingChannel>,
    pub acq_to_trk: Receiver<AcquisitionResult>,
    pub trk_to_acq: Sender<TrackingMessage>,
    h: *mut GmTrk,                 // all NUM_OF_CHANNELS channels in one handle (the batched path)
    ring: *mut GmRing,             // device mirror of the MulticastRingBuffer, fed from it below
    mirrored: usize,               // absolute index up to which the mirror has been handed the host ring's samples
    synced: Vec<GmTrkState>,       // what the batched handle holds for every channel, as last read back: a channel whose pub
                                   // fields still equal this is NOT written again (the dirty test that replaces 15 set_state calls)
    scratch: Vec<GmTrkState>,      // reused: records read back / written
    which: Vec<u8>,                // reused: per-channel dirty flags
    lost: Vec<u8>,                 // reused: [passes][channels]
    tickets: VecDeque<(u64, u32)>, // calls in flight on the ticket path: (ticket, passes), oldest first
    covered: Vec<usize>,           // ring head up to which passes for each channel have been enqueued (ticket path)
    slots_sized: bool,             // the first asynchronous call has sized the library's result slots for MAX_PASSES_PER_CALL
}
unsafe impl Send for TrackingManager {}

const MAX_IN_FLIGHT: usize = 6;                     // the library holds 8 result slots
const MAX_PASSES_PER_CALL: usize = 256;             // the result slots are sized for this by the first call

impl TrackingManager {
    pub fn new(
        acq_to_trk: Receiver<AcquisitionResult>,
        trk_to_acq: Sender<TrackingMessage>,
        fs: f32,
    ) -> Self {
        let mut h = std::ptr::null_mut();
        let st = unsafe { gm_trk_create(&trk_cfg(fs, NUM_OF_CHANNELS as u32), &mut h) };
        assert_eq!(st, 0, "gm_trk_create: {}", last_error());
        let mut synced = vec![GmTrkState::default(); NUM_OF_CHANNELS];
        let st = unsafe { gm_trk_get_states(h, synced.as_mut_ptr()) };
        assert_eq!(st, 0, "gm_trk_get_states: {}", last_error());
        Self { channels: (0..NUM_OF_CHANNELS).map(|id| TrackingChannel::new(id as u8, fs)).collect(),
               acq_to_trk, trk_to_acq, h, ring: std::ptr::null_mut(), mirrored: 0, synced,
               scratch: vec![GmTrkState::default(); NUM_OF_CHANNELS], which: vec![0u8; NUM_OF_CHANNELS],
               lost: vec![0u8; MAX_PASSES_PER_CALL * NUM_OF_CHANNELS], tickets: VecDeque::new(), covered: vec![0usize; NUM_OF_CHANNELS],
               slots_sized: false }
    }

    // process_channels' first half, unchanged (:352-362): hand new acquisitions to idle channels
    fn take_acquisitions(&mut self) {
        while let Ok(msg) = self.acq_to_trk.try_recv() {
            if let Some((i, channel)) = self.channels.iter_mut().enumerate().find(|(_, c)| c.state == ChannelState::Idle) {
                let _ = self.trk_to_acq.send(TrackingMessage::SatelliteLocked(msg.prn));
                channel.start(msg);            // the pub fields now differ from `synced`: written to the handle by push_dirty
                self.covered[i] = channel.next_sample_index;
            }
        }
    }

    // bring the device mirror up to the host ring's head WITHOUT a staging copy of our own and without waiting for the transfer:
    // the library stages straight out of the host ring's storage (<= 2 pieces at the wrap) and copies on the mirror's own stream
    // (rf_thread writes the host ring; a front-end that writes the mirror directly — gm_frontend_write_ring — makes this unnecessary)
    fn mirror_ring(&mut self, multi_ring_buf: &MulticastRingBuffer) -> usize {
        let head = multi_ring_buf.get_head();
        let size = multi_ring_buf.buffer.len();
        if self.ring.is_null() {
            let st = unsafe { gm_ring_create(size, &mut self.ring) };                          // same power-of-two size (:46-61)
            assert_eq!(st, 0, "gm_ring_create: {}", last_error());
        }
        let base = multi_ring_buf.buffer.as_ptr() as *const Complex32;                         // as write_samples / copy_to_slice do (:73, :112)
        while head > self.mirrored {
            let start = self.mirrored & (size - 1);
            let n = (head - self.mirrored).min(size - start);                                  // up to the wrap; the rest next turn
            let st = unsafe { gm_ring_write_samples_async(self.ring, base.add(start), n) };
            assert_eq!(st, 0, "gm_ring_write_samples_async: {}", last_error());
            self.mirrored += n;
        }
        head
    }

    // the pub fields are the truth between calls (the reference's tests poke them): a channel whose fields differ from what the
    // handle was last seen to hold is written — ONE call for all of them; after a call they are equal and nothing is written
    fn push_dirty(&mut self) {
        let mut any = false;
        for (i, ch) in self.channels.iter().enumerate() {
            let s = ch.raw_state();
            let dirty = s != self.synced[i];
            self.which[i] = dirty as u8;
            if dirty { self.scratch[i] = s; self.synced[i] = s; any = true; }
        }
        if any {
            let st = unsafe { gm_trk_set_states(self.h, self.scratch.as_ptr(), self.which.as_ptr()) };
            assert_eq!(st, 0, "gm_trk_set_states: {}", last_error());
        }
    }

    // records read back from the handle -> pub fields (+ `synced`), SatelliteLost for every loss flag
    fn adopt(&mut self, passes: usize) {
        for (i, ch) in self.channels.iter_mut().enumerate() {
            let s = self.scratch[i];      // (never older than a hand-over: process_channels_async collects everything before it starts a channel)
            self.synced[i] = s;
            ch.prn = s.prn; ch.lost_counter = s.lost_counter; ch.next_sample_index = s.next_sample_index as usize;
            ch.num_samples_per_code = s.num_samples_per_code as usize; ch.carrier_freq = s.carrier_freq;
            ch.carrier_phase = s.carrier_phase; ch.carrier_error = s.carrier_error; ch.carrier_nco = s.carrier_nco;
            ch.code_phase = s.code_phase; ch.code_error = s.code_error; ch.code_nco = s.code_nco; ch.code_rate = s.code_rate;
            ch.i_prompt = s.i_prompt; ch.q_prompt = s.q_prompt;
            if s.active == 0 { ch.state = ChannelState::Idle; }
        }
        for l in self.lost[..passes * self.channels.len()].iter() {          // SatelliteLost carries prn 0 (:199-201)
            if *l != 0 { let _ = self.trk_to_acq.send(TrackingMessage::SatelliteLost(0)); }
        }
    }

    // collect the calls of the ticket path that have finished (all of them, waiting, when `all`): results -> fields + messages.
    // A failed collect has consumed its ticket (include/gnss_mi355x.h): it is dropped here before the panic.
    fn collect_ready(&mut self, all: bool) {
        while let Some(&(ticket, passes)) = self.tickets.front() {
            let wait = all || self.tickets.len() >= MAX_IN_FLIGHT;
            let (mut done, mut ready) = (0u32, 0i32);
            let st = unsafe { gm_trk_collect(self.h, ticket, wait as i32, std::ptr::null_mut(), std::ptr::null_mut(), self.lost.as_mut_ptr(),
                                             self.scratch.as_mut_ptr(), &mut done, &mut ready) };
            if st != 0 { self.tickets.pop_front(); }
            assert_eq!(st, 0, "gm_trk_collect: {}", last_error());
            if ready == 0 { break; }
            self.tickets.pop_front();
            self.adopt(passes as usize);
        }
    }

    // where channel i will stand once the calls in flight have run: its record advanced by the whole periods up to covered[i]
    fn planned_index(&self, i: usize) -> usize {
        let ch = &self.channels[i];
        let (mut idx, n) = (ch.next_sample_index, ch.num_samples_per_code.max(1));
        if self.covered[i] > idx { idx += ((self.covered[i] - idx) / n) * n; }
        idx
    }

    pub fn process_channels(&mut self, multi_ring_buf: Arc<MulticastRingBuffer>) {
        self.collect_ready(true);                                        // (nothing in flight unless run_async was used before)
        self.take_acquisitions();                                        // unchanged (:352-362)
        self.mirror_ring(&multi_ring_buf);
        let st = unsafe { gm_ring_flush(self.ring) };                    // this form reads the PUBLISHED head: the samples have landed
        assert_eq!(st, 0, "gm_ring_flush: {}", last_error());
        self.push_dirty();
        // channels.par_iter_mut().filter(is_active).for_each(update) (:364-371), up to PROCESS_EPOCHS code periods per call
        let mut done = 0u32;
        let st = unsafe { gm_trk_update_all(self.h, self.ring, PROCESS_EPOCHS as u32, std::ptr::null_mut(), std::ptr::null_mut(),
                                            self.lost.as_mut_ptr(), &mut done) };
        assert_eq!(st, 0, "gm_trk_update_all: {}", last_error());
        let st = unsafe { gm_trk_get_states(self.h, self.scratch.as_mut_ptr()) };       // ONE call: the pub fields follow the device state
        assert_eq!(st, 0, "gm_trk_get_states: {}", last_error());
        self.adopt(PROCESS_EPOCHS);
    }

    /// process_channels WITHOUT a host wait (the ticket path run() uses): new acquisitions, the mirror's copies, dirty fields and
    /// the passes up to the host ring's head are ENQUEUED — the passes ordered on the device behind the copies
    /// (gm_trk_update_all_async: the Condvar wait of :392-406 as an event on the mirror's stream) — and results of earlier calls
    /// are adopted when they are there.  The pub fields lag by the calls in flight; next_tracking_index() plans with them.
    pub fn process_channels_async(&mut self, multi_ring_buf: Arc<MulticastRingBuffer>) {
        let had_new = !self.acq_to_trk.is_empty();
        if had_new { self.collect_ready(true); }                         // a hand-over writes a record: nothing may be in flight over it
        self.take_acquisitions();
        let head = self.mirror_ring(&multi_ring_buf);
        self.push_dirty();                                               // (gm_trk_set_states waits for the handle's stream: only ever after a hand-over or a poked field)
        let mut passes = 0usize;
        for i in 0..self.channels.len() {
            if !self.channels[i].is_active() { continue; }
            let (idx, n) = (self.planned_index(i), self.channels[i].num_samples_per_code.max(1));
            if head > idx { passes = passes.max((head - idx) / n); }
        }
        if passes >= 1 && self.tickets.len() < MAX_IN_FLIGHT {
            // + 2: the period length moves by a sample now and then.  The very first call is made with the largest pass count the
            // loop will ever ask for: the library sizes its pinned result slots by it and cannot grow them under tickets in flight
            let enq = if self.slots_sized { (passes + 2).min(MAX_PASSES_PER_CALL) } else { MAX_PASSES_PER_CALL };
            self.slots_sized = true;
            let mut ticket = 0u64;
            let st = unsafe { gm_trk_update_all_async(self.h, self.ring, enq as u32, &mut ticket) };
            assert_eq!(st, 0, "gm_trk_update_all_async: {}", last_error());
            self.tickets.push_back((ticket, enq as u32));
            for i in 0..self.channels.len() {
                if !self.channels[i].is_active() { continue; }
                let (idx, n) = (self.planned_index(i), self.channels[i].num_samples_per_code.max(1));
                self.covered[i] = (idx + enq * n).min(head);
            }
        }
        self.collect_ready(false);
    }

    fn next_tracking_index(&self) -> usize {                             // :373-381 on the PLANNED positions (= the fields when nothing is in flight)
        (0..self.channels.len()).filter(|i| self.channels[*i].is_active())
            .map(|i| self.planned_index(i) + self.channels[i].num_samples_per_code).min().unwrap_or(0)
    }
}
impl Drop for TrackingManager {
    fn drop(&mut self) { unsafe { gm_trk_destroy(self.h); if !self.ring.is_null() { gm_ring_destroy(self.ring); } } }
}

/// The tracking stage (do_tracking.rs:384-415): same signature and control flow — wait on the ring's condvar until the
/// earliest active channel has its next code period, then process_channels while data is there — on the ticket path: the
/// thread enqueues and goes back to the condvar; it never waits for a pass or a copy (the C++ twin is
/// gnss::run_tracking in host/gnss_sdr.hpp, whose loop tests/test_gpu_stage_drivers.py holds against the synchronous one bit for bit).
pub fn run(
    multi_ring_buf: Arc<MulticastRingBuffer>,
    acq_to_trk: Receiver<AcquisitionResult>,
    trk_to_acq: Sender<TrackingMessage>,
    fs: f32,
) -> Result<(), TrackingError> {
    let mut manager = TrackingManager::new(acq_to_trk, trk_to_acq, fs);
    loop {
        let mut curr_head = multi_ring_buf.get_head();
        let mut required_idx = manager.next_tracking_index();
        if (curr_head.wrapping_sub(required_idx) as isize) < 0 {
            let mut head_guard = multi_ring_buf.notifier.lock()?;
            while (multi_ring_buf.get_head().wrapping_sub(manager.next_tracking_index()) as isize) < 0 {
                head_guard = multi_ring_buf.condvar.wait(head_guard)?;
                manager.collect_ready(false);                            // every wake-up (a block written): adopt what has finished
                if !manager.acq_to_trk.is_empty() { break; }             // a hand-over is waiting: process_channels takes it
            }
            curr_head = multi_ring_buf.get_head();
            drop(head_guard);
        }
        manager.process_channels_async(multi_ring_buf.clone());         // (also with no channel active yet: the acquisitions are taken there, :352-362)
        required_idx = manager.next_tracking_index();
        while (curr_head.wrapping_sub(required_idx) as isize) >= 0 && required_idx != 0 {
            manager.process_channels_async(multi_ring_buf.clone());

            required_idx = manager.next_tracking_index();
            curr_head = multi_ring_buf.get_head();
        }
    }
}
