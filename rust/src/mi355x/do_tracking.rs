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
//   * Fields poked between calls are honoured on the batched path too: process_channels writes every active channel's pub
//     fields into the manager's handle before the launch (gm_trk_set_state) and reads them back after it.
//   * run — the tracking stage's thread body with the reference's signature and control flow (:384-415): main.rs:216-227 calls
//     it through `use gnss_sdr_rs::mi355x::do_tracking;` instead of `...::tracking::do_tracking;`.
use crate::acquisition::do_acquisition::{AcquisitionResult, ChannelState};
use crate::mi355x::*;
use crate::tracking::do_tracking::{LoopFilter, TrackingError, TrackingMessage};
use crate::utilities::multicast_ring_buffer::MulticastRingBuffer;
use crossbeam_channel::{Receiver, Sender};   // the crate's channels (do_tracking.rs:8, main.rs:183-184)
use num_complex::Complex32;
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
fn to_raw(r: &AcquisitionResult) -> GmAcqResult {
    GmAcqResult { prn: r.prn, code_phase_samples: r.code_phase_samples as u64, code_phase_chips: r.code_phase_chips,
                  carrier_freq: r.carrier_freq, fs: r.fs, mag_relative: r.mag_relative,
                  sample_global_index: r.sample_global_index as u64, doppler_bin: -1 }
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
    mirrored: usize,               // absolute index up to which the mirror holds the host ring's samples
}
unsafe impl Send for TrackingManager {}

impl TrackingManager {
    pub fn new(
        acq_to_trk: Receiver<AcquisitionResult>,
        trk_to_acq: Sender<TrackingMessage>,
        fs: f32,
    ) -> Self {
        let mut h = std::ptr::null_mut();
        let st = unsafe { gm_trk_create(&trk_cfg(fs, NUM_OF_CHANNELS as u32), &mut h) };
        assert_eq!(st, 0, "gm_trk_create: {}", last_error());
        Self { channels: (0..NUM_OF_CHANNELS).map(|id| TrackingChannel::new(id as u8, fs)).collect(),
               acq_to_trk, trk_to_acq, h, ring: std::ptr::null_mut(), mirrored: 0 }
    }

    pub fn process_channels(&mut self, multi_ring_buf: Arc<MulticastRingBuffer>) {
        while let Ok(msg) = self.acq_to_trk.try_recv() {                 // unchanged (:352-362)
            if let Some((i, channel)) = self.channels.iter_mut().enumerate().find(|(_, c)| c.state == ChannelState::Idle) {
                let _ = self.trk_to_acq.send(TrackingMessage::SatelliteLocked(msg.prn));
                let st = unsafe { gm_trk_start(self.h, i as u32, &to_raw(&msg)) };
                assert_eq!(st, 0, "gm_trk_start: {}", last_error());
                channel.start(msg);
            }
        }
        // bring the device mirror up to the host ring's head (rf_thread writes the host ring; a front-end that writes the
        // mirror directly — gm_frontend_write_ring — makes this copy unnecessary)
        let head = multi_ring_buf.get_head();
        if self.ring.is_null() {
            let st = unsafe { gm_ring_create(multi_ring_buf.buffer.len(), &mut self.ring) };   // same power-of-two size (:46-61)
            assert_eq!(st, 0, "gm_ring_create: {}", last_error());
        }
        while head > self.mirrored {                                     // at most one ring's worth per write
            let n = (head - self.mirrored).min(multi_ring_buf.buffer.len());
            let mut tmp = vec![Complex32::new(0.0, 0.0); n];
            multi_ring_buf.copy_to_slice(self.mirrored, &mut tmp);
            let st = unsafe { gm_ring_write_samples(self.ring, tmp.as_ptr(), tmp.len()) };
            assert_eq!(st, 0, "gm_ring_write_samples: {}", last_error());
            self.mirrored += n;
        }
        // the pub fields are the truth between calls (the reference's tests poke them): every active channel's fields go into
        // the batched handle before the launch.  After a call they equal the device state (read back below), so this changes
        // nothing unless the caller wrote a field.
        for (i, ch) in self.channels.iter().enumerate() {
            if ch.is_active() {
                let s = ch.raw_state();
                let st = unsafe { gm_trk_set_state(self.h, i as u32, &s) };
                assert_eq!(st, 0, "gm_trk_set_state: {}", last_error());
            }
        }
        // channels.par_iter_mut().filter(is_active).for_each(update) (:364-371), up to PROCESS_EPOCHS code periods per call
        let n = self.channels.len();
        let mut lost = vec![0u8; PROCESS_EPOCHS * n];
        let mut done = 0u32;
        let st = unsafe { gm_trk_update_all(self.h, self.ring, PROCESS_EPOCHS as u32, std::ptr::null_mut(), std::ptr::null_mut(),
                                            lost.as_mut_ptr(), &mut done) };
        assert_eq!(st, 0, "gm_trk_update_all: {}", last_error());
        for (i, ch) in self.channels.iter_mut().enumerate() {            // the pub fields follow the device state
            let mut s = GmTrkState::default();
            unsafe { gm_trk_get_state(self.h, i as u32, &mut s); }
            ch.prn = s.prn; ch.lost_counter = s.lost_counter; ch.next_sample_index = s.next_sample_index as usize;
            ch.num_samples_per_code = s.num_samples_per_code as usize; ch.carrier_freq = s.carrier_freq;
            ch.carrier_phase = s.carrier_phase; ch.carrier_error = s.carrier_error; ch.carrier_nco = s.carrier_nco;
            ch.code_phase = s.code_phase; ch.code_error = s.code_error; ch.code_nco = s.code_nco; ch.code_rate = s.code_rate;
            ch.i_prompt = s.i_prompt; ch.q_prompt = s.q_prompt;
            if s.active == 0 { ch.state = ChannelState::Idle; }
        }
        for l in lost.iter() {                                           // SatelliteLost carries prn 0 (:199-201)
            if *l != 0 { let _ = self.trk_to_acq.send(TrackingMessage::SatelliteLost(0)); }
        }
    }

    fn next_tracking_index(&self) -> usize {                             // unchanged (:373-381)
        self.channels.iter().filter(|c| c.is_active()).map(|c| c.next_sample_index + c.num_samples_per_code).min().unwrap_or(0)
    }
}
impl Drop for TrackingManager {
    fn drop(&mut self) { unsafe { gm_trk_destroy(self.h); if !self.ring.is_null() { gm_ring_destroy(self.ring); } } }
}

/// The tracking stage (do_tracking.rs:384-415): same signature and control flow — wait on the ring's condvar until the
/// earliest active channel has its next code period, then process_channels while data is there.
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
            }
            curr_head = multi_ring_buf.get_head();
            drop(head_guard);
        }
        while (curr_head.wrapping_sub(required_idx) as isize) >= 0 {
            manager.process_channels(multi_ring_buf.clone());
            required_idx = manager.next_tracking_index();
            curr_head = multi_ring_buf.get_head();
        }
    }
}
