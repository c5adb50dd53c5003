// mi355x.rs — the ONE module a maintainer adds to kewei/gnss-sdr-rs to put the MI355X path behind the crate's own
// acquisition / tracking API.  Mirror of include/gnss_mi355x.h (raw bindings) + safe wrappers with the
// reference's names.  Shipped as source: the build image has no Rust toolchain, so this file was NOT compiled here; the
// same ABI is exercised end to end by gnss-sdr-rs_amd/host/gnss_sdr.hpp + tests/cpp/test_host_api.cpp (C++) and by the
// ctypes mirror used in tests/ (Python).  Generated from the code blocks of INTEGRATION.md (keep the two in step).
#![allow(non_camel_case_types, dead_code)]
use crate::acquisition::do_acquisition::{AcqError, AcquisitionResult};

// ------------------------------------------------------------------ raw bindings
use num_complex::Complex32;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct GmAcqResult {            // gm_acq_result  <->  AcquisitionResult (do_acquisition.rs:93-116)
    pub prn: u8,
    pub code_phase_samples: u64,
    pub code_phase_chips: f32,
    pub carrier_freq: f32,
    pub fs: f32,
    pub mag_relative: f32,
    pub sample_global_index: u64,
    pub doppler_bin: i32,
}
#[repr(C)]
pub struct GmAcqCfg {               // gm_acq_cfg
    pub fs: f32, pub f_if: f32, pub fft_size: u32, pub n_integrations: u32, pub n_bins: u32,
    pub doppler_hz: *const f32, pub tables: *const Complex32, pub table_freq: *const f32,
    pub n_prn: u32, pub prn_ids: *const u8, pub codes: *const i8, pub code_len: u32, pub code_rate: f32,
    pub threshold: f32,
    pub decision_mode: i32,          // 0 = the reference's early exit (GM_DECIDE_REFERENCE), 1 = strongest bin
    pub strict_sum_order: i32,       // 1 = is_good_satellite's sum in the reference's 8-lane order (do_acquisition.rs:229-235)
}
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct GmTrkState {             // gm_trk_state  <->  the evolving fields of TrackingChannel (do_tracking.rs:88-116)
    pub prn: u8, pub active: u8, pub reserved: [u8; 2], pub lost_counter: u32,
    pub next_sample_index: u64, pub num_samples_per_code: u64,
    pub carrier_freq: f32, pub carrier_phase: f32, pub carrier_error: f32, pub carrier_nco: f32,
    pub code_phase: f32, pub code_error: f32, pub code_nco: f32, pub code_rate: f32,
    pub i_prompt: f32, pub q_prompt: f32,
}
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct GmTrkOut { pub ip: f32, pub qp: f32, pub ie: f32, pub qe: f32, pub il: f32, pub ql: f32,
                      pub ive: f32, pub qve: f32, pub ivl: f32, pub qvl: f32 }
#[repr(C)]
pub struct GmTrkCfg {               // gm_trk_cfg (zero = reference default)
    pub fs: f32, pub n_channels: u32, pub n_arms: u32, pub early_late_space: f32, pub very_early_late_space: f32,
    pub code_index_mode: i32, pub boc11: i32, pub codes: *const i8, pub n_codes: u32, pub code_len: u32,
    pub nominal_code_rate: f32, pub pll_bw: f32, pub pll_zeta: f32, pub pll_gain: f32, pub dll_bw: f32,
    pub dll_zeta: f32, pub dll_gain: f32, pub pll_dt: f32, pub dll_dt: f32, pub lock_threshold: f32,
    pub max_lost_epochs: u32,
}
pub enum GmAcq {} pub enum GmTrk {} pub enum GmRing {} pub enum GmComm {}

extern "C" {
    pub fn gm_init(device: c_int) -> c_int;
    pub fn gm_last_error() -> *const c_char;
    // do_acquisition.rs:252-271  (tables + AcquisitionWorker::new for every PRN)
    pub fn gm_acq_create(cfg: *const GmAcqCfg, out: *mut *mut GmAcq) -> c_int;
    pub fn gm_acq_destroy(a: *mut GmAcq) -> c_int;
    // do_acquisition.rs:302-313 + :158-226  (par_iter over workers / search_satellite)
    pub fn gm_acq_search_c32(a: *mut GmAcq, samples: *const Complex32, n: usize, local_tail: u64,
                             prn_mask: u64, results: *mut GmAcqResult, found: *mut u8) -> c_int;
    pub fn gm_acq_search_i8(a: *mut GmAcq, iq: *const i8, n: usize, local_tail: u64, prn_mask: u64,
                            results: *mut GmAcqResult, found: *mut u8) -> c_int;
    // doppler_shift.rs:10-22, :25-58
    pub fn gm_doppler_table_new(f_if: f32, doppler: f32, fs: f32, n: usize, freq_out: *mut f32, table: *mut Complex32) -> c_int;
    pub fn gm_apply_doppler_shift(s: *const Complex32, t: *const Complex32, out: *mut Complex32, n: usize) -> c_int;
    // multicast_ring_buffer.rs:46-129 (device mirror fed next to the host ring)
    pub fn gm_ring_create(buf_size: usize, out: *mut *mut GmRing) -> c_int;
    pub fn gm_ring_destroy(r: *mut GmRing) -> c_int;
    pub fn gm_ring_write_samples(r: *mut GmRing, s: *const Complex32, n: usize) -> c_int;
    // do_tracking.rs:118-158, 311-327
    pub fn gm_trk_create(cfg: *const GmTrkCfg, out: *mut *mut GmTrk) -> c_int;
    pub fn gm_trk_destroy(t: *mut GmTrk) -> c_int;
    pub fn gm_trk_start(t: *mut GmTrk, ch: u32, r: *const GmAcqResult) -> c_int;
    pub fn gm_trk_reset(t: *mut GmTrk, ch: u32) -> c_int;
    pub fn gm_trk_get_state(t: *mut GmTrk, ch: u32, out: *mut GmTrkState) -> c_int;
    // do_tracking.rs:231-272, :183-210 on caller samples; :351-371 batched over the ring
    pub fn gm_trk_correlate(t: *mut GmTrk, ch: u32, s: *const Complex32, n: usize, out: *mut GmTrkOut) -> c_int;
    pub fn gm_trk_do_work(t: *mut GmTrk, ch: u32, s: *const Complex32, n: usize, out: *mut GmTrkOut,
                          lost: *mut u8, lost_prn: *mut u8) -> c_int;
    pub fn gm_trk_update_all(t: *mut GmTrk, ring: *mut GmRing, max_epochs: u32, outs: *mut GmTrkOut,
                             processed: *mut u8, lost: *mut u8, epochs_done: *mut u32) -> c_int;
    // fft.rs:5-56
    pub fn gm_fft_c2c_f32(n: usize, dir: c_int, inout: *mut Complex32, batch: usize) -> c_int;
    pub fn gm_rfft_f32(n: usize, input: *const f32, out: *mut Complex32) -> c_int;
    // multi-GPU (nothing to replace in the reference; rayon's fan-out :302-313 becomes one process per GPU)
    pub fn gm_acq_search_dev(a: *mut GmAcq, d_samples: *const c_void, fmt: c_int, d_metrics: *mut c_void) -> c_int;
    pub fn gm_comm_get_unique_id(id: *mut u8 /* [128] */) -> c_int;
    pub fn gm_comm_init(nranks: c_int, rank: c_int, id: *const u8, out: *mut *mut GmComm) -> c_int;
    pub fn gm_comm_destroy(c: *mut GmComm) -> c_int;
    pub fn gm_acq_allgather_metrics(a: *mut GmAcq, c: *mut GmComm, d_local: *const c_void, d_all: *mut c_void) -> c_int;
    pub fn gm_acq_decide_dev(a: *mut GmAcq, d_metrics: *const c_void, n_prn: u32, prn_ids: *const u8, local_tail: u64) -> c_int;
    pub fn gm_acq_fetch_results(a: *mut GmAcq, n_prn: u32, results: *mut GmAcqResult, found: *mut u8) -> c_int;
}

// ------------------------------------------------------------------ safe wrappers
pub struct AcquisitionEngine { h: *mut GmAcq, n_prn: usize }   // replaces Vec<AcquisitionWorker> in run() (:268-271)
unsafe impl Send for AcquisitionEngine {}

impl AcquisitionEngine {
    /// what run() builds at :248-271: the Doppler grid and the 32 workers
    pub fn new(fs: f32, f_if: f32, fft_size: usize, doppler_hz: &[f32], prn_ids: &[u8], n_int: usize) -> Result<Self, AcqError> {
        let cfg = GmAcqCfg { fs, f_if, fft_size: fft_size as u32, n_integrations: n_int as u32,
            n_bins: doppler_hz.len() as u32, doppler_hz: doppler_hz.as_ptr(), tables: std::ptr::null(),
            table_freq: std::ptr::null(), n_prn: prn_ids.len() as u32, prn_ids: prn_ids.as_ptr(),
            codes: std::ptr::null(), code_len: 0, code_rate: 0.0, threshold: 7.0, decision_mode: 0, strict_sum_order: 0 };
        let mut h = std::ptr::null_mut();
        if unsafe { gm_acq_create(&cfg, &mut h) } != 0 { return Err(AcqError); }
        Ok(Self { h, n_prn: prn_ids.len() })
    }
    /// the body of `workers.par_iter_mut().enumerate().filter_map(..search_satellite..)` (:302-313)
    pub fn search(&mut self, chunk: &[Complex32], local_tail: usize, mask: u32) -> Vec<AcquisitionResult> {
        let mut raw = vec![GmAcqResult::default(); self.n_prn];
        let mut found = vec![0u8; self.n_prn];
        let st = unsafe { gm_acq_search_c32(self.h, chunk.as_ptr(), chunk.len(), local_tail as u64, mask as u64,
                                            raw.as_mut_ptr(), found.as_mut_ptr()) };
        assert_eq!(st, 0, "gm_acq_search_c32");      // the reference panics on a short chunk (:176)
        raw.iter().zip(found).filter(|(_, f)| *f != 0).map(|(r, _)| AcquisitionResult {
            prn: r.prn, code_phase_samples: r.code_phase_samples as usize, code_phase_chips: r.code_phase_chips,
            carrier_freq: r.carrier_freq, fs: r.fs, mag_relative: r.mag_relative,
            sample_global_index: r.sample_global_index as usize }).collect()
    }
}
impl Drop for AcquisitionEngine { fn drop(&mut self) { unsafe { gm_acq_destroy(self.h); } } }

// ------------------------------------------------------------------ TrackingManager::process_channels (do_tracking.rs:351-371)
// the body that replaces the par_iter_mut fan-out (shown as it would sit inside `impl TrackingManager`):
/*
while let Ok(msg) = self.acq_to_trk.try_recv() {                       // unchanged (:352-362)
    if let Some(ch) = self.idle_channel() { let _ = self.trk_to_acq.send(TrackingMessage::SatelliteLocked(msg.prn));
        unsafe { gm_trk_start(self.h, ch, &to_raw(&msg)); } } }
let mut lost = vec![0u8; LOOP_MS * self.n]; let mut done = 0u32;
unsafe { gm_trk_update_all(self.h, self.ring, LOOP_MS as u32, std::ptr::null_mut(), std::ptr::null_mut(),
                           lost.as_mut_ptr(), &mut done); }            // replaces par_iter_mut().for_each(update) (:364-371)
for (i, l) in lost.iter().enumerate() { if *l != 0 { let _ = self.trk_to_acq.send(TrackingMessage::SatelliteLost(0)); } }
*/
