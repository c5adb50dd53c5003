// mi355x.rs — DESTINATION: src/mi355x.rs of kewei/gnss-sdr-rs (a NEW module; `pub mod mi355x;` is added to src/lib.rs after
// `pub mod constants;`, lib.rs:14).  Raw bindings of include/gnss_mi355x.h: the `extern "C"` block and the
// #[repr(C)] mirrors.  The wrappers that carry the reference's own names and signatures are its SUBMODULES — new sibling
// modules of the reference's own, which stay in the crate untouched (nothing is edited in place, nothing is replaced):
//   src/mi355x/doppler_shift.rs   DopplerShiftTable::new, apply_doppler_shift                 (cf. src/acquisition/doppler_shift.rs:5-40)
//   src/mi355x/do_acquisition.rs  AcquisitionWorker::{new, search_satellite}, AcquisitionEngine, run
//                                                                                              (cf. src/acquisition/do_acquisition.rs:130-226, 241-327)
//   src/mi355x/do_tracking.rs     TrackingChannel (22 pub fields, every method), TrackingManager::{new, process_channels}, run
//                                                                                              (cf. src/tracking/do_tracking.rs:88-415)
//   src/mi355x/fft.rs             FFT<f32>, RealFFT<f32>                                       (cf. src/fft.rs:5-56)
// They import the items that do NOT change (AcquisitionResult, AcqError, ChannelState, AcquisitionManager, LoopFilter,
// TrackingMessage, TrackingError, MulticastRingBuffer, the reference's DopplerShiftTable) from the reference's modules.
// main.rs switches two `use` lines (rust/patches/main_rs.diff); the thread wiring at main.rs:204-227 is unchanged.
// Shipped as source: the build image has no Rust toolchain, so these files were NOT compiled here; the same ABI is
// exercised end to end by gnss-sdr-rs_amd/host/gnss_sdr.hpp + tests/cpp/test_host_api.cpp (C++) and by the ctypes mirror
// used in tests/ (Python); tests/test_abi_and_host.py checks struct layouts, extern names, wrapper signatures and that every
// `use crate::...` path of these files names an item that exists in the reference's module tree or in these files.
#![allow(non_camel_case_types, dead_code)]

pub mod doppler_shift;
pub mod do_acquisition;
pub mod do_tracking;
pub mod fft;

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
    pub reference_products: i32,     // 1 = x conj(code) and norm_sqr() rounded as num-complex rounds them (no fused multiply-add; :184-192)
}
#[repr(C)] #[derive(Clone, Copy, Debug, Default, PartialEq)]
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
    pub strict_libm: i32,           // 1 = the carrier's cos / sin are glibc's cosf / sinf restated on the device (bit-identical products)
    pub strict_sum_order: i32,      // 1 = the correlator sums added sample by sample like do_tracking.rs:256-262 (with strict_libm: bit-identical state)
    pub share_device: i32,          // 1 = a receiver: the tracking kernel leaves room for the front-end's and the acquisition's kernels beside it (ABI 6)
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
    pub fn gm_ring_get_head(r: *mut GmRing, head: *mut u64) -> c_int;
    pub fn gm_ring_write_samples_async(r: *mut GmRing, s: *const Complex32, n: usize) -> c_int;
    pub fn gm_ring_flush(r: *mut GmRing) -> c_int;
    pub fn gm_ring_get_enqueued_head(r: *mut GmRing, head: *mut u64) -> c_int;
    // do_tracking.rs:118-158, 311-327
    pub fn gm_trk_create(cfg: *const GmTrkCfg, out: *mut *mut GmTrk) -> c_int;
    pub fn gm_trk_destroy(t: *mut GmTrk) -> c_int;
    pub fn gm_trk_start(t: *mut GmTrk, ch: u32, r: *const GmAcqResult) -> c_int;
    pub fn gm_trk_reset(t: *mut GmTrk, ch: u32) -> c_int;
    pub fn gm_trk_get_state(t: *mut GmTrk, ch: u32, out: *mut GmTrkState) -> c_int;
    pub fn gm_trk_set_state(t: *mut GmTrk, ch: u32, state: *const GmTrkState) -> c_int;
    // do_tracking.rs:274-277, :52-71
    pub fn gm_trk_get_ca_chip(t: *mut GmTrk, ch: u32, phase: f32, chip: *mut f32) -> c_int;
    pub fn gm_loop_filter_new(noise_bw: f32, damping: f32, gain: f32, tau1: *mut f32, tau2: *mut f32) -> c_int;
    pub fn gm_loop_filter_update(tau1: f32, tau2: f32, d_err: f32, err: f32, dt: f32) -> f32;
    // do_tracking.rs:231-272, :183-210 on caller samples; :351-371 batched over the ring
    pub fn gm_trk_correlate(t: *mut GmTrk, ch: u32, s: *const Complex32, n: usize, out: *mut GmTrkOut) -> c_int;
    pub fn gm_trk_do_work(t: *mut GmTrk, ch: u32, s: *const Complex32, n: usize, out: *mut GmTrkOut,
                          lost: *mut u8, lost_prn: *mut u8) -> c_int;
    pub fn gm_trk_update_all(t: *mut GmTrk, ring: *mut GmRing, max_epochs: u32, outs: *mut GmTrkOut,
                             processed: *mut u8, lost: *mut u8, epochs_done: *mut u32) -> c_int;
    /// the same passes ordered on the DEVICE behind what the ring's writer has enqueued (the Condvar wait of :392-406 without a host wait)
    pub fn gm_trk_update_all_async(t: *mut GmTrk, ring: *mut GmRing, max_epochs: u32, ticket: *mut u64) -> c_int;
    /// states (ABI 7): the channel records as they stood behind THAT call's passes; a collect that fails has consumed the ticket
    pub fn gm_trk_collect(t: *mut GmTrk, ticket: u64, wait: c_int, outs: *mut GmTrkOut, processed: *mut u8, lost: *mut u8,
                          states: *mut GmTrkState, epochs_done: *mut u32, ready: *mut c_int) -> c_int;
    /// every channel's record in one synchronisation + one copy; `which`: NULL = all, else per-channel flags
    pub fn gm_trk_get_states(t: *mut GmTrk, out: *mut GmTrkState) -> c_int;
    pub fn gm_trk_set_states(t: *mut GmTrk, states: *const GmTrkState, which: *const u8) -> c_int;
    // fft.rs:5-56
    pub fn gm_fft_c2c_f32(n: usize, dir: c_int, inout: *mut Complex32, batch: usize) -> c_int;
    pub fn gm_fft_power_spectrum_f32(n: usize, inout: *mut Complex32, power: *mut f32) -> c_int;
    pub fn gm_rfft_f32(n: usize, input: *const f32, out: *mut Complex32) -> c_int;
    // multi-GPU (nothing to replace in the reference; rayon's fan-out :302-313 becomes one process per GPU)
    pub fn gm_acq_search_dev(a: *mut GmAcq, d_samples: *const c_void, fmt: c_int, d_metrics: *mut c_void) -> c_int;
    pub fn gm_comm_get_unique_id(id: *mut u8 /* [128] */) -> c_int;
    pub fn gm_comm_init(nranks: c_int, rank: c_int, id: *const u8, out: *mut *mut GmComm) -> c_int;
    pub fn gm_comm_destroy(c: *mut GmComm) -> c_int;
    pub fn gm_acq_allgather_metrics(a: *mut GmAcq, c: *mut GmComm, d_local: *const c_void, d_all: *mut c_void) -> c_int;
    pub fn gm_acq_allgather_metrics_async(a: *mut GmAcq, c: *mut GmComm, d_local: *const c_void, d_all: *mut c_void) -> c_int;
    pub fn gm_comm_wait(c: *mut GmComm, hip_stream: *mut c_void) -> c_int;
    pub fn gm_comm_allgather_words(c: *mut GmComm, d_local: *const c_void, d_all: *mut c_void, words: usize, hip_stream: *mut c_void) -> c_int;
    pub fn gm_grid_assemble_dev(d_gathered: *const c_void, nranks: u32, p_max: u32, n_bins: u32, d_row_map: *const u32,
                                n_rows: u32, d_out: *mut c_void, hip_stream: *mut c_void) -> c_int;
    pub fn gm_acq_decide_planes_dev(d_max: *const f32, d_argmax: *const u32, d_sum: *const f32, n_prn: u32, n_bins: u32,
                                    d_prn_ids: *const u8, d_table_freq: *const f32, fft_size: u32, fs: f32, code_rate: f32,
                                    threshold: f32, decision_mode: c_int, local_tail: u64, d_results: *mut GmAcqResult,
                                    d_found: *mut u8, hip_stream: *mut c_void) -> c_int;
    pub fn gm_acq_decide_dev(a: *mut GmAcq, d_metrics: *const c_void, n_prn: u32, prn_ids: *const u8, local_tail: u64) -> c_int;
    pub fn gm_acq_fetch_results(a: *mut GmAcq, n_prn: u32, results: *mut GmAcqResult, found: *mut u8) -> c_int;
    /// back-to-back dwells: the decision rides inside the next search's first kernel
    pub fn gm_acq_set_deferred_decision(a: *mut GmAcq, on: c_int) -> c_int;
    /// back-to-back dwells: stage F of the next dwell beside the current stage C (pays at N = 16368)
    pub fn gm_acq_prepare_dev(a: *mut GmAcq, d_samples: *const c_void, fmt: c_int, ready_stream: *mut c_void, token: *mut u64) -> c_int;
    pub fn gm_acq_search_prepared_dev(a: *mut GmAcq, token: u64, d_metrics: *mut c_void) -> c_int;
    pub fn gm_acq_drop_prepared(a: *mut GmAcq) -> c_int;
}

/// status -> the last error text of the library (for panics that mirror the reference's)
pub fn last_error() -> String {
    unsafe { std::ffi::CStr::from_ptr(gm_last_error()).to_string_lossy().into_owned() }
}
