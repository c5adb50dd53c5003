// do_acquisition.rs — DESTINATION: src/mi355x/do_acquisition.rs (module crate::mi355x::do_acquisition, a new sibling of
// crate::acquisition::do_acquisition, which stays in the crate untouched).  The items of src/acquisition/do_acquisition.rs that
// sit on the hot path, on the MI355X library.  Imported from the reference's module because they do not change:
// AcquisitionResult (:93-116), AcquisitionManager / SearchMode (:33-74), AcqError (:76-91), PRN_SEARCH_ACQUISITION_TOTAL (:22).
// The reference's private constants (:20-23) are restated below.
//
// Two shapes are offered:
//   * AcquisitionWorker — the reference's per-PRN type with the reference's signatures (:130-226): a drop-in for code
//     that calls `worker.search_satellite(&chunk, &tables, local_tail, n)` one PRN at a time.  The reference takes the
//     tables per call (:158-163); here each worker owns a one-PRN handle that holds a device copy of them and is rebuilt
//     whenever the tables it is called with DIFFER IN CONTENT from the ones it was built for: the key is every table's
//     doppler_freq_hz and length plus three of its phasors (not its address — a Vec rebuilt at the same address with other
//     Doppler values must not reuse the handle).  Code that edits phasors in place beyond that calls invalidate().
//   * AcquisitionEngine — all workers of run() in ONE handle (:268-271): `search` is the body of
//     `workers.par_iter_mut().enumerate().filter_map(..)` (:302-313) as one batched launch; this is the fast path.
//   * run — the acquisition stage's thread body with the reference's signature and control flow (:241-327): main.rs:204-214
//     calls it through `use gnss_sdr_rs::mi355x::do_acquisition;` instead of `...::acquisition::do_acquisition;`.
use crate::acquisition::do_acquisition::{AcqError, AcquisitionManager, AcquisitionResult, PRN_SEARCH_ACQUISITION_TOTAL};
use crate::acquisition::doppler_shift::DopplerShiftTable;
use crate::mi355x::*;
use crate::tracking::do_tracking::TrackingMessage;
use crate::utilities::multicast_ring_buffer::MulticastRingBuffer;
use crossbeam_channel::{Receiver, Sender};
use num_complex::Complex32;
use std::collections::HashSet;
use std::sync::Arc;

const FREQ_SEARCH_ACQUISITION_HZ: f32 = 14e3;       // :20
const FREQ_SEARCH_STEP_HZ: u16 = 500;               // :21
const LONG_SAMPLES_LENGTH: usize = 10;              // :23 (ms)

fn to_result(r: &GmAcqResult) -> AcquisitionResult {
    AcquisitionResult { prn: r.prn, code_phase_samples: r.code_phase_samples as usize, code_phase_chips: r.code_phase_chips,
                        carrier_freq: r.carrier_freq, fs: r.fs, mag_relative: r.mag_relative,
                        sample_global_index: r.sample_global_index as usize }
}

pub struct AcquisitionWorker {
    prn: u8,
    fft_size: usize,
    freq_sampling_hz: f32,
    h: *mut GmAcq,                 // built on the first search_satellite from the caller's tables
    tables_key: Vec<u32>,          // content fingerprint of the tables (+ num_integrations) the handle was built for
}

/// what identifies a set of Doppler tables by CONTENT: per table its frequency, its length and the phasors at 1, len/2, len-1
/// (DopplerShiftTable::new makes every phasor a function of (f_if + doppler, fs, i), doppler_shift.rs:14-18)
fn tables_fingerprint(doppler_table: &[DopplerShiftTable], num_integrations: usize) -> Vec<u32> {
    let mut k = Vec::with_capacity(2 + 8 * doppler_table.len());
    k.push(doppler_table.len() as u32);
    k.push(num_integrations as u32);
    for t in doppler_table {
        k.push(t.doppler_freq_hz.to_bits());
        k.push(t.table.len() as u32);
        let n = t.table.len();
        for i in [1usize.min(n.saturating_sub(1)), n / 2, n.saturating_sub(1)] {
            if n > 0 { k.push(t.table[i].re.to_bits()); k.push(t.table[i].im.to_bits()); }
        }
    }
    k
}
unsafe impl Send for AcquisitionWorker {}     // rayon moves one &mut worker to each task (:302-313)

impl AcquisitionWorker {
    pub fn new(prn: u8, fft_size: usize, freq_sampling_hz: f32) -> Self {
        Self { prn, fft_size, freq_sampling_hz, h: std::ptr::null_mut(), tables_key: Vec::new() }
    }

    /// forget the cached handle: the next search_satellite rebuilds it from the tables it is given
    pub fn invalidate(&mut self) {
        if !self.h.is_null() { unsafe { gm_acq_destroy(self.h); } self.h = std::ptr::null_mut(); }
        self.tables_key.clear();
    }

    pub fn search_satellite(
        &mut self,
        samples_chunk: &[Complex32],
        doppler_table: &[DopplerShiftTable],
        local_tail: usize,
        num_integrations: usize,
    ) -> Option<AcquisitionResult> {
        let key = tables_fingerprint(doppler_table, num_integrations);
        if self.h.is_null() || key != self.tables_key {
            if !self.h.is_null() { unsafe { gm_acq_destroy(self.h); } self.h = std::ptr::null_mut(); }
            // the caller's tables, laid out [n_bins][fft_size] as gm_acq_cfg.tables wants them
            let mut flat = Vec::with_capacity(doppler_table.len() * self.fft_size);
            for t in doppler_table { flat.extend_from_slice(&t.table[..self.fft_size]); }   // a short table panics like :178
            let freqs: Vec<f32> = doppler_table.iter().map(|t| t.doppler_freq_hz).collect();
            let ids = [self.prn];
            let cfg = GmAcqCfg { fs: self.freq_sampling_hz, f_if: 0.0, fft_size: self.fft_size as u32,
                n_integrations: num_integrations as u32, n_bins: doppler_table.len() as u32, doppler_hz: std::ptr::null(),
                tables: flat.as_ptr(), table_freq: freqs.as_ptr(), n_prn: 1, prn_ids: ids.as_ptr(),
                codes: std::ptr::null(), code_len: 0, code_rate: 0.0, threshold: 7.0, decision_mode: 0, strict_sum_order: 0, reference_products: 0 };
            let st = unsafe { gm_acq_create(&cfg, &mut self.h) };
            assert_eq!(st, 0, "gm_acq_create: {}", last_error());     // e.g. prn 0 / 33: the reference panics in ::new (:133)
            self.tables_key = key;
        }
        let n = self.fft_size * num_integrations;
        assert!(samples_chunk.len() >= n);                            // the reference's slice index panics (:176)
        let mut raw = GmAcqResult::default();
        let mut found = 0u8;
        let st = unsafe { gm_acq_search_c32(self.h, samples_chunk.as_ptr(), n, local_tail as u64, 1, &mut raw, &mut found) };
        assert_eq!(st, 0, "gm_acq_search_c32: {}", last_error());
        if found != 0 { Some(to_result(&raw)) } else { None }
    }
}
impl Drop for AcquisitionWorker { fn drop(&mut self) { if !self.h.is_null() { unsafe { gm_acq_destroy(self.h); } } } }

pub struct AcquisitionEngine { h: *mut GmAcq, n_prn: usize }   // replaces Vec<AcquisitionWorker> in run() (:268-271)
unsafe impl Send for AcquisitionEngine {}

impl AcquisitionEngine {
    /// what run() builds at :248-271: the Doppler grid and the 32 workers
    pub fn new(fs: f32, f_if: f32, fft_size: usize, doppler_hz: &[f32], prn_ids: &[u8], n_int: usize) -> Result<Self, AcqError> {
        let cfg = GmAcqCfg { fs, f_if, fft_size: fft_size as u32, n_integrations: n_int as u32,
            n_bins: doppler_hz.len() as u32, doppler_hz: doppler_hz.as_ptr(), tables: std::ptr::null(),
            table_freq: std::ptr::null(), n_prn: prn_ids.len() as u32, prn_ids: prn_ids.as_ptr(),
            codes: std::ptr::null(), code_len: 0, code_rate: 0.0, threshold: 7.0, decision_mode: 0, strict_sum_order: 0, reference_products: 0 };
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
        assert_eq!(st, 0, "gm_acq_search_c32: {}", last_error());      // the reference panics on a short chunk (:176)
        raw.iter().zip(found).filter(|(_, f)| *f != 0).map(|(r, _)| to_result(r)).collect()
    }
}
impl Drop for AcquisitionEngine { fn drop(&mut self) { unsafe { gm_acq_destroy(self.h); } } }

/// The acquisition stage (do_acquisition.rs:241-327): same signature, same pacing, same messages.  What differs from the
/// reference's body: the 29 Doppler tables and the 32 workers are ONE engine (:252-271), and the rayon fan-out over the workers
/// (:302-313) is one batched search on the GPU.
pub fn run(
    multi_buffer: Arc<MulticastRingBuffer>,
    freq_sampling_hz: f32,
    f_if: f32,
    to_tracking: Sender<AcquisitionResult>,
    from_tracking: Receiver<TrackingMessage>,
) -> Result<(), AcqError> {
    let capacity = (FREQ_SEARCH_ACQUISITION_HZ as u16 / FREQ_SEARCH_STEP_HZ) as usize + 1;
    let fft_size = (freq_sampling_hz / (1.023e6_f32 / 1023.0_f32)).round() as usize;                  // :249-251
    let doppler_hz: Vec<f32> = (0..capacity)
        .map(|i| -FREQ_SEARCH_ACQUISITION_HZ / 2.0 + i as f32 * FREQ_SEARCH_STEP_HZ as f32)             // :253-255
        .collect();
    let prns: Vec<u8> = (1..=PRN_SEARCH_ACQUISITION_TOTAL).collect();
    let mut engine = AcquisitionEngine::new(freq_sampling_hz, f_if, fft_size, &doppler_hz, &prns, LONG_SAMPLES_LENGTH)?;

    let mut active_prns: HashSet<u8> = HashSet::new();
    let mut acq_manager = AcquisitionManager::new();
    let samples_integration_size = fft_size * LONG_SAMPLES_LENGTH;
    let mut chunk_samples = vec![Complex32::new(0.0, 0.0); samples_integration_size];
    let mut last_run = std::time::Instant::now();

    loop {
        while let Ok(msg) = from_tracking.try_recv() {
            match msg {
                TrackingMessage::SatelliteLost(prn) => { active_prns.remove(&prn); }
                TrackingMessage::SatelliteLocked(prn) => { active_prns.insert(prn); }
            }
        }
        acq_manager.update_mode(active_prns.len());
        let (interval_ms, mask) = acq_manager.get_pacing_and_list(&active_prns);
        if last_run.elapsed().as_millis() < interval_ms as u128 {
            std::thread::sleep(std::time::Duration::from_millis(50));
            continue;
        }
        let head = multi_buffer.get_head();
        if (head.wrapping_sub(samples_integration_size) as isize) >= 0 {
            let local_tail = head.wrapping_sub(samples_integration_size);
            multi_buffer.copy_to_slice(local_tail, &mut chunk_samples);
            for result in engine.search(&chunk_samples, local_tail, mask) {                            // :302-313
                let prn = result.prn;
                if to_tracking.send(result).is_ok() {
                    active_prns.insert(prn);
                }
            }
            last_run = std::time::Instant::now();
        } else {
            std::thread::sleep(std::time::Duration::from_millis(1));
        }
    }
}
