// do_acquisition.rs — the items of src/acquisition/do_acquisition.rs that sit on the hot path, on the MI355X library.
// Unchanged in the crate and therefore not repeated here: AcquisitionResult (:93-116), AcquisitionManager / SearchMode /
// ChannelState (:25-74), AcqError (:76-91), the constants (:17-23).
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
use crate::acquisition::do_acquisition::{AcqError, AcquisitionResult};
use crate::acquisition::doppler_shift::DopplerShiftTable;
use crate::mi355x::*;
use num_complex::Complex32;

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
                codes: std::ptr::null(), code_len: 0, code_rate: 0.0, threshold: 7.0, decision_mode: 0, strict_sum_order: 0 };
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
        assert_eq!(st, 0, "gm_acq_search_c32: {}", last_error());      // the reference panics on a short chunk (:176)
        raw.iter().zip(found).filter(|(_, f)| *f != 0).map(|(r, _)| to_result(r)).collect()
    }
}
impl Drop for AcquisitionEngine { fn drop(&mut self) { unsafe { gm_acq_destroy(self.h); } } }

// run() (:241-327) keeps its loop; the two changed statements are
//     let mut engine = AcquisitionEngine::new(freq_sampling_hz, f_if, fft_size, &doppler_hz, &prns, LONG_SAMPLES_LENGTH)?;   // for :252-271
//     let results = engine.search(&chunk_samples, local_tail, mask);                                                       // for :302-313
