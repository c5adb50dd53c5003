// doppler_shift.rs — DESTINATION: src/mi355x/doppler_shift.rs (module crate::mi355x::doppler_shift; src/acquisition/doppler_shift.rs
// stays as it is, and AcquisitionWorker below keeps taking ITS DopplerShiftTable).  Same items, same signatures, on the library:
// `DopplerShiftTable::new` (:10-22) and `apply_doppler_shift` (:25-40) keep their meaning bit for bit: the table is built
// on the host with the platform cosf/sinf exactly like the reference (gm_doppler_table_new), the product uses the
// reference's rounding sequence (a*c - b*d, a*d + b*c, no FMA) and touches only the first 4*floor(n/4) outputs.
use crate::mi355x::*;
use num_complex::Complex32;

pub struct DopplerShiftTable {
    pub doppler_freq_hz: f32,
    pub table: Vec<Complex32>,
}

impl DopplerShiftTable {
    pub fn new(f_if: f32, doppler_freq_hz: f32, fs: f32, num_samples: usize) -> Self {
        let mut table = vec![Complex32::new(0.0, 0.0); num_samples];
        let mut carr_freq = 0.0f32;            // the reference stores f_if + doppler here (:20)
        let st = unsafe { gm_doppler_table_new(f_if, doppler_freq_hz, fs, num_samples, &mut carr_freq, table.as_mut_ptr()) };
        assert_eq!(st, 0, "gm_doppler_table_new: {}", last_error());
        Self { doppler_freq_hz: carr_freq, table }
    }
}

pub fn apply_doppler_shift(samples: &[Complex32], doppler_table: &DopplerShiftTable, output: &mut [Complex32]) {
    // the reference reads samples.len()/4 chunks from all three slices and panics on a short one (:26-38)
    assert!(doppler_table.table.len() >= samples.len() / 4 * 4 && output.len() >= samples.len() / 4 * 4);
    let st = unsafe { gm_apply_doppler_shift(samples.as_ptr(), doppler_table.table.as_ptr(), output.as_mut_ptr(), samples.len()) };
    assert_eq!(st, 0, "gm_apply_doppler_shift: {}", last_error());
}
