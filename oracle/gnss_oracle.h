/*
 * gnss_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the acquisition + tracking hot path of kewei/gnss-sdr-rs.
 * Every function cites the reference file:line it follows (paths relative to the
 * reference checkout, e.g. src/acquisition/do_acquisition.rs:158-226).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.  The product library (gnss-sdr-rs_amd/) never links,
 * imports or calls it.
 *
 * Pinning status (see oracle/README.md and DESIGN.md):
 *   - C/A code table: pinned by the reference's PRN-1 known-answer vector
 *     (src/bk/gps_ca_prn.rs:72-124) and a digest of GPS_CA_CODE_32_PRN.
 *   - AcquisitionManager: pinned by src/acquisition/do_acquisition.rs:339-395.
 *   - MulticastRingBuffer: pinned by src/utilities/multicast_ring_buffer.rs:147-209.
 *   - Loop-filter constants: pinned by src/tracking/do_tracking.rs:16-28,60-64.
 *   - FFT numerics (rustfft 6.1.0, external, not in /root/reference) and the
 *     search / correlator outputs on the reference's (missing) IF capture:
 *     PARITY UNPINNED by the reference's own tests.  The FFT here is a plain
 *     f32 mixed-radix Stockham restating the published DFT definition
 *     (forward e^{-j2pi kn/N}, inverse e^{+...}, neither normalised).
 *
 * Build: -O2 -ffp-contract=off (rustc never contracts a*b+c into an FMA).
 */
#ifndef GNSS_ORACLE_H
#define GNSS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } orc_c32;

/* ---------------- constants (src/constants/gps_property_constants.rs:3-30) */
#define ORC_GPS_L1_CA_CODE_RATE_CHIPS_PER_S 1.023e6f
#define ORC_GPS_L1_CA_CODE_LENGTH_CHIPS 1023.0f
#define ORC_PRN_SEARCH_ACQUISITION_TOTAL 32

/* ---------------- C/A code (src/constants/gps_ca_constants.rs, src/utilities/ca_code.rs) */
/* Row r of GPS_CA_CODE_32_PRN (r = 0..31 <-> PRN r+1), values +1/-1 (bit 1 -> +1).
 * Regenerated from the IS-GPS-200 G1/G2 generator instead of copying the table.
 * Returns 0, or -1 if row is out of 0..31 (the reference panics: index out of bounds). */
int orc_ca_code_row(int row, int8_t out[1023]);
/* generate_ca_code_samples (src/utilities/ca_code.rs:12-27).  Returns the number of
 * samples n = round(fs/(code_rate/1023)); writes min(n,cap) samples.  Returns -1 if the
 * reference would panic (prn not 1..32, or chip index 1023 reached). */
long orc_generate_ca_code_samples(int prn, float code_rate, float fs, int8_t *out, size_t cap);
/* the length alone (closed form of the above, ca_code.rs:13-16) */
size_t orc_num_samples_per_code(float code_rate, float fs);

/* ---------------- Doppler wipe-off (src/acquisition/doppler_shift.rs) */
/* DopplerShiftTable::new :10-22 ; returns the stored doppler_freq_hz (= f_if + doppler, :20) */
float orc_doppler_table_new(float f_if, float doppler_hz, float fs, size_t n, orc_c32 *table);
/* apply_doppler_shift + multiply_simd_block :25-58 ; touches only the first 4*floor(n/4) outputs */
void orc_apply_doppler_shift(const orc_c32 *samples, const orc_c32 *table, orc_c32 *out, size_t n);

/* ---------------- FFT (rustfft 6.1.0 call sites: do_acquisition.rs:132-143,182,188; src/fft.rs:13-23) */
typedef struct orc_fft_plan orc_fft_plan;
orc_fft_plan *orc_fft_plan_create(size_t n, int inverse);
void orc_fft_plan_destroy(orc_fft_plan *p);
/* in-place, unnormalised; scratch is owned by the plan (one plan per thread) */
void orc_fft_exec(orc_fft_plan *p, orc_c32 *data);
/* src/fft.rs:21-29 FFT<f32>::execute / power_spectrum */
int orc_fft_forward(orc_c32 *data, size_t n);
int orc_fft_power_spectrum(orc_c32 *data, size_t n, float *power);
/* src/fft.rs:32-56 RealFFT<f32>::execute: real input of length n -> n/2+1 complex bins */
int orc_rfft_forward(const float *in, size_t n, orc_c32 *out);

/* ---------------- Acquisition (src/acquisition/do_acquisition.rs) */
typedef struct {            /* AcquisitionResult :93-116 */
    uint8_t prn;
    uint64_t code_phase_samples;
    float code_phase_chips;
    float carrier_freq;
    float fs;
    float mag_relative;
    uint64_t sample_global_index;
    int32_t doppler_bin;    /* extra: index of the winning table (not in the reference struct) */
} orc_acq_result;

typedef struct orc_acq_worker orc_acq_worker;
/* AcquisitionWorker::new :130-156.  `code` may be NULL (GPS C/A row prn-1), or a custom
 * +-1 chip sequence of length code_len (used for the non-reference constellations). */
orc_acq_worker *orc_acq_worker_new(uint8_t prn, size_t fft_size, float fs);
orc_acq_worker *orc_acq_worker_new_custom(uint8_t prn, size_t fft_size, float fs,
                                          const int8_t *code, size_t code_len, float code_rate);
void orc_acq_worker_free(orc_acq_worker *w);
const orc_c32 *orc_acq_worker_code_fft(const orc_acq_worker *w);

/* search_satellite :158-226.
 *   tables[d] : D pointers to n-entry tables; table_freq[d] = DopplerShiftTable.doppler_freq_hz
 *   returns 1 (Some) / 0 (None).
 * Optional per-bin planes for parity tests (may be NULL): bin_max[d], bin_argmax[d], bin_sum[d]
 * (bin_sum = is_good_satellite's 8-lane sum of THAT bin's plane), *bins_done = bins visited.
 * no_early_exit != 0 keeps scanning after the first passing bin (result is still the first). */
int orc_search_satellite(orc_acq_worker *w, const orc_c32 *samples, size_t n_samples,
                         const orc_c32 *const *tables, const float *table_freq, size_t n_tables,
                         uint64_t local_tail, size_t num_integrations, orc_acq_result *out,
                         float *bin_max, uint32_t *bin_argmax, float *bin_sum, uint32_t *bins_done,
                         int no_early_exit);
/* is_good_satellite :229-238 ; also returns the 8-lane ordered sum through *sum_out */
int orc_is_good_satellite(const float *power, size_t n, float max_val, float *sum_out);
/* The same decision made from per-bin metrics only (what the GPU host replays). */
int orc_decide_from_metrics(const float *bin_max, const uint32_t *bin_argmax, const float *bin_sum,
                            const float *table_freq, size_t n_tables, size_t fft_size, uint8_t prn,
                            float fs, uint64_t local_tail, float threshold, orc_acq_result *out);

/* run()'s fan-out :302-313 with threads mirroring rayon: one worker per PRN bit set in mask.
 * found[p] (p = 0..31) <-> Option.  n_threads <= 0 -> 1.  cells_out = code phases x bins visited. */
int orc_acq_search_all(orc_acq_worker *const *workers, size_t n_workers, uint64_t mask,
                       const orc_c32 *samples, size_t n_samples, const orc_c32 *const *tables,
                       const float *table_freq, size_t n_tables, uint64_t local_tail,
                       size_t num_integrations, int n_threads, int no_early_exit,
                       orc_acq_result *results, uint8_t *found, uint64_t *cells_out);

/* AcquisitionManager :39-74.  mode: 0 Cold, 1 Warm, 2 Steady */
int orc_acq_mode_for(size_t tracked_count);
void orc_acq_pacing_and_list(int mode, uint32_t active_mask, uint64_t *interval_ms, uint32_t *mask);

/* ---------------- Tracking (src/tracking/do_tracking.rs) */
typedef struct { float tau1, tau2; } orc_loop_filter;             /* :52-71 */
orc_loop_filter orc_loop_filter_new(float noise_bw, float damping, float gain);
float orc_loop_filter_update(const orc_loop_filter *f, float d_err, float err, float dt);

enum { ORC_CODE_INDEX_FAITHFUL = 0, ORC_CODE_INDEX_FIXED = 1 };
enum { ORC_STATE_IDLE = 0, ORC_STATE_ACQUIRING = 1, ORC_STATE_TRACKING = 2, ORC_STATE_LOST = 3 };

typedef struct {            /* TrackingChannel :88-116 (fields that carry state) */
    uint8_t id, prn;
    int32_t state;          /* ORC_STATE_*; Tracking(prn) <-> state==TRACKING && state_prn==prn */
    uint8_t state_prn;
    uint32_t lost_counter;
    float fs;
    uint64_t next_sample_index;
    uint64_t num_samples_per_code;
    float carrier_freq, carrier_phase, carrier_error, carrier_nco;
    float code_phase, code_error, code_nco, code_rate;
    float i_prompt, q_prompt;
    orc_loop_filter pll_filter, dll_filter;
    int32_t code_index_mode; /* extra: ORC_CODE_INDEX_* (SURVEY §4 off-by-one switch) */
    /* generalisation for the constellations the reference does not implement (no reference code,
     * parity unpinned): 0 / NULL = the reference's GPS C/A E/P/L behaviour */
    int32_t n_arms;           /* 0 or 3: E/P/L; 5: + very early / very late */
    float el_space;           /* 0 -> EARLY_LATE_SPACE 0.5 */
    float vel_space;          /* 0 -> 1.0 */
    int32_t boc11;            /* chip x BOC(1,1) sub-carrier sign (+ first half chip, - second) */
    const int8_t *custom_codes; /* [n_codes][code_len] +-1 chips, row = prn-1 (FIXED) */
    uint32_t n_codes, code_len;
} orc_trk_channel;

void orc_trk_new(orc_trk_channel *c, uint8_t id, float fs);                 /* :118-146 */
void orc_trk_start(orc_trk_channel *c, const orc_acq_result *r);            /* :148-154 */
int orc_trk_is_active(const orc_trk_channel *c);                            /* :156-158 */
void orc_trk_reset(orc_trk_channel *c);                                     /* :311-327 */
/* get_ca_chip :274-277.  returns 0 and *chip, or -1 if the reference would index out of bounds */
int orc_trk_get_ca_chip(const orc_trk_channel *c, float phase, float *chip);
/* early_late_correlation :231-272.  data (n = num_samples_per_code) is mutated in place like the
 * reference.  out6 = (i_p,q_p,i_e,q_e,i_l,q_l).  If acc64 != NULL the same per-sample f32 products
 * are ALSO accumulated in double (accuracy yardstick for the GPU's tree sums). Returns 0 / -1 OOB. */
int orc_trk_early_late_correlation(orc_trk_channel *c, orc_c32 *data, float out6[6], double acc64[6]);
/* generalised form: out10 = (i_p,q_p,i_e,q_e,i_l,q_l,i_ve,q_ve,i_vl,q_vl); arms/BOC/custom code from the channel */
int orc_trk_early_late_correlation_ex(orc_trk_channel *c, orc_c32 *data, float out10[10], double acc64[10]);
void orc_trk_run_loop_filters(orc_trk_channel *c, float i_p, float q_p, float i_e, float q_e,
                              float i_l, float q_l);                        /* :279-302 */
/* do_work :183-210 on caller-supplied samples.  returns 0 none, 1 SatelliteLost (msg_prn as the
 * reference builds it, i.e. after reset -> 0), -1 OOB. */
int orc_trk_do_work(orc_trk_channel *c, orc_c32 *data, float out6[6], uint8_t *msg_prn);

/* ---------------- MulticastRingBuffer (src/utilities/multicast_ring_buffer.rs:36-130) */
typedef struct {
    orc_c32 *buffer;
    size_t buf_size, mask;
    uint64_t head;
} orc_ring;
int orc_ring_new(orc_ring *r, size_t buf_size);            /* :46-61, -1 if not a power of two */
void orc_ring_free(orc_ring *r);
void orc_ring_write_samples(orc_ring *r, const orc_c32 *s, size_t n);   /* :66-101 */
uint64_t orc_ring_get_head(const orc_ring *r);                          /* :103-105 */
void orc_ring_copy_to_slice(const orc_ring *r, uint64_t start, orc_c32 *dest, size_t n); /* :107-129 */
/* TrackingChannel::update :160-180 with the data buffer sized (deliberate deviation).
 * returns 0 None(no data / inactive), 1 processed, 2 processed + SatelliteLost, -1 OOB */
int orc_trk_update(orc_trk_channel *c, const orc_ring *ring, orc_c32 *scratch, float out6[6],
                   uint8_t *msg_prn);

/* the same generalised to the channel's code length / arms / BOC (no reference code, parity unpinned) */
int orc_trk_update_ex(orc_trk_channel *c, const orc_ring *ring, orc_c32 *scratch, float out10[10],
                      uint8_t *msg_prn);

/* teacher-forced variant for multi-epoch parity tests: sums computed from the channel's state -> computed10; decision,
 * loop filters and i/q_prompt driven by forced10 (see the .c) */
int orc_trk_update_forced(orc_trk_channel *c, const orc_ring *ring, orc_c32 *scratch, float computed10[10],
                          double computed64[10], const float forced10[10], uint8_t *msg_prn);

/* ---------------- fine Doppler (src/acquisition/acquisition_bk.rs:215-302, LEGACY file outside the reference's module
 * tree, no test) — SURVEY §8 f3.  PARITY UNPINNED.  Restated on the live path's data: `samples` are the c32 snapshot of
 * do_acquisition.rs:300 (the legacy took i16 pairs, :225-235), size_signal_use = (num_integrations-1)*N (:240), code
 * chips at floor((x*rate)/fs) % len (:241-247), mean removal (:236-237), fft_size = 8*next_pow2(size) (:249), first
 * index of the maximum |X| (:277-283).  freq_hz = (idx*fs)/fft_size for idx <= one_side (:250-253,298-299 without the
 * legacy's (-1)^is_complex factor); for idx > one_side the legacy indexes out of bounds (:285-288: it would panic), here
 * the negative frequency -((fft_size-idx)*fs)/fft_size is returned with *upper_half = 1. */
int orc_finer_doppler(const orc_c32 *samples, size_t n_samples, size_t code_phase, const int8_t *chips, size_t code_len,
                      float code_rate, float fs, size_t size_signal_use, uint64_t *peak_index, float *peak_mag,
                      float *freq_hz, int *upper_half, size_t *fft_size_out);

/* ---------------- bit sync + nav-bit accumulation (src/decoding.rs:8,40-227, LEGACY file outside the module tree; its
 * parity/subframe functions :229-352 panic as written) — SURVEY §8 f4.  PARITY UNPINNED (no reference test).
 * fixed = 0 restates the file line by line, bugs included: bits are only emitted when frame_sync_ind == 0 (:203-205 has
 * no modulo) and the preamble is tested only when exactly 8 bits have been collected (:131-135, the VecDeque keeps
 * growing); fixed = 1 wraps the bit boundary modulo 20 and slides an 8-bit window. */
typedef struct {
    int fixed;
    int flag_bit_sync, flag_frame_sync, sync_sw, loop_sw;
    uint64_t biti, frame_sync_ind, bit_code_cnt, sf_buffer_loc, sf_cnt, sf_start_biti, tow_expected_ind;
    uint64_t bit_sync_buff[20];
    float i_p;
    int8_t polarity;
    int8_t *frame_bits; size_t n_frame_bits, cap_frame_bits;
    int8_t buff_preamble[8]; size_t n_preamble;          /* len of the reference's VecDeque (keeps counting past 8) */
    int8_t last_bit;
} orc_nav_sync;
void orc_nav_sync_new(orc_nav_sync *s, int fixed);          /* NavSyncStatus::new :68-100 */
void orc_nav_sync_free(orc_nav_sync *s);
/* nav_decoding :102-162 up to frame sync (subframe decoding :147-160 panics in the reference: not restated).
 * returns 1 when this epoch completed a bit (sync_sw), else 0 */
int orc_nav_sync_update(orc_nav_sync *s, float old_i_prompt, float i_prompt, uint64_t cnt, uint64_t buff_loc);
/* the six parity equations of :259-346 on 32 symbols in +-1 form (bits[0..1] = D29*, D30* of the previous word,
 * bits[2..25] = d1..d24, bits[26..31] = D25..D30); returns 1 when every product equals its parity symbol
 * (the reference sums the differences, :348-350, so +2 and -2 cancel: *ref_sum_zero reports that form) */
int orc_nav_parity_check(const int8_t bits[32], int *ref_sum_zero);

/* ---------------- digital front-end (src/rf/frontend.rs, nco_lut.rs, dc_remove.rs) — SURVEY §8 f2.
 * The reference has no test for these files: PARITY UNPINNED beyond this line-by-line restatement. */
#define ORC_LUT_SIZE 2048                       /* nco_lut.rs:4 */
typedef struct {
    float lut_re[ORC_LUT_SIZE], lut_im[ORC_LUT_SIZE];   /* NcoLut :17-22 */
    float phase_accumulator, phase_step;
    float bias_re[8], bias_im[8], alpha, con;           /* DcRemoverSimd, dc_remove.rs:3-8 (8 SIMD lanes) */
} orc_frontend;
void orc_frontend_new(orc_frontend *fe, float f_if, float fs_in, float fs_out);   /* frontend.rs:19-30 */
/* DigitalFrontend::process_block frontend.rs:33-62: in place over interleaved I/Q f32; only whole chunks of 16 floats */
void orc_frontend_process_block(orc_frontend *fe, float *raw_floats, size_t n_floats);

/* TrackingManager::process_channels :351-371 (rayon par_iter_mut -> OpenMP), looped like run() :384-415 */
int64_t orc_trk_process_channels(orc_trk_channel *ch, int n_channels, const orc_ring *ring, orc_c32 *scratch,
                                 size_t scratch_stride, int max_passes, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
