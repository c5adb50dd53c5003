/*
 * gnss_mi355x.h — C ABI of the MI355X-native acquisition + tracking hot path.
 *
 * Drop-in boundary for the acquisition / tracking channel API of kewei/gnss-sdr-rs.  The reference
 * has no FFI for this path (its API is Rust: SURVEY.md §8b); every entry point below names the
 * reference item (file:line, relative to the reference checkout) it replaces, and INTEGRATION.md
 * shows the Rust `extern "C"` binding a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types; no exceptions cross the boundary
 *   - every function returns a gm_status (0 = GM_OK, < 0 = error); absence (`Option::None`) is
 *     reported through `found[]` flags, never through the status
 *   - handles are opaque and thread-compatible: one thread per handle at a time (the reference
 *     hands one `&mut` worker/channel to each rayon task)
 *   - host pointers unless the name ends in `_dev`; `_dev` pointers are HIP device pointers on the
 *     handle's device and the call is asynchronous on the handle's stream
 *   - STREAMS: every stream the library creates is NON-BLOCKING (hipStreamNonBlocking).  Work on the NULL stream (hipMemset,
 *     hipMemcpy of device memory, a launch without a stream) is NOT ordered against the library's work in either direction.  A
 *     device buffer handed to a `_dev` entry must be READY on the stream the handle works on: pass the producer's stream with
 *     gm_acq_set_stream / gm_trk_set_stream (the entry then runs behind the producer in stream order, and what it writes is ready
 *     on that stream), or synchronise the producer first.  gm_acq_prepare_dev takes the producer's stream per call
 *     (`ready_stream`).  The reference copies its 10 ms out of the ring and THEN searches them (do_acquisition.rs:297-313): the
 *     stream order is that sequence.  INTEGRATION.md §3.3; tests/test_gpu_acquisition.py::test_caller_stream_orders_...
 *   - arithmetic type: f32 everywhere, as in the reference (num_complex::Complex32 = {f32 re, im})
 *   - the compute path is HIP on gfx950 only; there is NO CPU fallback — without a usable device
 *     every compute entry returns GM_ERR_NO_DEVICE
 */
#ifndef GNSS_MI355X_H
#define GNSS_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GM_ABI_VERSION 7   /* 7: gm_trk_collect hands over the channel states, gm_trk_get_states / gm_trk_set_states,
                              gm_ring_get_enqueued_head (round 6); 6: gm_acq_prepare_dev returns a token (round 5) */

typedef enum {
    GM_OK = 0,
    GM_ERR_INVALID_ARG = -1,   /* null pointer, bad size, prn out of range ... (the reference panics) */
    GM_ERR_UNSUPPORTED_N = -2, /* no in-LDS FFT plan for this fft_size (see gm_fft_supported_sizes) */
    GM_ERR_NO_DEVICE = -3,     /* no HIP device / gm_init not called */
    GM_ERR_HIP = -4,           /* a HIP runtime call failed; gm_last_error() has the text */
    GM_ERR_OUT_OF_RANGE = -5,  /* slice/index out of range (the reference panics: index out of bounds) */
    GM_ERR_ALIGNMENT = -6,     /* fft_size % 8 != 0: the reference's SIMD tails (doppler_shift.rs:26,
                                  do_acquisition.rs:230-234) would leave stale / uncounted elements */
    GM_ERR_NOMEM = -7,
    GM_ERR_UNSUPPORTED = -8    /* optional component absent (RCCL for the gm_comm_* entries) */
} gm_status;

/* num_complex::Complex32 */
typedef struct { float re, im; } gm_c32;

/* ------------------------------------------------------------------ library / device */
int gm_abi_version(void);
/* Select the HIP device for subsequently created handles (one process per GPU). */
int gm_init(int device);
int gm_device_count(int *count);
const char *gm_last_error(void);
const char *gm_status_string(int status);

/* ------------------------------------------------------------------ constants / code table
 * GPS_CA_CODE_32_PRN (src/constants/gps_ca_constants.rs:1-1346): row r <-> PRN r+1, chips +-1. */
int gm_ca_code_row(int row, int8_t out_chips[1023]);
/* BeiDou B1I ranging code of PRN 1..37 (BDS-SIS-ICD-B1I: 11-stage Gold code, 2046 chips at 2.046 Mcps; +1 <-> logic 0).
 * Not in the reference (its README names BeiDou, its code has GPS only): provided for BASELINE configs[3]'s grid, to be
 * passed as gm_acq_cfg.codes.  n_chips = 2046 for the ICD's code; 2047 yields the untruncated period. */
int gm_b1i_code(uint32_t prn, int8_t *out_chips, uint32_t n_chips);
/* generate_ca_code_samples(prn, code_rate, f_sampling) (src/utilities/ca_code.rs:12-27).
 * *n_out = round(fs/(code_rate/1023)); writes min(n, cap) samples.  GM_ERR_OUT_OF_RANGE where the
 * reference would panic (prn not in 1..=32, chip index reaching 1023). */
int gm_generate_ca_code_samples(uint8_t prn, float code_rate, float f_sampling, int8_t *out, size_t cap,
                                size_t *n_out);

/* ------------------------------------------------------------------ Doppler wipe-off
 * DopplerShiftTable::new(f_if, doppler_freq_hz, fs, num_samples) (src/acquisition/doppler_shift.rs:10-22).
 * Host-side (glibc cosf/sinf, exactly the reference's arithmetic); *doppler_freq_hz_out receives the
 * stored field (= f_if + doppler, :20). */
int gm_doppler_table_new(float f_if, float doppler_freq_hz, float fs, size_t num_samples,
                         float *doppler_freq_hz_out, gm_c32 *table_out);
/* apply_doppler_shift(samples, table, output) (doppler_shift.rs:25-58) on the GPU.
 * Writes only the first 4*floor(n/4) outputs, like the reference. */
int gm_apply_doppler_shift(const gm_c32 *samples, const gm_c32 *table, gm_c32 *output, size_t n);

/* ------------------------------------------------------------------ FFT<T> / RealFFT<T> (src/fft.rs:5-56)
 * Unnormalised complex FFT, in place, `batch` contiguous transforms of length n.
 * dir: 0 forward (FFT::execute), 1 inverse.  n: one of gm_fft_supported_sizes() (one in-LDS transform), or ANY other
 * length up to 8192 (Bluestein on the smallest power-of-two plan >= 2n - 1; the reference's FFT<T> is generic over n). */
int gm_fft_c2c_f32(size_t n, int dir, gm_c32 *inout, size_t batch);
/* FFT::power_spectrum (src/fft.rs:27-29): transforms `inout` in place and writes |X|^2. */
int gm_fft_power_spectrum_f32(size_t n, gm_c32 *inout, float *power);
/* RealFFT::execute (src/fft.rs:45-49): n real inputs -> n/2+1 bins. */
int gm_rfft_f32(size_t n, const float *in, gm_c32 *out);
/* Sizes with an in-LDS plan; returns how many were written (<= cap). */
int gm_fft_supported_sizes(uint32_t *sizes, int cap);

/* ------------------------------------------------------------------ Acquisition
 * AcquisitionResult (src/acquisition/do_acquisition.rs:93-116) + the winning table index. */
typedef struct {
    uint8_t prn;
    uint64_t code_phase_samples;
    float code_phase_chips;
    float carrier_freq;        /* DopplerShiftTable.doppler_freq_hz of the winning bin = IF + Doppler */
    float fs;
    float mag_relative;        /* raw accumulated peak power (do_acquisition.rs:219) */
    uint64_t sample_global_index;
    int32_t doppler_bin;       /* extra: index into the table list */
} gm_acq_result;

typedef enum { GM_FMT_C32 = 0, GM_FMT_I8_IQ = 1, GM_FMT_I8_REAL = 2 } gm_sample_format;

typedef struct {
    float fs;                  /* freq_sampling_hz */
    float f_if;                /* used only when `tables` is NULL */
    uint32_t fft_size;         /* samples per code period (do_acquisition.rs:249-251): one of gm_fft_supported_sizes()
                                  (fused in-LDS kernels), or Q x {16000, 8000, 8192, 6000, 5000, 4000}, Q in {2,3,4,5,6,8}
                                  (e.g. 32000 for a 4 ms code at 8 Msps, 25000 for GPS at 25 Msps: composite path,
                                  transforms decimated in time, no intermediate plane in HBM) */
    uint32_t n_integrations;   /* LONG_SAMPLES_LENGTH = 10 (:23) */
    uint32_t n_bins;           /* Doppler bins; reference: 14000/500+1 = 29 (:248) */
    const float *doppler_hz;   /* [n_bins] offsets from f_if, ascending as the reference iterates */
    const gm_c32 *tables;      /* optional [n_bins][fft_size] caller-built DopplerShiftTable.table */
    const float *table_freq;   /* optional [n_bins] DopplerShiftTable.doppler_freq_hz (with `tables`) */
    uint32_t n_prn;            /* workers; reference: PRN_SEARCH_ACQUISITION_TOTAL = 32 (:22) */
    const uint8_t *prn_ids;    /* [n_prn] PRN of each worker (1..=32 when `codes` is NULL) */
    const int8_t *codes;       /* optional [n_prn][code_len] +-1 chips for non-GPS code families */
    uint32_t code_len;         /* chips per period of `codes` (ignored when NULL: 1023) */
    float code_rate;           /* chips/s of `codes` (ignored when NULL: 1.023e6) */
    float threshold;           /* is_good_satellite ratio, 7.0 (:237); 0 -> 7.0 */
    int32_t decision_mode;     /* GM_DECIDE_REFERENCE (0): first ascending bin whose running best passes the ratio test
                                  (the reference's early exit, :211-222).  GM_DECIDE_BEST_BIN (1): strongest bin of the
                                  whole grid, reported if IT passes the ratio test — not the reference's behaviour; for
                                  callers that hand the carrier to a PLL (a strong signal passes the test 1-2 kHz early) */
    int32_t strict_sum_order;  /* 0: the plane sum of is_good_satellite (:229-235) is a per-lane + shuffle-tree sum (within
                                  ~2e-6 of the reference's, FFT rounding aside).  1: summed in the reference's own order —
                                  eight running f32 sums over chunks_exact(8), then reduce_sum from -0.0 — and the
                                  integrations accumulated strictly in sequence (no grid-tail split); costs ~4 us per
                                  (worker, bin) workgroup.  Composite sizes (Q x a base plan) as well since ABI 6: their kernels
                                  then also store the accumulated power planes (n_prn * n_bins * fft_size * 4 bytes of device
                                  memory) and a second kernel sums each plane in that order. */
    int32_t reference_products; /* 0: `result_buf[i] *= conj(code_fft[i])` (:184-186) and `norm_sqr()` (:190-192) use two fused
                                  multiply-adds each (one rounding fewer per component; 4 instead of 6 instructions per
                                  element).  1: formed as num-complex forms them — every product and every sum rounded on its
                                  own (re = a.re*b.re - a.im*b.im, im = a.re*b.im + a.im*b.re; re*re + im*im) — so that the only
                                  arithmetic on the path that differs from the reference's is the FFT itself (rustfft's plan
                                  cannot be restated here).  ~3 % slower.  In-LDS sizes only.  (ABI 5) */
} gm_acq_cfg;
typedef enum { GM_DECIDE_REFERENCE = 0, GM_DECIDE_BEST_BIN = 1 } gm_decision_mode;

typedef struct gm_acq gm_acq;

/* = building the Doppler tables (:252-262) + AcquisitionWorker::new for every PRN (:268-271):
 * code replicas resampled, their forward FFTs computed on the GPU, plans/twiddles uploaded. */
int gm_acq_create(const gm_acq_cfg *cfg, gm_acq **out);
int gm_acq_destroy(gm_acq *a);

/* The batched equivalent of `workers.par_iter_mut()...search_satellite(...)` (:302-313, :158-226).
 *   samples  : n_integrations*fft_size samples in `fmt`
 *   prn_mask : bit i set <-> worker i searched (the (mask >> (prn-1)) & 1 test for the default list)
 *   results[i], found[i] for every worker i (found = 0 <-> None).
 * Identical outcome to the reference loop: ascending-Doppler running best, first bin passing
 * is_good_satellite wins (early exit), argmax = first strict maximum. */
int gm_acq_search(gm_acq *a, const void *samples, size_t n_samples, int fmt, uint64_t local_tail,
                  uint64_t prn_mask, gm_acq_result *results, uint8_t *found);
int gm_acq_search_c32(gm_acq *a, const gm_c32 *samples, size_t n_samples, uint64_t local_tail,
                      uint64_t prn_mask, gm_acq_result *results, uint8_t *found);
int gm_acq_search_i8(gm_acq *a, const int8_t *iq_interleaved, size_t n_samples, uint64_t local_tail,
                     uint64_t prn_mask, gm_acq_result *results, uint8_t *found);

/* run()'s snapshot + fan-out (do_acquisition.rs:297-313) against the device ring mirror: searches the
 * n_integrations*fft_size samples ending at the ring's head (device-to-device, wrap-aware); *local_tail_out =
 * head - M*N.  GM_ERR_OUT_OF_RANGE while head < M*N (the reference skips the round, :299). */
typedef struct gm_ring gm_ring;
int gm_acq_search_ring(gm_acq *a, gm_ring *ring, uint64_t prn_mask, gm_acq_result *results, uint8_t *found,
                       uint64_t *local_tail_out);

/* Fine-Doppler refinement after detection (finer_doppler, src/acquisition/acquisition_bk.rs:215-302 —
 * a legacy file outside the reference's module tree): for every found[p], the snapshot of the LAST search on this handle
 * (still in HBM) is code-stripped from results[p].code_phase_samples over (num_integrations-1)*fft_size samples (:240-272),
 * mean-removed (:236-237), zero-padded to 8*next_pow2 (:249) and transformed; the first index of the maximum |X| (:276-283)
 * gives fine_freq_hz[p] = (idx*fs)/fft_size (:251-253), i.e. IF + Doppler to fs/fft_size (7.6 Hz at 8 Msps, 10 ms).
 * Indices above fft_size/2 are reported as negative frequencies (the legacy indexes out of bounds there, :285-288, and
 * multiplies by (-1)^is_complex, :298-299: neither is reproduced).  Entries of not-found PRNs are left untouched.
 * Any output pointer may be NULL.  Synchronous.  GM_ERR_UNSUPPORTED_N if the long FFT does not factor into two in-LDS plans. */
int gm_acq_finer_doppler(gm_acq *a, const gm_acq_result *results, const uint8_t *found, uint32_t n_prn,
                         float *fine_freq_hz, uint64_t *peak_index, float *peak_mag, uint64_t *fft_size);

/* Device-resident form: samples already in HBM; kernels are enqueued on the handle's stream and the
 * call returns without synchronising.  d_metrics (optional, may be NULL -> internal buffer) receives
 * 3*n_prn*n_bins 32-bit words: max f32 [P][D], argmax u32 [P][D], sum f32 [P][D]. */
int gm_acq_search_dev(gm_acq *a, const void *d_samples, int fmt, void *d_metrics);
/* Which workers the device-resident form searches (bit i <-> worker i); default: all. */
int gm_acq_set_prn_mask(gm_acq *a, uint64_t prn_mask);
/* ---- multi-GPU exchange (nothing distributed exists in the reference: do_acquisition.rs:302-313
 * fans the PRNs out over rayon threads of one host).  One process per GPU; PRNs are sharded in contiguous blocks of
 * n_prn per rank.  gm_comm_get_unique_id on rank 0, the 128 bytes travel out of band (file, socket, MPI, the host
 * application's own channel), gm_comm_init on every rank after gm_init(device).  RCCL (librccl.so.1) is bound at the
 * first call; GM_ERR_UNSUPPORTED if it is not installed. */
#define GM_COMM_ID_BYTES 128
typedef struct gm_comm gm_comm;
int gm_comm_get_unique_id(uint8_t id[GM_COMM_ID_BYTES]);
int gm_comm_init(int nranks, int rank, const uint8_t id[GM_COMM_ID_BYTES], gm_comm **out);
int gm_comm_destroy(gm_comm *c);
int gm_comm_info(gm_comm *c, int *nranks, int *rank);
/* The path's one exchange step, enqueued on the handle's stream (no host synchronisation): all-gather this rank's
 * metrics block d_local ([3][P][D] words as written by gm_acq_search_dev; NULL -> the handle's internal block) from
 * every rank and regroup into d_all = [3][nranks*P][D] (rank-major worker order), the layout gm_acq_decide_dev
 * takes with n_prn = nranks*P.  Every rank obtains the same block, so every rank's decision is identical. */
int gm_acq_allgather_metrics(gm_acq *a, gm_comm *c, const void *d_local, void *d_all);
/* Overlapped form: the all-gather + regroup are ordered behind everything enqueued so far on the handle's stream but run
 * on the communicator's own stream, so the next dwell's kernels need not wait for the collective; gm_comm_wait makes
 * hip_stream wait (on the device, no host synchronisation) for the last such exchange before d_all is consumed.  The
 * two buffers must not be reused before that wait. */
int gm_acq_allgather_metrics_async(gm_acq *a, gm_comm *c, const void *d_local, void *d_all);
int gm_comm_wait(gm_comm *c, void *hip_stream);
/* The same exchange for a grid that mixes transform sizes (BASELINE configs[3]: GPS + Galileo-E1 geometry + BeiDou B1I
 * codes in one family-major list; a rank's contiguous block may span two families, i.e. two gm_acq handles): all-gather
 * of `words` 32-bit words per rank, d_all = [nranks][words], enqueued on hip_stream (NULL: the default stream). */
int gm_comm_allgather_words(gm_comm *c, const void *d_local, void *d_all, size_t words, void *hip_stream);
/* Gathered padded blocks [nranks][3][p_max][n_bins] -> ONE family-major grid d_out = [3][n_rows][n_bins]:
 * d_row_map[i] = rank * p_max + row of the block that holds code i of the family-major list (device, [n_rows]).
 * Replaces the per-PRN fan-in of do_acquisition.rs:302-313 for the sharded grid; pure data movement. */
int gm_grid_assemble_dev(const void *d_gathered, uint32_t nranks, uint32_t p_max, uint32_t n_bins,
                         const uint32_t *d_row_map, uint32_t n_rows, void *d_out, void *hip_stream);
/* gm_acq_decide_dev without a handle: the reference's decision (do_acquisition.rs:195-238) for `n_prn` codes of ONE
 * family from device-resident planes (each [n_prn][n_bins]; e.g. a family's rows inside gm_grid_assemble_dev's output),
 * d_prn_ids / d_table_freq device arrays, results into caller-owned device arrays.  decision_mode: gm_decision_mode.
 * Asynchronous on hip_stream. */
int gm_acq_decide_planes_dev(const float *d_max, const uint32_t *d_argmax, const float *d_sum, uint32_t n_prn,
                             uint32_t n_bins, const uint8_t *d_prn_ids, const float *d_table_freq, uint32_t fft_size,
                             float fs, float code_rate, float threshold, int decision_mode, uint64_t local_tail,
                             gm_acq_result *d_results, uint8_t *d_found, void *hip_stream);

/* Replay the reference's decision (running best + ratio test + early exit) on the GPU from a metrics
 * block laid out as above for `n_prn` workers (e.g. an all-gathered one).  prn_ids: host [n_prn].
 * Asynchronous; results land in an internal device buffer read back by gm_acq_fetch_results. */
int gm_acq_decide_dev(gm_acq *a, const void *d_metrics, uint32_t n_prn, const uint8_t *prn_ids,
                      uint64_t local_tail);
int gm_acq_fetch_results(gm_acq *a, uint32_t n_prn, gm_acq_result *results, uint8_t *found); /* syncs */
/* The same decision replay on host-resident metrics ([n_prn][n_bins] planes), no device involved. */
int gm_acq_decide_host(const float *max, const uint32_t *argmax, const float *sum, const float *table_freq,
                       uint32_t n_prn, uint32_t n_bins, const uint8_t *prn_ids, uint32_t fft_size, float fs,
                       float code_rate, float threshold, uint64_t local_tail, gm_acq_result *results,
                       uint8_t *found);
int gm_acq_synchronize(gm_acq *a);
/* Back-to-back dwells (a receiver that searches dwell after dwell): with `on`, a gm_acq_decide_dev on the metrics block the
 * LAST gm_acq_search_dev wrote is not launched on its own but kept, and runs inside the first kernel of the NEXT
 * gm_acq_search_dev (extra workgroups beside the forward transforms, one launch and one kernel boundary fewer per dwell), or
 * at the next gm_acq_synchronize / gm_acq_fetch_results / gm_acq_decide_dev / gm_acq_set_stream, whichever comes first.
 * Until then that metrics block must stay as the search left it; results are the same either way.  Off by default
 * (in-LDS transform sizes with n_bins <= 64 only; other handles accept the call and decide at once as before). */
int gm_acq_set_deferred_decision(gm_acq *a, int on);
/* Stage F (carrier mix + forward transforms) of the NEXT dwell ahead of time: runs on a stream of the handle's own into a second
 * spectrum buffer, beside whatever the handle's stream is doing, as soon as that buffer is free (the stage C that last read it
 * has ended).  The preparation is a SNAPSHOT of d_samples — the counterpart of the reference copying its 10 ms out of the ring
 * before it searches them (do_acquisition.rs:297-301) — and is named by the generation number written to *token (never 0), NOT by
 * the address: gm_acq_search_prepared_dev(a, token, d_metrics) launches stage C on those spectra, and nothing else ever uses
 * them.  gm_acq_search_dev always transforms the samples its own argument holds at that moment, whatever was prepared from the
 * same address before (ABI 5 matched a following search by pointer and format: a caller that refilled the buffer in between —
 * any ring-backed receiver — silently got the old samples' spectra).
 *   ready_stream: a HIP stream (or NULL).  Non-NULL: the samples are complete once the work queued on that stream SO FAR has run
 *     (an asynchronous copy, a front-end kernel): the library records an event there and stage F waits for it.  NULL: d_samples
 *     already holds the samples when the call is made (host-time contract, e.g. a published part of a device ring).
 *   Either way d_samples must stay unchanged until stage F has read it: until the search that consumes the token has been
 *     synchronised, or gm_acq_synchronize after gm_acq_drop_prepared.
 *   One preparation is outstanding at a time: a second gm_acq_prepare_dev replaces the first (its token becomes stale),
 *     gm_acq_drop_prepared forgets it; a stale / consumed / unknown token makes gm_acq_search_prepared_dev return
 *     GM_ERR_INVALID_ARG and launch nothing.  Plain searches in between leave the preparation intact.
 * Call order for dwell after dwell: search_prepared(k), prepare(k + 1), decide(k).  Pays where stage C leaves CUs idle in its last
 * round — N = 16368: 32 PRN x 29 bins are 3.6 rounds of one workgroup per CU and stage F fits into the rest; not at N = 8000, whose
 * last round is already filled.  Same metric words as the plain search.  The first call allocates the second spectrum buffer
 * (n_bins * n_integrations * fft_size * 8 bytes), the stream and three events — all of them or, on failure, none;
 * gm_acq_destroy releases them.  Composite sizes prepare nothing: they issue a token all the same and run the whole search at
 * gm_acq_search_prepared_dev from d_samples as it is THEN — behind the event recorded on `ready_stream` at prepare time (ABI 7: the
 * ordering promise holds on every size). */
int gm_acq_prepare_dev(gm_acq *a, const void *d_samples, int fmt, void *ready_stream, uint64_t *token);
int gm_acq_search_prepared_dev(gm_acq *a, uint64_t token, void *d_metrics);
int gm_acq_drop_prepared(gm_acq *a);
/* Use an existing HIP stream (e.g. torch's current stream, or the stream that fills the sample buffer) instead of the handle's
 * own: every later entry of the handle is enqueued there, i.e. BEHIND what the caller has queued on it and in front of what the
 * caller queues next — the way to order a `_dev` entry against the producer of its samples and the consumer of its metrics without
 * a synchronisation (see STREAMS at the top).  The caller keeps the stream alive until the handle is destroyed or given another. */
int gm_acq_set_stream(gm_acq *a, void *hip_stream);

/* Per-(worker, bin) planes of the last search: max, first-argmax, sum of the accumulated power
 * plane ([n_prn][n_bins] each; any pointer may be NULL).  For parity tests and the all-gather. */
int gm_acq_metrics(gm_acq *a, float *max, uint32_t *argmax, float *sum);
/* ca_code_samples_fft of worker i (AcquisitionWorker field, :126): fft_size bins. */
int gm_acq_code_fft(gm_acq *a, uint32_t worker, gm_c32 *out);
/* The table list in use: [n_bins][fft_size] and [n_bins] (either may be NULL). */
int gm_acq_tables(gm_acq *a, gm_c32 *tables, float *table_freq);
/* Kernel timing of the last gm_acq_search*_dev call, measured with HIP events on the handle's
 * stream: ms_mix_fft (stage F), ms_corr (stage C, the dominant kernel), ms_decide.  Enable first: on = 1 times every
 * search, on = k > 1 every k-th (an event record costs about 2 us of stream time; four per timed search), 0 disables. */
int gm_acq_enable_timing(gm_acq *a, int on);
int gm_acq_last_timing(gm_acq *a, float *ms_mix_fft, float *ms_corr, float *ms_decide);
/* Averages over every gm_acq_search_dev call since timing was enabled (the last 512 at most). */
int gm_acq_timing_summary(gm_acq *a, uint32_t *launches, float *avg_ms_mix_fft, float *avg_ms_corr);

/* Diagnostic (not in the reference): out == NULL arms, then out = [n_integrations][8][8] int64 shader-clock stamps
 * of workgroup 0's waves at the phase boundaries of each transform of the last search.  The stamped kernels are compiled into a
 * DIAGNOSTIC build of the library only (-DGM_DIAG_STAMPS): the product library returns GM_ERR_UNSUPPORTED and launches nothing. */
int gm_acq_debug_stamps(gm_acq *a, long long *out);

/* AcquisitionManager (do_acquisition.rs:39-74): mode 0 ColdStart / 1 WarmStart / 2 SteadyState. */
int gm_acq_manager_mode_for(size_t tracked_count);
int gm_acq_manager_pacing_and_list(int mode, uint32_t active_prn_mask, uint64_t *interval_ms, uint32_t *mask);

/* ------------------------------------------------------------------ MulticastRingBuffer device mirror
 * (src/utilities/multicast_ring_buffer.rs:36-130): power-of-two ring of Complex32 addressed by the
 * absolute sample index `head`; the mirror keeps the same bytes in HBM for the tracking kernels. */
int gm_ring_create(size_t buf_size, gm_ring **out);                      /* ::new :46-61 */
int gm_ring_destroy(gm_ring *r);
int gm_ring_write_samples(gm_ring *r, const gm_c32 *samples, size_t n);  /* :66-101 */
int gm_ring_get_head(gm_ring *r, uint64_t *head);                        /* :103-105 */
int gm_ring_copy_to_slice(gm_ring *r, uint64_t start, gm_c32 *dest, size_t n); /* :107-129 */
/* write_samples that does not block the producer on the H2D copy: pinned staging + the ring's own copy stream; `head`
 * advances only after the samples have landed in HBM.  gm_ring_flush waits for all of it.  The head is published by PULL
 * (ABI 6): gm_ring_get_head, gm_ring_wait_head and every stage entry that snapshots the ring first retire the blocks whose copies
 * (and front-end kernels) have completed; nothing runs on the copy stream but copies and kernels (a host callback there — round 4 —
 * held the stream, i.e. the next block, until the runtime's callback thread woke up: milliseconds, now and then). */
int gm_ring_write_samples_async(gm_ring *r, const gm_c32 *samples, size_t n);
int gm_ring_flush(gm_ring *r);
/* What the asynchronous writer has ENQUEUED so far (>= gm_ring_get_head, which counts what has landed): the head that
 * gm_trk_update_all_async's passes are gated on, i.e. what a stage driver that never waits for the copies plans its pass count
 * from.  Equal to the head for a ring that is written synchronously.  ABI 7. */
int gm_ring_get_enqueued_head(gm_ring *r, uint64_t *head);
/* The notifier/Condvar of the reference ring (:42-43, :94-98) as used by do_tracking::run (do_tracking.rs:392-406):
 * sleep until head >= required_idx (wrapping signed comparison) or timeout_ms elapsed; *reached = 1 / 0. */
int gm_ring_wait_head(gm_ring *r, uint64_t required_idx, uint32_t timeout_ms, int *reached);

/* ------------------------------------------------------------------ Digital front-end
 * rf::frontend::DigitalFrontend (src/rf/frontend.rs:6-62): DC removal (DcRemoverSimd, src/rf/dc_remove.rs:10-29:
 * eight one-pole IIR lanes per component, alpha = 0.001) + LUT NCO down-mix (NcoLut, src/rf/nco_lut.rs:17-42: 2048
 * entries, f32 phase accumulator `% 2048`) + mix_simd (:8-15), bit-exact with the reference's f32 evaluation order.
 * Only whole chunks of 8 samples are processed (chunks_exact_mut(16), :35); a tail passes through unprocessed. */
typedef struct gm_frontend gm_frontend;
int gm_frontend_create(float f_if, float fs_in, float fs_out, gm_frontend **out);     /* ::new :19-30 */
int gm_frontend_destroy(gm_frontend *f);
/* NcoLut tables and phase_step (nco_lut.rs:25-34); any pointer may be NULL */
int gm_frontend_lut(gm_frontend *f, float lut_re[2048], float lut_im[2048], float *phase_step);
int gm_frontend_get_state(gm_frontend *f, float *phase_accumulator, float bias_re[8], float bias_im[8]);
int gm_frontend_set_state(gm_frontend *f, float phase_accumulator, const float bias_re[8], const float bias_im[8]);
/* process_block(&mut [f32]) :33-62 — host buffer of interleaved I/Q, in place (H2D + kernel + D2H, synchronous) */
int gm_frontend_process_block(gm_frontend *f, float *raw_floats, size_t n_floats);
/* device-resident form: d_in (GM_FMT_C32 or GM_FMT_I8_IQ) -> d_out (c32; may alias d_in for c32), asynchronous on
 * `stream` (a hipStream_t, NULL -> the handle's own stream; gm_frontend_synchronize waits for that one) */
int gm_frontend_process_dev(gm_frontend *f, const void *d_in, int fmt, void *d_out, size_t n_samples, void *stream);
/* n_streams independent streams (antennas, bands) in one launch, one workgroup each: fes[i] processes d_in[i] -> d_out[i]
 * (n_samples each); every front-end keeps its own state and NCO step; a handle may appear once per call.
 * Asynchronous on `stream` (NULL -> fes[0]'s own). */
int gm_frontend_process_dev_batch(gm_frontend *const *fes, uint32_t n_streams, const void *const *d_in, int fmt,
                                  void *const *d_out, size_t n_samples, void *stream);
int gm_frontend_synchronize(gm_frontend *f);
/* rf_thread's block step (src/rf/rf_thread.rs:43-48: process_block, then shared_ring_buffer.write_samples) fused and
 * non-blocking: host samples (c32, or int8 IQ: 2 B/sample over PCIe, converted on the GPU) -> pinned staging -> front-end
 * kernel writing straight into the ring mirror; head advances when the block is in HBM (gm_ring_flush to wait).  Blocks of up
 * to 2^19 samples per copy + launch; the kernels run on a stream of the ring's own behind the copies (an event per staging slot). */
int gm_frontend_write_ring(gm_frontend *f, gm_ring *ring, const void *samples, size_t n_samples, int fmt);
/* Long blocks (>= 48 pipeline segments of 3840 samples, output not aliasing the input) run in the SPECULATIVE form since round 6: the
 * block is cut into up to 32 runs on as many workgroups, each starting from a GUESSED DC-remover state (the recurrence in exact arithmetic over the
 * 16 384 steps before its warm-up) that an 8 640-step warm-up lets fall onto the true f32 chain; a second kernel verifies run by run
 * that the state a run entered with is bit for bit the state its predecessor left, and does a run again — sequentially, from the right
 * state — where it is not.  The results are those of the sequential front-end, word for word, whatever the guesses were
 * (tests/test_gpu_frontend.py::test_speculative_blocks_are_exact spoils every one of them); 0.52 -> 0.15 ms per 2^19-sample block.
 * Diagnostic (not in the reference): *runs = how many runs the verification has had to repeat on this handle so far. */
int gm_frontend_debug_repairs(gm_frontend *f, uint32_t *runs);

/* ------------------------------------------------------------------ Tracking
 * The evolving fields of TrackingChannel (src/tracking/do_tracking.rs:88-116). */
typedef struct {
    uint8_t prn;
    uint8_t active;            /* state == ChannelState::Tracking(prn) */
    uint8_t reserved[2];
    uint32_t lost_counter;
    uint64_t next_sample_index;
    uint64_t num_samples_per_code;
    float carrier_freq, carrier_phase, carrier_error, carrier_nco;
    float code_phase, code_error, code_nco, code_rate;
    float i_prompt, q_prompt;
} gm_trk_state;

/* Correlator outputs of one epoch: (i_p,q_p,i_e,q_e,i_l,q_l) = early_late_correlation() :231-272,
 * plus very-early / very-late for 5-arm configurations. */
typedef struct {
    float ip, qp, ie, qe, il, ql, ive, qve, ivl, qvl;
} gm_trk_out;

typedef enum { GM_CODE_INDEX_FAITHFUL = 0, GM_CODE_INDEX_FIXED = 1 } gm_code_index_mode;

typedef struct {
    float fs;
    uint32_t n_channels;       /* NUM_OF_CHANNELS = 15 (:18) */
    uint32_t n_arms;           /* 3 (E/P/L, reference) or 5 (VE/E/P/L/VL) */
    float early_late_space;    /* EARLY_LATE_SPACE = 0.5 chips (:28); 0 -> 0.5 */
    float very_early_late_space; /* 5-arm only; 0 -> 1.0 */
    int32_t code_index_mode;   /* FAITHFUL: get_ca_chip indexes row `prn` and saturates negative phases to
                                  chip 0 exactly like :274-277; FIXED: row prn-1, wrapping */
    int32_t boc11;             /* 1: multiply the chip by the BOC(1,1) sub-carrier sign (no reference code) */
    const int8_t *codes;       /* optional [n_codes][code_len] custom +-1 chips (NULL: GPS C/A table) */
    uint32_t n_codes, code_len;
    float nominal_code_rate;   /* 0 -> 1.023e6 */
    /* loop constants (:16-27); 0 -> reference value */
    float pll_bw, pll_zeta, pll_gain, dll_bw, dll_zeta, dll_gain, pll_dt, dll_dt;
    float lock_threshold;      /* LOCK_THRESHOLD = 15 (:16) */
    uint32_t max_lost_epochs;  /* MAX_LOST_EPOCHS = 20 (:17) */
    int32_t strict_libm;       /* 1: the carrier's cos / sin (`phase.cos()`, `phase.sin()`, :234-235) are glibc 2.35's cosf / sinf
                                  restated on the device, bit for bit (csrc/gm_libm.h sincosf_glibc: f64 reduction + polynomial,
                                  one rounding) — every sample's products then equal the reference host's; 0 (default): the
                                  device's own < 1 ulp forms, which differ from glibc's in the last bit on a quarter of the
                                  samples.  ABI version 4. */
    int32_t strict_sum_order;  /* 1: the correlator sums are added sample by sample in f32, the reference's own order
                                  (`i_p += re * p_chip`, :256-262) — one serial wave per channel instead of the persistent
                                  kernel's tree (csrc/trk_kernels.hip trk_serial_sum_kernel; ~40x its time per epoch).  With
                                  strict_libm as well, every correlator sum and every word of the channel state equal the
                                  reference's bit for bit, free-running.  0 (default): tree sums, within 1e-5 of the
                                  envelope and closer to the exact sum.  ABI version 4. */
    int32_t share_device;      /* 1: a receiver — the persistent tracking kernel takes at most a QUARTER of the device's resident
                                  workgroup places (workgroups per channel chosen accordingly), so that the other stages' kernels
                                  (the digital front-end writing the ring needs a whole CU per stream, an acquisition dwell every
                                  CU it can get) run BESIDE a tracking launch instead of queueing behind its all-resident grid.
                                  0 (default): every place, the fastest epoch (a tracking-only load).  The split of a code period
                                  over a channel's workgroups follows the count: sums differ in the last bits between the two
                                  settings, inside the default mode's tolerance.  ABI version 6. */
} gm_trk_cfg;

typedef struct gm_trk gm_trk;

int gm_trk_create(const gm_trk_cfg *cfg, gm_trk **out);   /* TrackingManager::new :336-348 + Channel::new :118-146 */
int gm_trk_destroy(gm_trk *t);
/* TrackingChannel::start :148-154.  FIXED mode starts code_phase at 0 (sample_global_index is already the code
 * start) and restores the nominal code_rate after a reset(); FAITHFUL copies code_phase_chips like the reference. */
int gm_trk_start(gm_trk *t, uint32_t ch, const gm_acq_result *r);
int gm_trk_reset(gm_trk *t, uint32_t ch);                              /* ::reset :311-327 */
int gm_trk_get_state(gm_trk *t, uint32_t ch, gm_trk_state *out);
int gm_trk_set_state(gm_trk *t, uint32_t ch, const gm_trk_state *in);
/* All n_channels records in ONE synchronisation + ONE copy (ABI 7) — what a manager that mirrors the channels' pub fields
 * (TrackingManager.channels, do_tracking.rs:329-333) calls once per pass instead of 2 x n_channels single-channel calls.
 * which: NULL = every channel, else [n_channels] flags — only flagged channels are written (the others keep the device's words). */
int gm_trk_get_states(gm_trk *t, gm_trk_state *out);
int gm_trk_set_states(gm_trk *t, const gm_trk_state *in, const uint8_t *which);
/* get_ca_chip(phase) :274-277 for channel ch (host-side table look-up with the configured mode). */
int gm_trk_get_ca_chip(gm_trk *t, uint32_t ch, float phase, float *chip);
/* LoopFilter::new / ::update (:52-71), host-side scalars. */
int gm_loop_filter_new(float noise_bw, float damping, float gain, float *tau1, float *tau2);
float gm_loop_filter_update(float tau1, float tau2, float d_err, float err, float dt);

/* early_late_correlation() :231-272 for one channel on caller-supplied samples
 * (n must equal the channel's num_samples_per_code): advances carrier_phase / code_phase and sets
 * i_prompt/q_prompt exactly like the reference; no loop filters, no lock logic. */
int gm_trk_correlate(gm_trk *t, uint32_t ch, const gm_c32 *samples, size_t n, gm_trk_out *out);
/* do_work() :183-210 on caller-supplied samples (the reference's real-data test path, :734-739).
 * *lost = 1 <-> Some(TrackingMessage::SatelliteLost); *lost_prn = the prn the reference puts in the
 * message (0, because reset() runs first, :199-201). */
int gm_trk_do_work(gm_trk *t, uint32_t ch, const gm_c32 *samples, size_t n, gm_trk_out *out, uint8_t *lost,
                   uint8_t *lost_prn);
/* The batched equivalent of TrackingManager::process_channels' par_iter over active channels
 * (:364-371) repeated while data is available: up to `max_epochs` passes; in each pass every active
 * channel with head >= next_sample_index + num_samples_per_code runs update() :160-180 against the
 * device ring.  outs: [max_epochs][n_channels] (may be NULL); processed/lost: [max_epochs][n_channels]
 * flags (may be NULL).  *epochs_done = passes in which at least one channel ran. */
int gm_trk_update_all(gm_trk *t, gm_ring *ring, uint32_t max_epochs, gm_trk_out *outs, uint8_t *processed,
                      uint8_t *lost, uint32_t *epochs_done);
/* Asynchronous device-resident form for benchmarking: enqueues `epochs` passes, no host readback. */
int gm_trk_update_all_dev(gm_trk *t, gm_ring *ring, uint32_t epochs);
/* gm_trk_update_all without a host wait per block (ABI 6) — for a receiver loop that feeds block after block.  The reference's
 * tracking thread sleeps on the ring's Condvar until the head has passed what it needs (do_tracking.rs:392-406); here that wait
 * happens ON THE DEVICE: the passes are ordered behind everything the ring's asynchronous writer (gm_ring_write_samples_async /
 * gm_frontend_write_ring) has ENQUEUED so far, by an event on the ring's copy stream, and their data gate uses that enqueued
 * head — the host neither waits for the samples to land nor for the passes to run.  Results ([outs | processed | lost] as in
 * gm_trk_update_all) land in one of 8 pinned slots; *ticket (never 0) names the call.
 * gm_trk_collect(ticket, wait, ...): wait = 0 -> *ready = 0 and nothing else when the call has not finished; otherwise the results
 * are handed over (any of outs / processed / lost / states / epochs_done may be NULL), *ready = 1 and the slot is free again.
 * states (ABI 7): [n_channels] records as they stood when THIS call's passes had run (a snapshot taken on the device in stream
 * order, whatever has been enqueued behind it) — the pub fields a manager shows for that moment, without a synchronisation.  Tickets are
 * collected in any order; a call is refused (GM_ERR_OUT_OF_RANGE, nothing launched) while the ticket issued eight calls earlier —
 * whose slot it would take — has not been collected.  A collect that returns an ERROR (a HIP failure of the wait, an exchange
 * time-out reported by the kernel) has CONSUMED the ticket: its slot is free, its results are gone, a second collect of it is
 * GM_ERR_INVALID_ARG; wrappers drop their record of a ticket whenever the library returns an error for it.
 * Same channel states and sums as the synchronous entry, bit for bit (tests/test_gpu_pipeline.py). */
int gm_trk_update_all_async(gm_trk *t, gm_ring *ring, uint32_t max_epochs, uint64_t *ticket);
int gm_trk_collect(gm_trk *t, uint64_t ticket, int wait, gm_trk_out *outs, uint8_t *processed, uint8_t *lost, gm_trk_state *states,
                   uint32_t *epochs_done, int *ready);
int gm_trk_synchronize(gm_trk *t);
/* The handle's own stream is created at the device's HIGHEST priority: the tracking loop is the receiver's latency path (one short
 * launch per block of samples) and must not queue behind a front-end block or an acquisition dwell in flight on another stream.
 * gm_trk_set_stream replaces it with the caller's (whose priority is then the caller's choice). */
int gm_trk_set_stream(gm_trk *t, void *hip_stream);
/* Diagnostic (not in the reference): call with out == NULL to arm `cap` epochs of per-phase shader-clock stamps
 * of workgroup 0 in the persistent kernel, then with out = [cap][48] int64 after a launch to read them. */
int gm_trk_debug_stamps(gm_trk *t, uint32_t cap, long long *out);
int gm_trk_enable_timing(gm_trk *t, int on);
int gm_trk_last_timing(gm_trk *t, float *ms_correlate_total, uint32_t *launches);

/* ------------------------------------------------------------------ Bit sync + nav-bit accumulation
 * src/decoding.rs:8,40-227 (legacy file outside the reference's module tree): 20-bin histogram of prompt-I sign
 * changes (check_bit_sync :164-182, threshold 30), 20 ms accumulation into +-1 bits (bit_accumulation :184-214),
 * 8-bit preamble correlation (check_preamble_syn :216-227).  Host-side integer work, no device needed.
 * GM_NAV_FAITHFUL keeps the file's bugs (bits only emitted when frame_sync_ind == 0, :203-205; preamble tested only at
 * exactly 8 collected bits, :131-135); GM_NAV_FIXED wraps the bit boundary modulo 20 and slides the 8-bit window.
 * The subframe decoding of :147-160,229-257 panics as written (todo!(), indexing an empty Vec) and is not provided. */
enum { GM_NAV_FAITHFUL = 0, GM_NAV_FIXED = 1 };
typedef struct gm_nav_sync gm_nav_sync;
typedef struct {
    uint8_t flag_bit_sync, flag_frame_sync, sync_sw;   /* sync_sw: this epoch completed a bit */
    int8_t bit;                                        /* the completed bit (+1/-1), 0 otherwise */
    int8_t polarity;                                   /* preamble polarity (-1 until frame sync, like ::new :85) */
    uint32_t frame_sync_ind;                           /* ms offset of the bit edge */
    uint64_t n_frame_bits;
    float i_p;                                         /* running 20 ms accumulator */
    uint64_t sf_cnt, sf_start_biti, tow_expected_ind;
} gm_nav_status;
int gm_nav_sync_create(int mode, gm_nav_sync **out);                                       /* NavSyncStatus::new :68-100 */
int gm_nav_sync_destroy(gm_nav_sync *s);
/* nav_decoding's per-epoch step (:102-145) for epoch number `cnt` (1 ms each) with the channel's previous and current
 * prompt I (TrackingResult.old_i_prompt / i_prompt) */
int gm_nav_sync_update(gm_nav_sync *s, float old_i_prompt, float i_prompt, uint64_t cnt, uint64_t buff_loc,
                       gm_nav_status *out);
/* The same step for n consecutive epochs cnt0 .. cnt0 + n - 1 of one channel: i_prompt[k * stride] is epoch k's prompt I, the
 * previous one of epoch 0 is old_i_prompt0 (a caller that drives 15 channels x 1000 epochs per second through a foreign-function
 * boundary makes one call per channel and block instead of one per epoch).  *out = the status after the last epoch;
 * *first_bit_sync / *first_frame_sync (may be NULL) = the index k of the epoch at which the flag first became set in THIS call,
 * -1 if it did not.  n = 0: nothing happens, *out is left as it is. */
int gm_nav_sync_update_many(gm_nav_sync *s, float old_i_prompt0, const float *i_prompt, size_t stride, size_t n, uint64_t cnt0,
                            uint64_t buff_loc, gm_nav_status *out, int64_t *first_bit_sync, int64_t *first_frame_sync);
int gm_nav_sync_frame_bits(gm_nav_sync *s, int8_t *bits, size_t cap, size_t *n);
int gm_nav_sync_histogram(gm_nav_sync *s, uint64_t hist[20]);
/* parity_check (:259-352) on 32 symbols in +-1 form [D29*, D30*, d1..d24, D25..D30]: *ok = all six products match;
 * *ref_sum_zero (may be NULL) = the reference's own criterion, the i8 sum of the six differences is zero (:348-350) */
int gm_nav_parity_check(const int8_t bits[32], int *ok, int *ref_sum_zero);

#ifdef __cplusplus
}
#endif
#endif /* GNSS_MI355X_H */
