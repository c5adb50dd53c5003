// gm_internal.h — shared declarations between the kernel translation units and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gnss_mi355x.h"
#include "fft_core.h"

namespace gm {

// Fine Doppler (finer_doppler, acquisition_bk.rs:215-302): the zero-padded long FFT of the code-stripped
// snapshot as a four-step N1 x N2 transform built from two in-LDS plans; never materialised: the column pass reads the
// samples (strip + zero-pad fused), the row pass reduces |X|^2 to {max, first index} per row.
struct FineArgs {
    const void* samples; int fmt;          // the snapshot of the last search (M*N samples)
    const float* mean;                     // device {re, im}
    const int8_t* chips; uint32_t code_len; float code_rate, fs;   // [P][code_len]
    const uint32_t* sat_worker;            // [S] worker index of each satellite to refine
    const uint32_t* sat_code_phase;        // [S]
    uint32_t size_use, N1, N2;             // fft_size = N1 * N2
    cf* B;                                 // [S][N2][N1] intermediate (columns transformed, twiddled)
    const cf *tw1, *tw2;                   // forward base twiddles of plan N1 / plan N2
    float* rowmax; uint32_t* rowarg;       // [S][N1]
};

// One entry per shipped transform size: launchers for the kernels instantiated on that plan.
struct DecideArgs;
struct PlanOps {
    int n;             // transform length
    int threads;       // workgroup size
    int tw_total;      // base-twiddle table entries (per direction)
    int lds_bytes;     // static LDS per workgroup of the correlation kernel
    int split_slab;    // floats of one partial power plane of the tail split (0: this plan cannot split)
    int code_paired;   // 1: corr() takes the code spectra in the paired layout (pair_codes), 0: in natural order
    void (*fill_tw)(cf* tw, bool inverse);
    // order[p] = the spectrum element index stored at position p of a stored spectrum, for sizes whose correlation plan reads its
    // input in a permuted order (prime-factor / hybrid plans); returns the table length (0: natural or merely paired order,
    // no table).  order == nullptr: just the length.
    int (*fill_order)(uint16_t* order);
    // stage F: carrier mix (apply_doppler_shift, doppler_shift.rs:25-58) fused into the forward FFT
    // (do_acquisition.rs:177-182).  One workgroup per (doppler bin, ms block); shared by all PRNs.
    // clear_tickets (may be null): the tail split's ticket counters, zeroed by the first workgroup for the corr() launch
    // that follows on the same stream (saves that launch its own hipMemsetAsync: ~8 us per dwell)
    // dec (may be null): a decision deferred by gm_acq_decide_dev — the PREVIOUS dwell's — runs as dec->n_prn trailing workgroups
    void (*mix_fft)(hipStream_t, const void* samples, int fmt, const cf* tables, const cf* tw_fwd,
                    cf* spectra, int n_bins, int n_int, uint32_t* clear_tickets, const uint16_t* order, const DecideArgs* dec);
    // stage C (spectra and code_fft in the PAIRED layout): x conj(code spectrum) -> inverse FFT -> |.|^2 accumulated over the integrations ->
    // {max, first argmax, sum} per (worker, bin)  (do_acquisition.rs:184-202, 229-235)
    void (*corr)(hipStream_t, const cf* spectra, const cf* code_fft, const cf* tw_inv, float* mmax,
                 uint32_t* margmax, float* msum, const uint32_t* worker_list, int n_workers, int n_bins,
                 int n_int, float* split_scratch, int split_planes, uint32_t* split_counter, int strict_sum, int tickets_cleared,
                 int ref_mul);   // ref_mul: gm_acq_cfg.reference_products;   // split_planes: power planes the scratch holds (<= GM_CORR_SPLIT_MAX_SLABS); tickets_cleared: mix_fft(.., split_counter) ran just before on this stream
    // AcquisitionWorker::new's replica spectrum (do_acquisition.rs:132-138)
    void (*code_fft)(hipStream_t, const int8_t* code_samples, const cf* tw_fwd, cf* code_fft, int n_codes);
    // the same spectra re-stored in the paired layout stage C reads (PairLayout in acq_kernels.hip); stage F writes its
    // spectra in that layout directly
    void (*pair_codes)(hipStream_t, const cf* natural, cf* paired, int n_codes);
    // FFT<T>::execute (fft.rs:21-25) on `batch` contiguous transforms
    void (*fft_batch)(hipStream_t, cf* data, const cf* tw, int inverse, int batch);
    // four-step long FFT passes (power-of-two plans only, else null): this plan as N1 (columns) / as N2 (rows)
    void (*fine_cols)(hipStream_t, const FineArgs&, int n_sats);
    void (*fine_rows)(hipStream_t, const FineArgs&, int n_sats);
    int fine_rows_per_wg;   // rows of the row pass one workgroup takes (its results: N1 / this per satellite)
    // long power-of-two FFT, four-step (power-of-two plans only, else null): columns x -> B[n2][k1] (twiddled), rows B -> X natural
    void (*big_cols)(hipStream_t, const cf* x, cf* B, const cf* tw, uint32_t n2, int inverse);
    void (*big_rows)(hipStream_t, const cf* B, cf* X, const cf* tw, uint32_t n1, int inverse);
    // stage F's forward transform may run on a plan of its own (MixPlanOf, acq_device.h): ITS base-twiddle table is what mix_fft takes
    int tw_total_mix;
    void (*fill_tw_mix)(cf* tw, bool inverse);
    // the inverse base-twiddle table of acq_corr_kernel, built for ITS plan (CorrPlanOf, acq_device.h: a plain plan of other radices
    // than the registered one since round 6, e.g. N = 8192; empty for prime-factor and hybrid correlation plans)
    int tw_total_corr;
    void (*fill_tw_corr)(cf* tw);
};
// mean of the snapshot (finer_doppler :236) and the final per-satellite reduction over the rows
void launch_fine_mean(hipStream_t, const void* samples, int fmt, uint32_t n, float* d_mean);
void launch_fine_final(hipStream_t, const float* rowmax, const uint32_t* rowarg, uint32_t n_rows, int n_sats,
                       float* peak_pow, uint32_t* peak_idx);
// tail split of the correlation grid: per XCD about one resident round of slots (64) worth of parts, one part per
// integration (up to GM_CORR_SPLIT_MAX_K integrations; longer dwells are not cut); the scratch holds GM_CORR_SPLIT_MAX_SLABS
// power planes — one per part — and there is one ticket per cut item
constexpr int GM_CORR_SPLIT_MAX_K = 16;
constexpr int GM_CORR_SPLIT_MAX_SLABS = 8 * 320;
constexpr int GM_CORR_SPLIT_MAX_ITEMS = 8 * 80;
// Diagnostic overrides (item maps, tail split, workgroups per channel ...: tools/README.md).  The library reads NO environment
// variable unless the process was started with GM_DIAGNOSTICS=1: a receiver that links this library must not change kernels
// because its environment happens to carry a GM_* name.  diag_int returns `dflt` when diagnostics are off or `name` is unset.
int diag_int(const char* name, int dflt);
const PlanOps* find_plan(int n);
int list_plans(uint32_t* sizes, int cap);

// diagnostic: device buffer [n_int][8 waves][8 phases] of s_memtime stamps written by workgroup 0 of acq_corr_kernel
void set_corr_stamps(long long* d_ptr);
bool corr_stamps_built();          // false in the product library (the stamped kernels are compiled under -DGM_DIAG_STAMPS only)

// composite transform sizes N = Q * Nb (acq_composite.hip): decimated in time, the inverse fused with the power reduction
struct CompOps {
    int nb, q;         // base in-LDS plan length and the factor: N = q * nb
    // forward step 1: Q in-LDS transforms per item over the decimated inputs (signal: carrier mix fused; codes: int8 chips)
    // order (may be null): the base plan's PlanOps::fill_order table -> A leaves in storage order (permuted correlation plans)
    void (*fwd_sub)(hipStream_t, const void* samples, int fmt, const cf* tables, const int8_t* code_samples,
                    const cf* tw_fwd, cf* A, uint32_t n_items, uint32_t n_int, const uint16_t* order);
    // forward step 2: twiddle + Q-point DFTs -> natural block order; paired != 0 stores each block in the paired layout
    void (*fwd_post)(hipStream_t, const cf* A, cf* X, uint32_t n_items, int paired, const uint16_t* order);
    // inverse, fused: {max, first argmax, sum} per (worker, bin) straight from the spectra
    // planes (null normally; gm_acq_cfg.strict_sum_order): [n_workers_total * n_bins][N] floats — the kernel also stores every accumulated
    // power value at its natural index, for launch_plane_strict_sum
    void (*corr)(hipStream_t, const cf* spectra, const cf* code_paired, const cf* twn, const cf* tw_inv, float* mmax,
                 uint32_t* margmax, float* msum, const uint32_t* worker_list, int n_workers, int n_bins, int n_int, float* planes);
    void (*fill_twn)(cf* out);   // host: [q][nb] inverse twiddles W_N^{-n1 k2} in the paired position of k2
    // once per handle: comb[p][n1][k1][pos] = conj(code_paired[p][k1][pos]) * W_Q^{-n1 k1} * twn[n1][pos], the whole code-side
    // factor of sub-transform n1 (the codes are static, so corr multiplies a spectrum value by ONE table entry)
    void (*comb)(hipStream_t, const cf* code_paired, const cf* twn, cf* comb, uint32_t n_codes);
    // the stored order of a block of Nb spectrum elements on THIS path (its plan may differ from the fused kernels': CompPlanOf):
    // order table as PlanOps::fill_order, and natural -> stored re-layout of n_blocks blocks
    int (*fill_order)(uint16_t* order);
    void (*relayout)(hipStream_t, const cf* natural, cf* stored, int n_blocks);
    // base plans with an order table: forward step 2 folded into the table — comb2[p][n1][n1'][pos] such that a sub-transform's input is
    // sum_n1' A[n1'][pos] * comb2[n1][n1'][pos] on the forward sub-transforms A as fwd_sub leaves them (no fwd_post per dwell)
    void (*fold_post)(hipStream_t, const cf* comb, const uint16_t* order, cf* comb2, uint32_t n_codes);
};
const CompOps* find_comp(uint32_t n);
// strict_sum_order on the composite path: msum[o] = is_good_satellite's eight-lane ordered sum (do_acquisition.rs:229-235) of plane o,
// for the planes of the listed workers (o = worker * n_bins + bin)
void launch_plane_strict_sum(hipStream_t, const float* planes, float* msum, const uint32_t* worker_list, int n_workers, int n_bins, uint32_t N);

// elementwise apply_doppler_shift (doppler_shift.rs:25-58)
void launch_apply_doppler(hipStream_t, const cf* s, const cf* t, cf* out, size_t n);
// |X|^2 (fft.rs:27-29)
void launch_power(hipStream_t, const cf* x, float* p, size_t n);

// decision replay (do_acquisition.rs:195-225 + 229-238) from per-(worker,bin) metrics
struct DecideArgs {
    const float* mmax; const uint32_t* margmax; const float* msum;  // [n_prn][n_bins]
    const float* table_freq;                                        // [n_bins]
    const uint8_t* prn_ids;                                         // [n_prn]
    uint64_t mask_lo;                                               // worker i searched <-> bit i (i < 64), else all
    int n_prn, n_bins, fft_size;
    float fs, threshold, code_rate;
    int best_bin_mode;                                              // 0: reference early exit, 1: strongest bin
    uint64_t local_tail;
    gm_acq_result* results; uint8_t* found;                         // [n_prn]
};
void launch_decide(hipStream_t, const DecideArgs&);

// ---------------------------------------------------------------- tracking
struct TrkDevCfg {
    float fs;
    int n_channels, n_arms;
    float el_space, vel_space;
    int code_index_mode, boc11;
    int code_len;             // chips per period
    float code_len_f;
    int gps_ca;               // 1: built-in C/A table rows (row = prn or prn-1 by mode)
    int n_codes;
    float lock_threshold; uint32_t max_lost_epochs;
    float pll_tau1, pll_tau2, dll_tau1, dll_tau2, pll_dt, dll_dt;
    float nominal_code_rate;
    // per-launch constants of the scalar epilogue, computed once on the host with IEEE f32 / f64 division
    // (fill_trk_derived): quotients of configuration constants and correctly rounded reciprocals for the
    // constant-divisor divisions
    float inv_fs, inv_len, inv_2pi;
    float pll_dt_tau1, pll_tau2_tau1, dll_dt_tau1, dll_tau2_tau1;   // dt/tau1, tau2/tau1 (LoopFilter::update :68-70)
    int div_fs_ok;            // fs significand not all ones: div_const(x, fs) is the correctly rounded quotient
    int strict_libm;          // gm_trk_cfg.strict_libm: carrier cos / sin by gm_libm.h's sincosf_glibc
    int strict_sum_order;     // gm_trk_cfg.strict_sum_order: the six / ten sums added sample by sample (trk_serial_sum_kernel)
};
inline void fill_trk_derived(TrkDevCfg& d) {
    d.inv_fs = 1.0f / d.fs; d.inv_len = 1.0f / d.code_len_f; d.inv_2pi = 1.0f / (2.0f * 3.14159265358979323846f);
    d.pll_dt_tau1 = d.pll_dt / d.pll_tau1; d.pll_tau2_tau1 = d.pll_tau2 / d.pll_tau1;
    d.dll_dt_tau1 = d.dll_dt / d.dll_tau1; d.dll_tau2_tau1 = d.dll_tau2 / d.dll_tau1;
    uint32_t bits; __builtin_memcpy(&bits, &d.fs, 4);
    d.div_fs_ok = ((bits & 0x7fffffu) != 0x7fffffu && d.fs > 1.0f && d.fs < 1.0e12f) ? 1 : 0;
}
enum { TRK_MODE_CORRELATE = 0, TRK_MODE_DO_WORK = 1 };
// sample source: ring (mask = size-1, absolute index = next_sample_index + i) or linear (mask = ~0, base 0)
struct TrkSrc {
    const cf* base; uint64_t mask; uint64_t head; int linear;   // linear: index i directly, no head test
    int only_channel;                                           // >= 0: run just this channel (unit entries)
};
void launch_trk_epoch(hipStream_t, const TrkDevCfg&, const int8_t* d_codes, gm_trk_state* d_states,
                      const TrkSrc&, int slices, float* d_partials, uint8_t* d_ready, int mode,
                      gm_trk_out* d_outs, uint8_t* d_processed, uint8_t* d_lost, uint8_t* d_lost_prn,
                      float* d_terms = nullptr, size_t terms_cap = 0, int* d_error = nullptr);   // (strict_sum_order: per-sample product streams)

// Persistent tracking geometry: workgroups of TRK_PERSIST_THREADS lanes, TRK_PERSIST_WG_PER_CU of them per CU.
// Two independent workgroups per CU let one channel's serial exchange + loop-filter epilogue (one wave) overlap
// the other's correlation phase.
#ifndef GM_TRK_THREADS          // (A/B switch of the workgroup shape: 256 x 4 per CU and 1024 x 1 were measured in round 6, DESIGN_HISTORY R6.10)
#define GM_TRK_THREADS 512
#define GM_TRK_WG_PER_CU 2
#endif
constexpr int TRK_PERSIST_THREADS = GM_TRK_THREADS;
constexpr int TRK_PERSIST_WG_PER_CU = GM_TRK_WG_PER_CU;

inline int trk_persistent_slots(int n_channels) { return (n_channels + 7) / 8 * 8; }   // grid = slots * G workgroups
int trk_persistent_blocks_per_cu(const TrkDevCfg&);   // resident workgroups per CU of the instantiation this config selects
// persistent multi-epoch tracking (one launch = `epochs` passes over all channels); G workgroups per channel
int trk_persistent_granule_stride(int G);      // granules per arm in the exchange block of the persistent kernel (16 for G <= 16)
void launch_trk_results_to_host(hipStream_t, const void* d_src, void* h_dst_pinned, size_t bytes, const void* d_src2 = nullptr, void* h_dst2_pinned = nullptr,
                                size_t bytes2 = 0);
void launch_trk_persistent(hipStream_t, const TrkDevCfg&, const int8_t* d_codes, gm_trk_state* d_states,
                           const cf* ring, uint64_t mask, uint64_t head, int G, int packed, int epochs, uint32_t tag_base,
                           unsigned long long* d_xchg, gm_trk_out* d_outs, uint8_t* d_processed, uint8_t* d_lost,
                           uint8_t* d_lost_prn, int* d_error, int* d_error_dev, long long* d_stamps);

// ---------------------------------------------------------------- digital front-end (fe_kernels.hip)
struct FeState { float phase_accumulator; float bias_re[8]; float bias_im[8]; };   // NcoLut.phase_accumulator, DcRemoverSimd.bias_*
struct FrontendArgs {
    struct Stream {
        const void* in;            // n_samples of c32 or int8 IQ (device)
        void* out;                 // c32 destination: linear buffer (out_mask = ~0) or ring base (out_mask = size-1)
        uint64_t out_start, out_mask;
        size_t n_samples;
        FeState* state;
        float phase_step;          // NcoLut.phase_step of this stream's front-end
        int fast_fmod;             // |phase_step| < 2048: fmodf reduces to one exact conditional add/subtract
        // tabulated NCO phase orbit (fast kernel; null: the sequential phase chain of frontend_kernel)
        const float* ph_table;     // |phase_accumulator| / 2048 before step k of the orbit that starts at phase 0
        uint32_t tab_len;          // entries: transient mu + period lambda' (lambda' = the period repeated to >= FE_FAST_SEG)
        uint32_t tab_lambda;       // lambda'
        uint32_t tab_pos;          // table index of this call's first sample (< tab_len)
        uint32_t tab_pos_end;      // table index after this call's last processed sample
        float tab_scale;           // +-2048: sign of phase_step
    };
    const Stream* streams;         // device array, one entry (and one workgroup) per stream
    Stream one;                    // used when streams == nullptr (single-stream launches need no upload)
    const float* lut;              // [2][2048]: lut_re, lut_im (nco_lut.rs:28-32), built on the host with glibc cosf/sinf
    float alpha, con;
    // speculative form (launch_frontend_spec): the block of `one` on spec_k workgroups; spec_buf = [spec_k][32] floats (the state run s
    // entered with / left with) + one word (repairs, diagnostic)
    int spec_k = 0, spec_poison = 0, spec_warm = 0;      // spec_warm: warm-up pipeline segments (0: the kernel's default, 18)
    float* spec_buf = nullptr;
    unsigned int* spec_repairs = nullptr;   // runs the walk has had to repeat (diagnostic counter)
};
void launch_frontend_spec(hipStream_t, const FrontendArgs&, int fmt);
void launch_frontend(hipStream_t, const FrontendArgs&, int n_streams, int fmt);
// every stream carries a phase table: the pipelined kernel (DC chains on one wave, everything else data-parallel)
void launch_frontend_fast(hipStream_t, const FrontendArgs&, int n_streams, int fmt);
constexpr int FE_SPEC_K_MAX = 64;     // most workgroups (runs) per block in the speculative form: the record block's size (32 are used)
constexpr int FE_FAST_SEG = 3840;      // samples per pipeline segment of the fast kernel: 480 chain steps = 120 quads x 8 SIMD
                                       // lanes = 960 stage-C items = one per helper lane (the table period is padded to >= this)

}  // namespace gm
