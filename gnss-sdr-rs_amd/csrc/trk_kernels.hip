// trk_kernels.hip — tracking kernels for gfx950 (CDNA4, wave64).
//
// Replaces TrackingChannel::{update, do_work, early_late_correlation, run_loop_filters}
// (src/tracking/do_tracking.rs:160-302) and the rayon fan-out over channels (:364-371) with
//
//   trk_correlate_kernel : grid (slices, channels).  Each workgroup wipes the carrier off its slice
//       of the channel's code period and correlates it with the E/P/L (or VE/E/P/L/VL) replicas:
//       the carrier-NCO mix and the replica multiply are fused per sample, the channel's chip row
//       sits in LDS, per-lane partial sums are reduced with wavefront shuffles, one partial vector
//       per (channel, slice) goes to HBM.
//   trk_update_kernel    : one lane per channel sums the slices in a fixed order and runs the
//       reference's scalar epilogue on-device (phase advances, lock detector, atan PLL / normalised
//       envelope DLL, loop filters, bookkeeping), so consecutive epochs need no host round trip.
//
// Per-sample arithmetic follows the reference's f32 rounding sequence (compiled with
// -ffp-contract=off, IEEE division by __fdiv_rn):
//   phase = carrier_phase + ((2*PI*carrier_freq) * (i as f32)) / fs                     (:233)
//   data *= (cos(phase), -sin(phase))          num-complex Mul                          (:234-237)
//   chip_idx = (code_phase + (i as f32) * (code_rate / fs)) % 1023.0                    (:252)
//   p/e/l = get_ca_chip(chip_idx {, +0.5, -0.5})                                         (:253-255)
// cos/sin: the f32 phase (up to ~3e4 rad) is reduced in f64 and evaluated with f64 polynomials,
// then rounded to f32 (glibc's cosf/sinf, which the reference calls, are likewise < 1 ulp).
// Sums: the reference adds sequentially in f32; here each lane adds its strided samples in order
// and lanes/waves/slices combine as a fixed tree (deterministic; closer to the exact sum).
#include "gm_internal.h"

namespace gm {

#define GM_PI_F 3.14159265358979323846f

// sin/cos of an f32 argument, evaluated in f64, rounded once to f32
__device__ __forceinline__ void sincos_f32_via_f64(float x, float& s, float& c) {
    const double xd = double(x);
    const double kd = __builtin_rint(xd * 0.63661977236758134308);   // 2/pi
    const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17;
    double r = __builtin_fma(-kd, PIO2_HI, xd);
    r = __builtin_fma(-kd, PIO2_LO, r);
    const double r2 = r * r;
    // minimax-quality Taylor cores on |r| <= pi/4 (truncation < 1e-13)
    double sp = -7.6471637318198164759e-13;                  // -1/15!
    sp = __builtin_fma(sp, r2, 1.6059043836821614599e-10);   //  1/13!
    sp = __builtin_fma(sp, r2, -2.5052108385441718775e-08);  // -1/11!
    sp = __builtin_fma(sp, r2, 2.7557319223985890653e-06);   //  1/9!
    sp = __builtin_fma(sp, r2, -1.9841269841269841270e-04);  // -1/7!
    sp = __builtin_fma(sp, r2, 8.3333333333333333333e-03);   //  1/5!
    sp = __builtin_fma(sp, r2, -1.6666666666666666667e-01);  // -1/3!
    const double sr = __builtin_fma(sp * r2, r, r);
    double cp = 4.7794773323873852974e-14;                   //  1/16!
    cp = __builtin_fma(cp, r2, -1.1470745597729724714e-11);  // -1/14!
    cp = __builtin_fma(cp, r2, 2.0876756987868098979e-09);   //  1/12!
    cp = __builtin_fma(cp, r2, -2.7557319223985890653e-07);  // -1/10!
    cp = __builtin_fma(cp, r2, 2.4801587301587301587e-05);   //  1/8!
    cp = __builtin_fma(cp, r2, -1.3888888888888888889e-03);  // -1/6!
    cp = __builtin_fma(cp, r2, 4.1666666666666666667e-02);   //  1/4!
    cp = __builtin_fma(cp, r2, -0.5);
    const double cr = __builtin_fma(cp, r2, 1.0);
    const int q = int(static_cast<long long>(kd)) & 3;
    const double sv = (q & 1) ? cr : sr, cv = (q & 1) ? sr : cr;
    s = float((q & 2) ? -sv : sv);
    c = float(((q + 1) & 2) ? -cv : cv);
}

// Rust `%` on f32 == fmodf; exact fast paths for the operating range [0, 2*len)
__device__ __forceinline__ float fmod_pos(float t, float len) {
    if (t >= 0.0f && t < len) return t;
    if (t >= len && t < 2.0f * len) return t - len;      // exact (Sterbenz)
    return fmodf(t, len);
}

// get_ca_chip's index (:275): `(phase.floor() as usize) % 1023` — the cast saturates, so a negative
// phase (late arm just after the code wraps) reads chip 0 in FAITHFUL mode; FIXED mode wraps.
__device__ __forceinline__ int chip_index(float phase, int len, int mode) {
    const float f = floorf(phase);
    if (mode == GM_CODE_INDEX_FAITHFUL) {
        if (!(f > 0.0f)) return 0;
        if (f >= 2147483648.0f) return int((unsigned long long)f % (unsigned long long)len);
        return int(f) % len;
    }
    int i = int(f) % len;
    return i < 0 ? i + len : i;
}

template <int ARMS>
__global__ __launch_bounds__(256) void trk_correlate_kernel(TrkDevCfg cfg, const int8_t* __restrict__ codes,
                                                            const gm_trk_state* __restrict__ states, TrkSrc src,
                                                            int slices, float* __restrict__ partials,
                                                            uint8_t* __restrict__ ready) {
    constexpr int NV = 2 * ARMS;
    const int ch = src.only_channel >= 0 ? src.only_channel : int(blockIdx.y);
    const int slice = blockIdx.x, tid = threadIdx.x;
    const gm_trk_state st = states[ch];
    float* pout = partials + (size_t(ch) * slices + slice) * NV;

    // update(): n = generate_ca_code_samples(..).len() = round(fs/(code_rate/len)) (:165-166,
    // ca_code.rs:13-16); early_late_correlation()/do_work() on caller samples use the field (:232)
    uint64_t n = st.num_samples_per_code;
    if (!src.linear) {
        const float nf = roundf(__fdiv_rn(cfg.fs, __fdiv_rn(st.code_rate, cfg.code_len_f)));
        n = nf > 0.0f ? uint64_t(nf) : 0;
    }
    // row of the code table: FAITHFUL indexes GPS_CA_CODE_32_PRN[prn] (:276), FIXED [prn-1]
    int row = cfg.gps_ca ? (cfg.code_index_mode == GM_CODE_INDEX_FAITHFUL ? int(st.prn) : int(st.prn) - 1)
                         : int(st.prn) - 1;
    bool run = st.active && n > 0 && n < (1ull << 31) && row >= 0 && row < cfg.n_codes;
    if (run && !src.linear)   // (head - (next + n)) as isize >= 0  (:170-172)
        run = (int64_t)(src.head - (st.next_sample_index + n)) >= 0;
    if (!run) {
        if (tid < NV) pout[tid] = 0.0f;
        if (slice == 0 && tid == 0) ready[ch] = 0;
        return;
    }

    extern __shared__ int8_t chips[];   // code_len chips of this channel's row
    const int8_t* crow = codes + size_t(row) * cfg.code_len;
    for (int i = tid; i < cfg.code_len; i += 256) chips[i] = crow[i];
    __syncthreads();

    const float two_pi_f = 2.0f * GM_PI_F * st.carrier_freq;   // (2.0*PI)*carrier_freq
    const float step = __fdiv_rn(st.code_rate, cfg.fs);        // self.code_rate / self.fs
    const int len = cfg.code_len;
    const float lenf = cfg.code_len_f;
    const int mode = cfg.code_index_mode;

    // slice bounds: whole multiples of 256 samples so lanes stay coalesced
    const uint32_t per = uint32_t(((n + slices - 1) / slices + 255) / 256 * 256);
    const uint32_t i0 = uint32_t(slice) * per;
    const uint32_t i1 = (uint64_t(i0) + per < n) ? i0 + per : uint32_t(n);

    float acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.0f;

    const uint64_t base = src.linear ? 0 : st.next_sample_index;
    for (uint32_t i = i0 + tid; i < i1; i += 256) {
        const cf d = src.base[(base + i) & src.mask];
        const float fi = float(i);
        const float phase = st.carrier_phase + __fdiv_rn(two_pi_f * fi, cfg.fs);
        float sn, cs;
        sincos_f32_via_f64(phase, sn, cs);
        const float wc = cs, ws = -sn;                          // Complex32::new(cos_p, -sin)
        const float xr = d.x * wc - d.y * ws;                   // num-complex Mul
        const float xi = d.x * ws + d.y * wc;
        const float chip_idx = fmod_pos(st.code_phase + fi * step, lenf);
        float pc = float(chips[chip_index(chip_idx, len, mode)]);
        float ec = float(chips[chip_index(chip_idx + cfg.el_space, len, mode)]);
        float lc = float(chips[chip_index(chip_idx - cfg.el_space, len, mode)]);
        if (cfg.boc11) {   // BOC(1,1): sub-carrier sign = +1 on the first half chip, -1 on the second
            const float a = chip_idx, b = chip_idx + cfg.el_space, c = chip_idx - cfg.el_space;
            pc = (a - floorf(a)) < 0.5f ? pc : -pc;
            ec = (b - floorf(b)) < 0.5f ? ec : -ec;
            lc = (c - floorf(c)) < 0.5f ? lc : -lc;
        }
        acc[0] += xr * pc; acc[1] += xi * pc;
        acc[2] += xr * ec; acc[3] += xi * ec;
        acc[4] += xr * lc; acc[5] += xi * lc;
        if constexpr (ARMS == 5) {
            const float ve = chip_idx + cfg.vel_space, vl = chip_idx - cfg.vel_space;
            float vec = float(chips[chip_index(ve, len, mode)]);
            float vlc = float(chips[chip_index(vl, len, mode)]);
            if (cfg.boc11) {
                vec = (ve - floorf(ve)) < 0.5f ? vec : -vec;
                vlc = (vl - floorf(vl)) < 0.5f ? vlc : -vlc;
            }
            acc[6] += xr * vec; acc[7] += xi * vec;
            acc[8] += xr * vlc; acc[9] += xi * vlc;
        }
    }
    // per-lane partial sums -> wavefront butterfly (64 lanes) -> 4 waves through LDS
#pragma unroll
    for (int k = 0; k < NV; ++k) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc[k] += __shfl_xor(acc[k], off, 64);
    }
    __shared__ float wsum[4][NV];
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) wsum[wave][k] = acc[k];
    }
    __syncthreads();
    if (tid < NV) pout[tid] = ((wsum[0][tid] + wsum[1][tid]) + wsum[2][tid]) + wsum[3][tid];
    if (slice == 0 && tid == 0) ready[ch] = 1;
}

// LoopFilter::update (:68-70)
__device__ __forceinline__ float loop_filter_update(float tau1, float tau2, float d_err, float err, float dt) {
    return d_err * __fdiv_rn(dt, tau1) + (d_err - err) * __fdiv_rn(tau2, tau1);
}

__device__ __forceinline__ uint64_t samples_per_code(float fs, float code_rate, float lenf) {
    const float v = roundf(__fdiv_rn(fs, __fdiv_rn(code_rate, lenf)));
    return v > 0.0f ? uint64_t(v) : 0;
}

// TrackingChannel::reset (:311-327)
__device__ __forceinline__ void reset_state(gm_trk_state& s) {
    s.prn = 0; s.active = 0; s.lost_counter = 0; s.next_sample_index = 0;
    s.carrier_freq = 0.f; s.carrier_phase = 0.f; s.carrier_error = 0.f; s.carrier_nco = 0.f;
    s.code_phase = 0.f; s.code_error = 0.f; s.code_nco = 0.f; s.code_rate = 0.f;
    s.i_prompt = 0.f; s.q_prompt = 0.f;
}

template <int ARMS>
__global__ void trk_update_kernel(TrkDevCfg cfg, gm_trk_state* __restrict__ states, const float* __restrict__ partials,
                                  const uint8_t* __restrict__ ready, int slices, int mode, int only_channel, int linear,
                                  gm_trk_out* __restrict__ outs, uint8_t* __restrict__ processed,
                                  uint8_t* __restrict__ lost, uint8_t* __restrict__ lost_prn) {
    constexpr int NV = 2 * ARMS;
    int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (only_channel >= 0) { if (ch != 0) return; ch = only_channel; }
    if (ch >= cfg.n_channels) return;
    gm_trk_out o;
    o.ip = o.qp = o.ie = o.qe = o.il = o.ql = o.ive = o.qve = o.ivl = o.qvl = 0.0f;
    uint8_t did = 0, lst = 0, lprn = 0;
    if (ready[ch]) {
        gm_trk_state s = states[ch];
        float v[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = 0.0f;
        const float* p = partials + size_t(ch) * slices * NV;
        for (int sl = 0; sl < slices; ++sl)
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] += p[sl * NV + k];
        uint64_t n = s.num_samples_per_code;
        if (!linear) { n = samples_per_code(cfg.fs, s.code_rate, cfg.code_len_f); s.num_samples_per_code = n; }  // (:166)
        const float nf = float(n);
        // carrier_phase = (carrier_phase + 2*PI*carrier_freq*(n as f32 / fs)) % (2*PI)      (:240-242)
        s.carrier_phase = fmodf(s.carrier_phase + 2.0f * GM_PI_F * s.carrier_freq * __fdiv_rn(nf, cfg.fs),
                                2.0f * GM_PI_F);
        // code_phase = (code_phase + (code_rate/fs) * n as f32) % 1023.0                     (:265-267)
        s.code_phase = fmodf(s.code_phase + __fdiv_rn(s.code_rate, cfg.fs) * nf, cfg.code_len_f);
        s.i_prompt = v[0]; s.q_prompt = v[1];
        o.ip = v[0]; o.qp = v[1]; o.ie = v[2]; o.qe = v[3]; o.il = v[4]; o.ql = v[5];
        if constexpr (ARMS == 5) { o.ive = v[6]; o.qve = v[7]; o.ivl = v[8]; o.qvl = v[9]; }
        did = 1;
        if (mode == TRK_MODE_DO_WORK) {                      // do_work (:183-210)
            const float power = v[0] * v[0] + v[1] * v[1];
            bool advance = true;
            if (power > cfg.lock_threshold) {
                s.lost_counter = 0;
                // run_loop_filters (:279-302)
                const float pll_err = __fdiv_rn(atanf(__fdiv_rn(v[1], v[0])), 2.0f * GM_PI_F);
                s.carrier_nco = loop_filter_update(cfg.pll_tau1, cfg.pll_tau2, pll_err, s.carrier_error, cfg.pll_dt);
                s.carrier_error = pll_err;
                s.carrier_freq += s.carrier_nco;
                const float pow_e = __fsqrt_rn(v[2] * v[2] + v[3] * v[3]);
                const float pow_l = __fsqrt_rn(v[4] * v[4] + v[5] * v[5]);
                const float dll_err = ((pow_e + pow_l) != 0.0f) ? __fdiv_rn(pow_e - pow_l, pow_e + pow_l) : 0.0f;
                s.code_nco = loop_filter_update(cfg.dll_tau1, cfg.dll_tau2, dll_err, s.code_error, cfg.dll_dt);
                s.code_error = dll_err;
                s.code_rate += s.code_nco;
            } else {
                s.lost_counter += 1;
                if (s.lost_counter >= cfg.max_lost_epochs) {
                    reset_state(s);                          // reset() first ...
                    lst = 1; lprn = s.prn;                   // ... so the message carries prn 0 (:199-201)
                    advance = false;
                }
            }
            if (advance) {
                s.next_sample_index += n;                    // (:192 / :203)
                s.num_samples_per_code = samples_per_code(cfg.fs, s.code_rate, cfg.code_len_f);
            }
        }
        states[ch] = s;
    }
    if (outs) outs[ch] = o;
    if (processed) processed[ch] = did;
    if (lost) lost[ch] = lst;
    if (lost_prn) lost_prn[ch] = lprn;
}

void launch_trk_epoch(hipStream_t st, const TrkDevCfg& cfg, const int8_t* d_codes, gm_trk_state* d_states,
                      const TrkSrc& src, int slices, float* d_partials, uint8_t* d_ready, int mode,
                      gm_trk_out* d_outs, uint8_t* d_processed, uint8_t* d_lost, uint8_t* d_lost_prn) {
    const int nch = src.only_channel >= 0 ? 1 : cfg.n_channels;
    const dim3 grid(slices, nch);
    const size_t lds = size_t(cfg.code_len);
    const int ub = 64, ug = src.only_channel >= 0 ? 1 : (cfg.n_channels + ub - 1) / ub;
    if (cfg.n_arms == 5) {
        hipLaunchKernelGGL(trk_correlate_kernel<5>, grid, dim3(256), lds, st, cfg, d_codes, d_states, src, slices,
                           d_partials, d_ready);
        hipLaunchKernelGGL(trk_update_kernel<5>, dim3(ug), dim3(ub), 0, st, cfg, d_states, d_partials, d_ready,
                           slices, mode, src.only_channel, src.linear, d_outs, d_processed, d_lost, d_lost_prn);
    } else {
        hipLaunchKernelGGL(trk_correlate_kernel<3>, grid, dim3(256), lds, st, cfg, d_codes, d_states, src, slices,
                           d_partials, d_ready);
        hipLaunchKernelGGL(trk_update_kernel<3>, dim3(ug), dim3(ub), 0, st, cfg, d_states, d_partials, d_ready,
                           slices, mode, src.only_channel, src.linear, d_outs, d_processed, d_lost, d_lost_prn);
    }
}

}  // namespace gm
