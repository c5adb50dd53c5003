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
//   trk_persistent_kernel : the production path — one launch runs up to 4095 passes over all channels (below).
//   trk_terms_kernel + trk_serial_sum_kernel : gm_trk_cfg.strict_sum_order — every sample's products, then the
//       reference's own sequential f32 sums by one wave per channel (bit-identical sums; with strict_libm, state).
//
// Per-sample arithmetic follows the reference's f32 rounding sequence (compiled with
// -ffp-contract=off, IEEE division by __fdiv_rn):
//   phase = carrier_phase + ((2*PI*carrier_freq) * (i as f32)) / fs                     (:233)
//   data *= (cos(phase), -sin(phase))          num-complex Mul                          (:234-237)
//   chip_idx = (code_phase + (i as f32) * (code_rate / fs)) % 1023.0                    (:252)
//   p/e/l = get_ca_chip(chip_idx {, +0.5, -0.5})                                         (:253-255)
// cos/sin: the f32 phase (up to ~3e4 rad) is reduced in f64 (general path) or by an exact f32 Cody-Waite split (fast path)
// and evaluated with f32 minimax polynomials: < 1 ulp like glibc's cosf/sinf, which the reference calls, but not the same
// bits on a quarter of the arguments; gm_trk_cfg.strict_libm switches to gm_libm.h's restatement of glibc's own algorithm.
// Sums: the reference adds sequentially in f32; here each lane adds its strided samples in order
// and lanes/waves/slices combine as a fixed tree (deterministic; closer to the exact sum).
#include "gm_internal.h"
#include "gm_libm.h"

namespace gm {

#define GM_PI_F 3.14159265358979323846f

// a wave-uniform float, moved to a scalar register.  Also used to keep loop-invariant set-up code next to its (rare) use:
// hoisted out of the epoch loop such values sat in VGPRs for the whole launch and were what the register allocator spilled.
__device__ __forceinline__ float uniform_f32(float x) {
    uint32_t u = __builtin_amdgcn_readfirstlane(__float_as_uint(x));
    asm volatile("" : "+s"(u));       // opaque and pinned to this point of the program (not hoisted, not folded)
    return __uint_as_float(u);
}

// sin/cos of an f32 argument up to ~1e5 rad in magnitude: the f32 value is reduced EXACTLY in f64
// (r = x - k*pi/2 with a two-term pi/2, |r| <= pi/4, error < 1e-11), then evaluated with f32 minimax
// polynomials (Cephes sinf/cosf cores; max error 1.56 * 2^-24, measured in tests/cpu/test_libm.cpp).  glibc's cosf/sinf,
// which the reference calls, are < 1 ulp; the two differ in the last bit on a quarter of the samples, never by more than 2 ulp.
// Used where the phase may be large (the general path); the fast path uses gm_libm.h's sincos_cw (f32 reduction).
__device__ __forceinline__ void sincos_f32_via_f64(float x, float& s, float& c) {
    const double xd = double(x);
    const double kd = __builtin_rint(xd * 0.63661977236758134308);   // 2/pi
    double r = __builtin_fma(-kd, 1.57079632679489655800e+00, xd);
    r = __builtin_fma(-kd, 6.12323399573676603587e-17, r);
    const float rf = float(r);
    const float z = rf * rf;
    float sp = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    sp = __builtin_fmaf(sp, z, -1.6666654611e-1f);
    const float sr = __builtin_fmaf(sp * z, rf, rf);
    float cp = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    cp = __builtin_fmaf(cp, z, 4.166664568298827e-2f);
    const float cr = __builtin_fmaf(cp * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    const int q = int(kd) & 3;   // |kd| < 2^31 for |x| < 3e9
    const float sv = (q & 1) ? cr : sr, cv = (q & 1) ? sr : cr;
    s = (q & 2) ? -sv : sv;
    c = ((q + 1) & 2) ? -cv : cv;
}


// (code_phase + i*step) % len.  FAST: the caller has checked once per epoch that code_phase lies in
// (-len, len) and i*step in [0, 2*len), so t lies in (-len, 3*len) and each branch below is exact
// (Sterbenz) and equals fmodf(t, len): sign of the dividend, magnitude < len.  !FAST: plain fmodf.
template <bool FAST> __device__ __forceinline__ float fmod_code(float t, float len) {
    if constexpr (FAST) {   // both differences first, then two selects: straight-line code (the nested ternary became branches)
        const float two = 2.0f * len, a = t - len, b = t - two;
        float r = t >= len ? a : t;
        r = t >= two ? b : r;
        return r;
    } else return fmodf(t, len);
}

// get_ca_chip's index (:275): `(phase.floor() as usize) % 1023` — the cast saturates, so a negative
// phase (late arm just after the code wraps) reads chip 0 in FAITHFUL mode; FIXED mode wraps.
// phase = chip_idx +- spacing with |chip_idx| < len and spacing < len (checked at gm_trk_create), so
// one conditional subtraction replaces the modulo.
__device__ __forceinline__ int chip_index(float phase, int len, int mode) {
    const float f = floorf(phase);
    if (mode == GM_CODE_INDEX_FAITHFUL) {
        if (!(f > 0.0f)) return 0;
        const int i = int(f);
        return i >= len ? i - len : i;
    }
    int i = int(f);
    if (i < 0) i += len;
    if (i < 0) i += len;
    return i >= len ? i - len : i;
}

// LoopFilter::update (:68-70): d_err * (dt / tau1) + (d_err - err) * (tau2 / tau1); the two quotients are
// configuration constants, divided once on the host (TrkDevCfg::*_tau1, same IEEE f32 quotients)
__device__ __forceinline__ float loop_filter_update(float dt_over_tau1, float tau2_over_tau1, float d_err, float err) {
    return d_err * dt_over_tau1 + (d_err - err) * tau2_over_tau1;
}

// (div_const and fmod_bounded: gm_libm.h, host/device portable and checked on the CPU by tests/cpu/test_libm.cpp)

// generate_ca_code_samples(..).len() = round(fs / (code_rate / len))  (ca_code.rs:13-16)
__device__ __forceinline__ uint64_t samples_per_code(float fs, float code_rate, float lenf) {
    const float v = roundf(__fdiv_rn(fs, __fdiv_rn(code_rate, lenf)));
    return v > 0.0f ? uint64_t(v) : 0;
}
__device__ __forceinline__ uint64_t samples_per_code(const TrkDevCfg& cfg, float code_rate) {   // same value, one true division
    const float v = roundf(__fdiv_rn(cfg.fs, div_const(code_rate, cfg.code_len_f, cfg.inv_len)));
    return v > 0.0f ? uint64_t(v) : 0;
}

// TrackingChannel::reset (:311-327)
__device__ __forceinline__ void reset_state(gm_trk_state& s) {
    s.prn = 0; s.active = 0; s.lost_counter = 0; s.next_sample_index = 0;
    s.carrier_freq = 0.f; s.carrier_phase = 0.f; s.carrier_error = 0.f; s.carrier_nco = 0.f;
    s.code_phase = 0.f; s.code_error = 0.f; s.code_nco = 0.f; s.code_rate = 0.f;
    s.i_prompt = 0.f; s.q_prompt = 0.f;
}

#ifndef GM_TRK_IL5
#define GM_TRK_IL5 2          // samples per straight-line block of the five-arm fast path (2 or 4)
#endif
// persistent kernel's dynamic LDS.  Plain: [code_len + 2 floats of the padded chip row].  BOC: [2 code_len + 4 half chips, whole
// float4s][the padded chip row] — the table the fast path reads sits FIRST, at a compile-time LDS address, so that its look-ups are
// `v_lshlrev` + `ds_read offset:constant` (a run-time base costs a half-rate v_lshl_add with a scalar source per look-up)
__host__ __device__ __forceinline__ int boc_plain_offset(int code_len) { return (2 * code_len + 4 + 3) & ~3; }

// per-epoch constants of one channel, derived from its state exactly once per epoch
struct EpochConsts {
    float carrier_phase, two_pi_f, code_phase, step, fs, inv_fs, lenf, el, vel;
    int len, mode, boc11, strict;
};
__device__ __forceinline__ EpochConsts epoch_consts(const TrkDevCfg& cfg, const gm_trk_state& st) {
    EpochConsts c;
    c.carrier_phase = st.carrier_phase;
    c.two_pi_f = 2.0f * GM_PI_F * st.carrier_freq;      // (2.0*PI)*carrier_freq
    c.code_phase = st.code_phase;
    c.step = cfg.div_fs_ok ? div_const(st.code_rate, cfg.fs, cfg.inv_fs) : __fdiv_rn(st.code_rate, cfg.fs);   // code_rate / fs
    c.inv_fs = cfg.inv_fs;                              // correctly rounded reciprocal for div_by_fs()
    c.fs = cfg.fs; c.lenf = cfg.code_len_f; c.len = cfg.code_len; c.mode = cfg.code_index_mode;
    c.boc11 = cfg.boc11; c.el = cfg.el_space; c.vel = cfg.vel_space;
    c.strict = cfg.strict_libm;
    return c;
}

// once per epoch: may the exact fast forms be used for samples 0..n-1 ?
//  * fmod_code: code_phase in [0, len), 0 <= step, n*step < 1.99 len -> one conditional subtraction, result in [0, len);
//  * chip look-ups: with the chip phase in [0, len) and arm spacings in (0, 1] every arm's floor lies in [-1, len], so one
//    select per arm replaces the general modulo (chip_index_arm);
//  * x / fs: q0 = x*r, q = fma(fma(-q0, fs, x), r, q0) with r = RN(1/fs) is the correctly rounded quotient (Markstein)
//    unless fs's significand is all ones or the quotient leaves the normal range — excluded here.  Checked against IEEE
//    division on 1.3e9 operands for 21 sample rates (tools/ubench note in DESIGN.md); only the sign of a zero quotient can
//    differ, which cannot change a sample's products.
__device__ __forceinline__ bool fast_code_ok(const EpochConsts& c, uint64_t n) {     // the code replica's share
    const bool code_ok = n < (1ull << 24) && c.code_phase >= 0.0f && c.code_phase < c.lenf && c.step >= 0.0f &&
                         float(uint32_t(n)) * c.step < 1.99f * uniform_f32(c.lenf);    // n < 2^24: sample indices are exact as floats
    const bool arms_ok = c.el > 0.0f && c.el <= 1.0f && c.vel > 0.0f && c.vel <= 1.0f;
    return code_ok && arms_ok;
}
// n_cap: an upper bound of the epoch's sample count (the phase must stay below 1e5 rad for sincos_cw: an IF of a few MHz
// over a 1 ms epoch is ~3e4 rad)
__device__ __forceinline__ bool fast_car_ok(const EpochConsts& c, float n_cap) {      // the carrier's share
    return (__float_as_uint(c.fs) & 0x7fffffu) != 0x7fffffu && c.fs > 1.0f && c.fs < 1.0e12f &&
           (c.two_pi_f == 0.0f || (fabsf(c.two_pi_f) > 1.0e-12f && fabsf(c.two_pi_f) < 1.0e12f)) &&
           fabsf(c.carrier_phase) + fabsf(c.two_pi_f) * (uniform_f32(n_cap) * c.inv_fs) < 1.0e5f;
}
__device__ __forceinline__ bool fast_code_range(const EpochConsts& c, uint64_t n) {
    return fast_code_ok(c, n) && fast_car_ok(c, float(uint32_t(n)));
}

// x / fs, correctly rounded (see fast_code_range)
__device__ __forceinline__ float div_by_fs(float x, float fs, float inv_fs) {
    const float q0 = x * inv_fs;
    return __builtin_fmaf(__builtin_fmaf(-q0, fs, x), inv_fs, q0);
}

// get_ca_chip's index (:275) for phase = chip_idx +- spacing when floor(phase) is known to lie in [-1, len]:
// len -> 0 (`% len`), -1 -> 0 in FAITHFUL mode (`as usize` saturates) or len-1 in FIXED mode (wrap)
__device__ __forceinline__ int chip_index_arm(float phase, int len, int mode) {
    const int i = int(floorf(phase));
    if (mode == GM_CODE_INDEX_FAITHFUL) return i >= len ? 0 : (i < 0 ? 0 : i);
    return i >= len ? 0 : (i < 0 ? len - 1 : i);
}

// one sample: carrier wipe-off fused with the replica multiplies (early_late_correlation :231-263)
// MODE_T / BOC_T: compile-time code-index mode and BOC flag (straight-line code the scheduler can interleave
// across samples), or -1 to read them from the epoch constants at run time (unit-entry kernels).
// CT: int8_t — the chip row as stored (unit-entry kernels); float — the persistent kernel's padded row (chip k at [k + 1])
template <class CT> __device__ __forceinline__ float chip_at(const CT* t, int k);
template <> __device__ __forceinline__ float chip_at<int8_t>(const int8_t* t, int k) { return float(t[k]); }
template <> __device__ __forceinline__ float chip_at<float>(const float* t, int k) { return t[k + 1]; }

template <int ARMS, bool FAST, int MODE_T = -1, int BOC_T = -1, class CT = int8_t, int STRICT_T = -1>
__device__ __forceinline__ void correlate_sample(const EpochConsts& c, const CT* chips, cf d, uint32_t i,
                                                 float (&acc)[2 * ARMS]) {
    const int mode = MODE_T >= 0 ? MODE_T : c.mode;
    const bool boc = BOC_T >= 0 ? (BOC_T != 0) : (c.boc11 != 0);
    const float fi = float(i);
    const float w = c.two_pi_f * fi;
    const float phase = c.carrier_phase + (FAST ? div_by_fs(w, c.fs, c.inv_fs) : __fdiv_rn(w, c.fs));
    float sn, cs;
    // gm_trk_cfg.strict_libm: glibc's cosf / sinf bit for bit.  STRICT_T: compile-time in the persistent kernel (its default
    // instantiations carry no trace of the f64 path), run-time in the unit-entry kernels (a uniform branch)
    if (STRICT_T >= 0 ? (STRICT_T != 0) : (c.strict != 0)) sincosf_glibc(phase, sn, cs);
    else sincos_f32_via_f64(phase, sn, cs);
    const float wc = cs, ws = -sn;                          // Complex32::new(cos_p, -sin)
    const float xr = d.x * wc - d.y * ws;                   // num-complex Mul
    const float xi = d.x * ws + d.y * wc;
    const float chip_idx = fmod_code<FAST>(c.code_phase + fi * c.step, c.lenf);
    float pc, ec, lc;
    if (FAST) {   // chip_idx in [0, len): the prompt index needs no reduction, the arms one select each
        pc = chip_at(chips, int(floorf(chip_idx)));
        ec = chip_at(chips, chip_index_arm(chip_idx + c.el, c.len, mode));
        lc = chip_at(chips, chip_index_arm(chip_idx - c.el, c.len, mode));
    } else {
        pc = chip_at(chips, chip_index(chip_idx, c.len, mode));
        ec = chip_at(chips, chip_index(chip_idx + c.el, c.len, mode));
        lc = chip_at(chips, chip_index(chip_idx - c.el, c.len, mode));
    }
    if (boc) {   // BOC(1,1): sub-carrier sign = +1 on the first half chip, -1 on the second
        const float a = chip_idx, b = chip_idx + c.el, e = chip_idx - c.el;
        pc = (a - floorf(a)) < 0.5f ? pc : -pc;
        ec = (b - floorf(b)) < 0.5f ? ec : -ec;
        lc = (e - floorf(e)) < 0.5f ? lc : -lc;
    }
    // chips are exactly +-1 (x BOC sign): the product is exact, so fma(x, chip, acc) == acc + x*chip bitwise
    acc[0] = __builtin_fmaf(xr, pc, acc[0]); acc[1] = __builtin_fmaf(xi, pc, acc[1]);
    acc[2] = __builtin_fmaf(xr, ec, acc[2]); acc[3] = __builtin_fmaf(xi, ec, acc[3]);
    acc[4] = __builtin_fmaf(xr, lc, acc[4]); acc[5] = __builtin_fmaf(xi, lc, acc[5]);
    if constexpr (ARMS == 5) {
        const float ve = chip_idx + c.vel, vl = chip_idx - c.vel;
        float vec = chip_at(chips, FAST ? chip_index_arm(ve, c.len, mode) : chip_index(ve, c.len, mode));
        float vlc = chip_at(chips, FAST ? chip_index_arm(vl, c.len, mode) : chip_index(vl, c.len, mode));
        if (boc) {
            vec = (ve - floorf(ve)) < 0.5f ? vec : -vec;
            vlc = (vl - floorf(vl)) < 0.5f ? vlc : -vlc;
        }
        acc[6] = __builtin_fmaf(xr, vec, acc[6]); acc[7] = __builtin_fmaf(xi, vec, acc[7]);
        acc[8] = __builtin_fmaf(xr, vlc, acc[8]); acc[9] = __builtin_fmaf(xi, vlc, acc[9]);
    }
}

// ---- the persistent kernel's sample, for epochs that passed fast_car_ok / fast_code_ok.  Same arithmetic as
// correlate_sample<.., FAST = true> on the values that reach the sums, with the instruction count of the hot loop cut
// (it is bound by VALU issue slots: ~115 -> ~80 issue units per sample):
//  * the chip row is a float table padded by one entry at each end — tab[0] is what floor = -1 reads (chip 0 in
//    FAITHFUL mode, the last chip in FIXED mode), tab[len + 1] = chip 0 is what floor = len reads (`% len`) — so an arm
//    is floor -> load, with no selects and no int -> float conversion;
//  * the code phase's conditional subtractions are one unsigned minimum: t >= 0, so of {t, t - len, t - 2 len} the
//    non-negative ones order like their bit patterns and the negative ones (sign bit set) compare above all of them —
//    the minimum is the reduced phase, the same value fmod_code<true> selects;
//  * the sample index arrives as a float (exact: n < 2^24 is part of fast_code_ok), formed by additions;
//  * sin/cos by gm_libm.h's sincos_cw (f32 Cody-Waite reduction; |phase| < 1e5 rad is part of fast_car_ok): the same
//    values as the f64-reduced form on 99.999 % of arguments, the same polynomial cores.
__device__ __forceinline__ int floor_i32(float x) {      // int(floorf(x)) in one instruction
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
template <int ARMS, int BOC_T>
__device__ __forceinline__ void correlate_sample_fast(const EpochConsts& c, const float* tab, cf d, float fi, float (&acc)[2 * ARMS]) {
    const float w = c.two_pi_f * fi;
    const float phase = c.carrier_phase + div_by_fs(w, c.fs, c.inv_fs);
    float sn, cs;
    sincos_cw(phase, sn, cs);
    const float wc = cs, ws = -sn;                          // Complex32::new(cos_p, -sin)
    const float xr = d.x * wc - d.y * ws;                   // num-complex Mul
    const float xi = d.x * ws + d.y * wc;
    const float t = c.code_phase + fi * c.step;
    const uint32_t ua = __float_as_uint(t), ub = __float_as_uint(t - c.lenf), uc = __float_as_uint(t - 2.0f * c.lenf);
    const float chip_idx = __uint_as_float(min(min(ua, ub), uc));
    const float ea = chip_idx + c.el, la = chip_idx - c.el;
    float pc = tab[floor_i32(chip_idx) + 1], ec = tab[floor_i32(ea) + 1], lc = tab[floor_i32(la) + 1];
    if (BOC_T) {   // BOC(1,1): sub-carrier sign = +1 on the first half chip, -1 on the second (fract = x - floor(x))
        pc = __builtin_amdgcn_fractf(chip_idx) < 0.5f ? pc : -pc;
        ec = __builtin_amdgcn_fractf(ea) < 0.5f ? ec : -ec;
        lc = __builtin_amdgcn_fractf(la) < 0.5f ? lc : -lc;
    }
    acc[0] = __builtin_fmaf(xr, pc, acc[0]); acc[1] = __builtin_fmaf(xi, pc, acc[1]);
    acc[2] = __builtin_fmaf(xr, ec, acc[2]); acc[3] = __builtin_fmaf(xi, ec, acc[3]);
    acc[4] = __builtin_fmaf(xr, lc, acc[4]); acc[5] = __builtin_fmaf(xi, lc, acc[5]);
    if constexpr (ARMS == 5) {
        const float ve = chip_idx + c.vel, vl = chip_idx - c.vel;
        float vec = tab[floor_i32(ve) + 1], vlc = tab[floor_i32(vl) + 1];
        if (BOC_T) {
            vec = __builtin_amdgcn_fractf(ve) < 0.5f ? vec : -vec;
            vlc = __builtin_amdgcn_fractf(vl) < 0.5f ? vlc : -vlc;
        }
        acc[6] = __builtin_fmaf(xr, vec, acc[6]); acc[7] = __builtin_fmaf(xi, vec, acc[7]);
        acc[8] = __builtin_fmaf(xr, vlc, acc[8]); acc[9] = __builtin_fmaf(xi, vlc, acc[9]);
    }
}

// NB samples of one lane at once, written "structure of arrays": every step of correlate_sample_fast is applied to all NB
// samples before the next step, in three stages fenced against the instruction scheduler — (1) code phases -> chip
// addresses -> ALL the LDS look-ups issued; (2) the carrier phases and their sin/cos (gm_libm.h's sincos_cw, the same
// operations in the same order per sample), whose ~40 dependent instructions per sample now interleave NB ways and cover
// the look-ups' latency; (3) products into the sums.  Calling correlate_sample_fast NB times in a row left it to the
// scheduler, which kept the samples one after another (least registers): a wave then issued one DEPENDENT instruction per
// ~8 cycles and sat out an LDS round trip per sample.  Values are bit-identical to the one-sample form.
// BOC(1,1) (BOC_T): `tab` is the HALF-CHIP table (trk_persistent_kernel: entry j + 2 = chip floor(j / 2) x the sub-carrier sign of
// half chip j, j = -2 .. 2 len + 1).  An arm's phase p = fl(chip_idx + s) is looked up at j = floor(2 p), and 2 p = fl(2 chip_idx
// + 2 s) EXACTLY — binary rounding commutes with the doubling — so one fused multiply-add gives the index of the chip AND of its
// sub-carrier half: floor(p) = floor(j / 2), fract(p) < 0.5 <=> j even.  Same values as the chip look-up + fract / compare / select
// of the plain table, bit for bit (the products are exact either way), at 4 instructions per arm instead of 8 (round 6: the
// five-arm BOC geometry of BASELINE configs[4] is throughput-shaped, 200 000 samples per channel-epoch).
template <int ARMS, int BOC_T, int NB>
__device__ __forceinline__ void correlate_block_fast(const EpochConsts& c, const float* tab, const cf (&d)[NB], const float (&f)[NB],
                                                     float (&acc0)[2 * ARMS], float (&acc1)[2 * ARMS]) {
    // ---- stage 1: chips
    float chip_idx[NB], pc[NB], ec[NB], lc[NB], vec[NB], vlc[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const float t = c.code_phase + f[k] * c.step;
        const uint32_t ua = __float_as_uint(t), ub = __float_as_uint(t - c.lenf), uc = __float_as_uint(t - 2.0f * c.lenf);
        chip_idx[k] = __uint_as_float(min(min(ua, ub), uc));
    }
    if constexpr (BOC_T == 1) {
        const float el2 = c.el + c.el, vel2 = c.vel + c.vel;          // exact doublings
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            pc[k] = tab[floor_i32(chip_idx[k] + chip_idx[k]) + 2];
            ec[k] = tab[floor_i32(__builtin_fmaf(chip_idx[k], 2.0f, el2)) + 2];
            lc[k] = tab[floor_i32(__builtin_fmaf(chip_idx[k], 2.0f, -el2)) + 2];
            if constexpr (ARMS == 5) {
                vec[k] = tab[floor_i32(__builtin_fmaf(chip_idx[k], 2.0f, vel2)) + 2];
                vlc[k] = tab[floor_i32(__builtin_fmaf(chip_idx[k], 2.0f, -vel2)) + 2];
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            pc[k] = tab[floor_i32(chip_idx[k]) + 1];
            ec[k] = tab[floor_i32(chip_idx[k] + c.el) + 1];
            lc[k] = tab[floor_i32(chip_idx[k] - c.el) + 1];
            if constexpr (ARMS == 5) {
                vec[k] = tab[floor_i32(chip_idx[k] + c.vel) + 1];
                vlc[k] = tab[floor_i32(chip_idx[k] - c.vel) + 1];
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- stage 2: carrier wipe-off.  phase = carrier_phase + ((2 pi f) * i) / fs (:233); sincos_cw step by step
    float r[NB], kq[NB], xr[NB], xi[NB];
    {
        const float c1 = f32_from_bits(0x3fc90fdbu), c2 = f32_from_bits(0xb33bbd2eu);
        float w[NB], q0[NB], ph[NB], pl[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) w[k] = c.two_pi_f * f[k];
#pragma unroll
        for (int k = 0; k < NB; ++k) q0[k] = w[k] * c.inv_fs;
#pragma unroll
        for (int k = 0; k < NB; ++k) w[k] = __builtin_fmaf(-q0[k], c.fs, w[k]);
#pragma unroll
        for (int k = 0; k < NB; ++k) q0[k] = __builtin_fmaf(w[k], c.inv_fs, q0[k]);       // div_by_fs
#pragma unroll
        for (int k = 0; k < NB; ++k) w[k] = c.carrier_phase + q0[k];                        // the phase x
#pragma unroll
        for (int k = 0; k < NB; ++k) kq[k] = __builtin_rintf(w[k] * 0.636619747f);
        if constexpr (ARMS == 3) {      // the form the 3 us latency chain of 32 channels x 25 Msps was tuned with: the one below costs
#pragma unroll                          // it 45 ns per epoch (DESIGN_HISTORY R6.6)
            for (int k = 0; k < NB; ++k) ph[k] = kq[k] * c1;
#pragma unroll
            for (int k = 0; k < NB; ++k) pl[k] = __builtin_fmaf(kq[k], c1, -ph[k]);
#pragma unroll
            for (int k = 0; k < NB; ++k) w[k] = w[k] - ph[k];
        } else {
#pragma unroll
            for (int k = 0; k < NB; ++k) ph[k] = kq[k] * -c1;                               // -(k c1), exactly (held negated: the fused
#pragma unroll                                                                              // multiply-add then takes c1 as a literal, not from
            for (int k = 0; k < NB; ++k) pl[k] = __builtin_fmaf(kq[k], c1, ph[k]);          // a scalar register: half the issue cost)
#pragma unroll
            for (int k = 0; k < NB; ++k) w[k] = w[k] + ph[k];
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) pl[k] = __builtin_fmaf(kq[k], c2, pl[k]);
#pragma unroll
        for (int k = 0; k < NB; ++k) r[k] = w[k] - pl[k];
    }
    {
        float z[NB], sp[NB], cp[NB], sr[NB], cr[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) z[k] = r[k] * r[k];
#pragma unroll
        for (int k = 0; k < NB; ++k) { sp[k] = __builtin_fmaf(-1.9515295891e-4f, z[k], 8.3321608736e-3f); cp[k] = __builtin_fmaf(2.443315711809948e-5f, z[k], -1.388731625493765e-3f); }
#pragma unroll
        for (int k = 0; k < NB; ++k) { sp[k] = __builtin_fmaf(sp[k], z[k], -1.6666654611e-1f); cp[k] = __builtin_fmaf(cp[k], z[k], 4.166664568298827e-2f); }
#pragma unroll
        for (int k = 0; k < NB; ++k) { sp[k] = sp[k] * z[k]; cp[k] = cp[k] * z[k]; cr[k] = __builtin_fmaf(-0.5f, z[k], 1.0f); }
#pragma unroll
        for (int k = 0; k < NB; ++k) { sr[k] = __builtin_fmaf(sp[k], r[k], r[k]); cr[k] = __builtin_fmaf(cp[k], z[k], cr[k]); }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const uint32_t q = uint32_t(int(kq[k]));
            const float sv = (q & 1u) ? cr[k] : sr[k], cv = (q & 1u) ? sr[k] : cr[k];
            const float sn = __uint_as_float(__float_as_uint(sv) ^ ((q << 30) & 0x80000000u));
            const float cs = __uint_as_float(__float_as_uint(cv) ^ (((q + 1u) << 30) & 0x80000000u));
            const float wc = cs, ws = -sn;                          // Complex32::new(cos_p, -sin)
            xr[k] = d[k].x * wc - d[k].y * ws;                      // num-complex Mul
            xi[k] = d[k].x * ws + d[k].y * wc;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- stage 3: sums (chips are exactly +-1 x BOC sign: the product is exact, fma(x, chip, acc) == acc + x*chip bitwise)
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        if (BOC_T != 0 && BOC_T != 1) {   // (run-time BOC flag: not instantiated by the persistent kernel; the half-chip table carries the sign)
            pc[k] = __builtin_amdgcn_fractf(chip_idx[k]) < 0.5f ? pc[k] : -pc[k];
            ec[k] = __builtin_amdgcn_fractf(chip_idx[k] + c.el) < 0.5f ? ec[k] : -ec[k];
            lc[k] = __builtin_amdgcn_fractf(chip_idx[k] - c.el) < 0.5f ? lc[k] : -lc[k];
            if constexpr (ARMS == 5) {
                vec[k] = __builtin_amdgcn_fractf(chip_idx[k] + c.vel) < 0.5f ? vec[k] : -vec[k];
                vlc[k] = __builtin_amdgcn_fractf(chip_idx[k] - c.vel) < 0.5f ? vlc[k] : -vlc[k];
            }
        }
        float (&acc)[2 * ARMS] = (k & 1) ? acc1 : acc0;      // even samples of the block into one set of sums, odd into the other
        acc[0] = __builtin_fmaf(xr[k], pc[k], acc[0]); acc[1] = __builtin_fmaf(xi[k], pc[k], acc[1]);
        acc[2] = __builtin_fmaf(xr[k], ec[k], acc[2]); acc[3] = __builtin_fmaf(xi[k], ec[k], acc[3]);
        acc[4] = __builtin_fmaf(xr[k], lc[k], acc[4]); acc[5] = __builtin_fmaf(xi[k], lc[k], acc[5]);
        if constexpr (ARMS == 5) {
            acc[6] = __builtin_fmaf(xr[k], vec[k], acc[6]); acc[7] = __builtin_fmaf(xi[k], vec[k], acc[7]);
            acc[8] = __builtin_fmaf(xr[k], vlc[k], acc[8]); acc[9] = __builtin_fmaf(xi[k], vlc[k], acc[9]);
        }
    }
}

__device__ __forceinline__ int code_row(const TrkDevCfg& cfg, const gm_trk_state& st) {
    // FAITHFUL indexes GPS_CA_CODE_32_PRN[prn] (:276), FIXED [prn-1]
    return cfg.gps_ca ? (cfg.code_index_mode == GM_CODE_INDEX_FAITHFUL ? int(st.prn) : int(st.prn) - 1)
                      : int(st.prn) - 1;
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
    if (!src.linear) n = samples_per_code(cfg.fs, st.code_rate, cfg.code_len_f);
    const int row = code_row(cfg, st);
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

    const EpochConsts ec = epoch_consts(cfg, st);
    // slice bounds: whole multiples of 256 samples so lanes stay coalesced
    const uint32_t per = uint32_t(((n + slices - 1) / slices + 255) / 256 * 256);
    const uint32_t i0 = uint32_t(slice) * per;
    const uint32_t i1 = (uint64_t(i0) + per < n) ? i0 + per : uint32_t(n);

    float acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.0f;
    const uint64_t base = src.linear ? 0 : st.next_sample_index;
    if (fast_code_range(ec, n)) {
        for (uint32_t i = i0 + tid; i < i1; i += 256)
            correlate_sample<ARMS, true>(ec, chips, src.base[(base + i) & src.mask], i, acc);
    } else {
        for (uint32_t i = i0 + tid; i < i1; i += 256)
            correlate_sample<ARMS, false>(ec, chips, src.base[(base + i) & src.mask], i, acc);
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


// ---- gm_trk_cfg.strict_sum_order: the correlator sums in the reference's own order ------------------------------------
// early_late_correlation adds the products of sample 0, 1, 2, ... into each of its six sums one after the other in f32
// (`i_p += re * p_chip`, do_tracking.rs:256-262); a parallel sum is a different (and more accurate) rounding sequence,
// ~7e-6 of the envelope away at n = 25 000 — the whole of what separates this library's tracking from the reference's once
// the carrier's cos / sin are glibc's (strict_libm).  With the switch on an epoch is three launches instead of the
// persistent kernel's loop: trk_terms_kernel — every sample's products (the same correlate_sample as everywhere else, on a
// zeroed accumulator: x * (+-1) is exact), stored as one contiguous stream per sum; trk_serial_sum_kernel — ONE wave per
// channel, lane k walks stream k from sample 0 to n - 1 (n dependent f32 adds, ~10 cycles each: ~110 us at n = 25 000, all
// channels side by side), the streams staged through LDS by the workgroup's other three waves; trk_update_kernel — the
// scalar epilogue as before.  The sums, and with strict_libm every word of the channel state, then equal the reference's
// bit for bit, free-running, for as many epochs as one likes (tests/test_gpu_tracking_shapes.py).  ~40x the persistent
// kernel's time per epoch at BASELINE configs[2]: a parity switch, not the production path.
template <int ARMS>
__global__ __launch_bounds__(256) void trk_terms_kernel(TrkDevCfg cfg, const int8_t* __restrict__ codes,
                                                        const gm_trk_state* __restrict__ states, TrkSrc src, int slices,
                                                        float* __restrict__ terms, unsigned long long cap,
                                                        uint8_t* __restrict__ ready, int* __restrict__ error_flag) {
    constexpr int NV = 2 * ARMS;
    const int ch = src.only_channel >= 0 ? src.only_channel : int(blockIdx.y);
    const int slice = blockIdx.x, tid = threadIdx.x;
    const gm_trk_state st = states[ch];
    uint64_t n = st.num_samples_per_code;                                  // as in trk_correlate_kernel (:165-166, :232)
    if (!src.linear) n = samples_per_code(cfg.fs, st.code_rate, cfg.code_len_f);
    const int row = code_row(cfg, st);
    bool run = st.active && n > 0 && n < (1ull << 31) && row >= 0 && row < cfg.n_codes;
    if (run && !src.linear) run = (int64_t)(src.head - (st.next_sample_index + n)) >= 0;   // (:170-172)
    if (run && n > cap) {            // the code rate has left the +1 % the streams were sized for: reported, never skipped silently
        if (slice == 0 && tid == 0) *error_flag = 2;
        run = false;
    }
    if (!run) {
        if (slice == 0 && tid == 0) ready[ch] = 0;
        return;
    }
    extern __shared__ int8_t chips[];
    const int8_t* crow = codes + size_t(row) * cfg.code_len;
    for (int i = tid; i < cfg.code_len; i += 256) chips[i] = crow[i];
    __syncthreads();
    const EpochConsts ec = epoch_consts(cfg, st);
    const uint32_t per = uint32_t(((n + slices - 1) / slices + 255) / 256 * 256);
    const uint32_t i0 = uint32_t(slice) * per;
    const uint32_t i1 = (uint64_t(i0) + per < n) ? i0 + per : uint32_t(n);
    const uint64_t base = src.linear ? 0 : st.next_sample_index;
    float* out = terms + size_t(ch) * NV * cap;
    const bool fast = fast_code_range(ec, n);
    for (uint32_t i = i0 + tid; i < i1; i += 256) {
        float acc[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) acc[k] = 0.0f;
        if (fast) correlate_sample<ARMS, true>(ec, chips, src.base[(base + i) & src.mask], i, acc);
        else correlate_sample<ARMS, false>(ec, chips, src.base[(base + i) & src.mask], i, acc);
#pragma unroll
        for (int k = 0; k < NV; ++k) out[size_t(k) * cap + i] = acc[k];
    }
    if (slice == 0 && tid == 0) ready[ch] = 1;
}

template <int ARMS> constexpr int trk_serial_chunk() { return ARMS == 5 ? 512 : 1024; }   // samples per LDS stage (x NV streams x 2 buffers: 48 / 40 KB)
template <int ARMS>
__global__ __launch_bounds__(256) void trk_serial_sum_kernel(TrkDevCfg cfg, const gm_trk_state* __restrict__ states, TrkSrc src,
                                                             const float* __restrict__ terms, unsigned long long cap,
                                                             const uint8_t* __restrict__ ready, float* __restrict__ partials) {
    constexpr int NV = 2 * ARMS, CH = trk_serial_chunk<ARMS>();
    const int ch = src.only_channel >= 0 ? src.only_channel : int(blockIdx.x);
    const int tid = threadIdx.x;
    float* pout = partials + size_t(ch) * NV;                              // one "slice" per channel
    if (!ready[ch]) {
        if (tid < NV) pout[tid] = 0.0f;
        return;
    }
    const gm_trk_state st = states[ch];
    uint64_t n64 = st.num_samples_per_code;
    if (!src.linear) n64 = samples_per_code(cfg.fs, st.code_rate, cfg.code_len_f);
    const uint32_t n = uint32_t(n64);
    __shared__ float4 stage4[2 * NV * CH / 4];                             // [2][NV][CH] floats
    const float* in = terms + size_t(ch) * NV * cap;                       // cap is a multiple of 4 (trk_reserve_terms): rows 16-byte aligned
    // chunk c -> buffer c & 1, by the NT threads t0 .. t0 + NT - 1: 16-byte loads, all of a thread's loads in flight before its
    // first LDS store (the element-by-element form waited for every load by itself and was what bounded the kernel).  Words
    // beyond the epoch's n inside the last 16 bytes are never summed.
    auto fetch = [&](uint32_t c, int t0, auto nt_tag) {
        constexpr int NT = decltype(nt_tag)::value, PER = (NV * CH / 4 + NT - 1) / NT;
        float4* dst = stage4 + size_t(c & 1u) * (NV * CH / 4);
        const uint32_t s0 = c * CH, cnt = n - s0 < uint32_t(CH) ? n - s0 : uint32_t(CH);
        const uint32_t q4 = (cnt + 3) / 4;                                 // 16-byte groups per stream in this chunk
        float4 v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const uint32_t f = uint32_t(tid - t0) + uint32_t(u) * NT, k = f / (CH / 4), j4 = f % (CH / 4);
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < uint32_t(NV * CH / 4) && j4 < q4) v[u] = *reinterpret_cast<const float4*>(in + size_t(k) * cap + s0 + 4 * j4);
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const uint32_t f = uint32_t(tid - t0) + uint32_t(u) * NT;
            if (f < uint32_t(NV * CH / 4)) dst[f] = v[u];
        }
    };
    const uint32_t chunks = (n + CH - 1) / CH;
    fetch(0, 0, std::integral_constant<int, 256>());
    __syncthreads();
    float acc = 0.0f;                                                      // let mut i_p = 0.0_f32 (:244-249)
    for (uint32_t c = 0; c < chunks; ++c) {
        if (tid >= 64) { if (c + 1 < chunks) fetch(c + 1, 64, std::integral_constant<int, 192>()); }   // waves 1-3 stage the next chunk
        else if (tid < NV) {                                               // lane k: sum k, sample by sample
            const float4* srcp = stage4 + size_t(c & 1u) * (NV * CH / 4) + tid * (CH / 4);
            const uint32_t s0 = c * CH, cnt = n - s0 < uint32_t(CH) ? n - s0 : uint32_t(CH);
            // batches of 16 samples in TWO named register sets: while the 16 dependent adds of one set issue, the other set's
            // four LDS reads are in flight (written as a copy `b = a` the compiler turned the hand-over into 16 v_mov per
            // batch and waited for every read right behind its issue: 8.5 ns per add)
            const uint32_t full = cnt / 16;
#define GM_ADD16(q0, q1, q2, q3)                                                                     \
            acc = acc + q0.x; acc = acc + q0.y; acc = acc + q0.z; acc = acc + q0.w;                  \
            acc = acc + q1.x; acc = acc + q1.y; acc = acc + q1.z; acc = acc + q1.w;                  \
            acc = acc + q2.x; acc = acc + q2.y; acc = acc + q2.z; acc = acc + q2.w;                  \
            acc = acc + q3.x; acc = acc + q3.y; acc = acc + q3.z; acc = acc + q3.w      /* i_p += re * p_chip, in sample order */
            float4 a0, a1, a2, a3, b0, b1, b2, b3;
            if (full) { a0 = srcp[0]; a1 = srcp[1]; a2 = srcp[2]; a3 = srcp[3]; }
            uint32_t bt = 0;
            for (; bt + 2 <= full; bt += 2) {
                b0 = srcp[4 * bt + 4]; b1 = srcp[4 * bt + 5]; b2 = srcp[4 * bt + 6]; b3 = srcp[4 * bt + 7];
                __builtin_amdgcn_sched_barrier(0);                      // the reads stay IN FRONT of the other set's adds
                GM_ADD16(a0, a1, a2, a3);
                __builtin_amdgcn_sched_barrier(0);
                const uint32_t nx = bt + 2 < full ? bt + 2 : bt;       // unconditional (a re-read of batch bt past the end, unused): a
                a0 = srcp[4 * nx]; a1 = srcp[4 * nx + 1]; a2 = srcp[4 * nx + 2]; a3 = srcp[4 * nx + 3];   // conditional load made the
                __builtin_amdgcn_sched_barrier(0);                      // compiler merge the two sets with 32 v_mov per round
                GM_ADD16(b0, b1, b2, b3);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (bt < full) { GM_ADD16(a0, a1, a2, a3); }             // an odd batch is left in the first set
#undef GM_ADD16
            const float* tail = reinterpret_cast<const float*>(srcp);
            for (uint32_t j = full * 16; j < cnt; ++j) acc = acc + tail[j];
        }
        __syncthreads();
    }
    if (tid < NV) pout[tid] = acc;
}

// The reference's scalar epilogue of one epoch: early_late_correlation's phase advances (:240-242,
// :265-270) and, in DO_WORK mode, do_work + run_loop_filters (:183-210, :279-302).
// v = the 2*ARMS correlator sums; n = samples of this epoch.
// The scalar work between two epochs, in two independent halves so that two waves can run them side by side in the
// persistent kernel (a lone wave issues about one instruction per five cycles whatever their dependencies: the section's
// length is its instruction count).  Each half reads the OLD state and produces only the fields it owns:
//   carrier half: prn, active, lost_counter, next_sample_index, carrier_{freq,phase,error,nco}, i/q_prompt, messages
//   code half   : code_{phase,error,nco,rate}, num_samples_per_code
// Both evaluate the lock test and the give-up rule (:183-210) from the same inputs, so they agree.
// carrier_phase = (carrier_phase + 2*PI*carrier_freq*(n as f32 / fs)) % (2*PI)          (:240-242); two_pi_f = (2*PI)*carrier_freq
__device__ __forceinline__ float advance_carrier_phase(const TrkDevCfg& cfg, float phase, float two_pi_f, float nf) {
    return fmod_bounded(phase + two_pi_f * (cfg.div_fs_ok ? div_const(nf, cfg.fs, cfg.inv_fs) : __fdiv_rn(nf, cfg.fs)),
                        2.0f * GM_PI_F, cfg.inv_2pi);
}
// code_phase = (code_phase + (code_rate/fs) * n as f32) % len                               (:265-267); step = code_rate/fs
__device__ __forceinline__ float advance_code_phase(const TrkDevCfg& cfg, float phase, float step, float nf) {
    return fmod_bounded(phase + step * nf, cfg.code_len_f, cfg.inv_len);
}

struct CarrierHalf {
    uint8_t prn, active, lst, lprn;
    uint32_t lost_counter;
    uint64_t next_sample_index;
    float carrier_freq, carrier_phase, carrier_error, carrier_nco, i_prompt, q_prompt;
};
struct CodeHalf {
    uint64_t num_samples_per_code;
    float code_phase, code_error, code_nco, code_rate;
};
__device__ __forceinline__ bool trk_locked(const TrkDevCfg& cfg, float ip, float qp) { return ip * ip + qp * qp > cfg.lock_threshold; }
__device__ __forceinline__ bool trk_give_up(const TrkDevCfg& cfg, const gm_trk_state& s, bool locked) {
    return !locked && s.lost_counter + 1u >= cfg.max_lost_epochs;
}

// pre_phase: the advanced carrier phase if the caller has it already (it depends on the old state and n only), else null
template <int ARMS>
__device__ __forceinline__ CarrierHalf carrier_half(const TrkDevCfg& cfg, const gm_trk_state& s, const float (&v)[2 * ARMS],
                                                    uint64_t n, int mode, const float* pre_phase = nullptr) {
    CarrierHalf o;
    o.prn = s.prn; o.active = s.active; o.lst = 0; o.lprn = 0; o.lost_counter = s.lost_counter;
    o.next_sample_index = s.next_sample_index;
    o.carrier_freq = s.carrier_freq; o.carrier_error = s.carrier_error; o.carrier_nco = s.carrier_nco;
    const float nf = float(n);
    // carrier_phase = (carrier_phase + 2*PI*carrier_freq*(n as f32 / fs)) % (2*PI)      (:240-242)
    o.carrier_phase = pre_phase ? *pre_phase : advance_carrier_phase(cfg, s.carrier_phase, 2.0f * GM_PI_F * s.carrier_freq, nf);
    o.i_prompt = v[0]; o.q_prompt = v[1];
    if (mode != TRK_MODE_DO_WORK) return o;
    const bool locked = trk_locked(cfg, v[0], v[1]);     // do_work (:183-210)
    if (locked) {
        o.lost_counter = 0;
        // run_loop_filters (:279-302), carrier part
        // f32::atan = the host libm's atanf (gm_libm.h); / (2*PI): constant divisor, correctly rounded through its reciprocal
        // (Markstein form, equal to IEEE division on 8e6 arguments spanning atan's range — the division was ~12 dependent
        // instructions of the serial section)
        const float pll_err = div_const(atanf_glibc(__fdiv_rn(v[1], v[0])), 2.0f * GM_PI_F, cfg.inv_2pi);
        o.carrier_nco = loop_filter_update(cfg.pll_dt_tau1, cfg.pll_tau2_tau1, pll_err, s.carrier_error);
        o.carrier_error = pll_err;
        o.carrier_freq = s.carrier_freq + o.carrier_nco;
    } else if (trk_give_up(cfg, s, locked)) {
        // reset() first, so the message carries prn 0 (:199-201)
        o.prn = 0; o.active = 0; o.lost_counter = 0; o.next_sample_index = 0;
        o.carrier_freq = 0.f; o.carrier_phase = 0.f; o.carrier_error = 0.f; o.carrier_nco = 0.f;
        o.i_prompt = 0.f; o.q_prompt = 0.f;
        o.lst = 1; o.lprn = 0;
        return o;
    } else {
        o.lost_counter = s.lost_counter + 1u;
    }
    o.next_sample_index = s.next_sample_index + n;       // (:192 / :203)
    return o;
}

// rate_lo / rate_hi: an interval of code rates proven to keep the epoch length at n_stored (gm_libm.h spc_rate_bounds), or an
// empty one (lo > hi): inside it the two divisions and the rounding of the definition are skipped
template <int ARMS>
__device__ __forceinline__ CodeHalf code_half(const TrkDevCfg& cfg, const gm_trk_state& s, const float (&v)[2 * ARMS], uint64_t n,
                                              uint64_t n_stored, int mode, const float* pre_phase = nullptr,
                                              float rate_lo = 1.0f, float rate_hi = 0.0f) {
    CodeHalf o;
    o.num_samples_per_code = n_stored; o.code_error = s.code_error; o.code_nco = s.code_nco; o.code_rate = s.code_rate;
    const float nf = float(n);
    // code_phase = (code_phase + (code_rate/fs) * n as f32) % 1023.0                     (:265-267)
    o.code_phase = pre_phase ? *pre_phase
                             : advance_code_phase(cfg, s.code_phase,
                                                  cfg.div_fs_ok ? div_const(s.code_rate, cfg.fs, cfg.inv_fs) : __fdiv_rn(s.code_rate, cfg.fs), nf);
    if (mode != TRK_MODE_DO_WORK) return o;
    const bool locked = trk_locked(cfg, v[0], v[1]);
    if (locked) {                                        // run_loop_filters (:279-302), code part
        const float pow_e = sqrt_rn(v[2] * v[2] + v[3] * v[3]);     // f32::sqrt is correctly rounded (gm_libm.h)
        const float pow_l = sqrt_rn(v[4] * v[4] + v[5] * v[5]);
        const float dll_err = ((pow_e + pow_l) != 0.0f) ? __fdiv_rn(pow_e - pow_l, pow_e + pow_l) : 0.0f;
        o.code_nco = loop_filter_update(cfg.dll_dt_tau1, cfg.dll_tau2_tau1, dll_err, s.code_error);
        o.code_error = dll_err;
        o.code_rate = s.code_rate + o.code_nco;
    } else if (trk_give_up(cfg, s, locked)) {            // reset(): num_samples_per_code keeps the stored length
        o.code_phase = 0.f; o.code_error = 0.f; o.code_nco = 0.f; o.code_rate = 0.f;
        return o;
    }
    if (o.code_rate >= rate_lo && o.code_rate <= rate_hi) o.num_samples_per_code = n_stored;     // (false for NaN and for the empty interval)
    else o.num_samples_per_code = samples_per_code(cfg, o.code_rate);
    return o;
}

// The bookkeeping both halves agree on (do_work :183-210): the wave that runs the code half in the persistent kernel keeps
// its own copy of these fields; same rules as carrier_half, which owns them in the state that is stored.
struct CommonHalf { uint8_t prn, active; uint32_t lost_counter; uint64_t next_sample_index; };
__device__ __forceinline__ CommonHalf common_half(const TrkDevCfg& cfg, const gm_trk_state& s, float ip, float qp, uint64_t n, int mode) {
    CommonHalf o;
    o.prn = s.prn; o.active = s.active; o.lost_counter = s.lost_counter; o.next_sample_index = s.next_sample_index + n;
    if (mode != TRK_MODE_DO_WORK) return o;
    const bool locked = trk_locked(cfg, ip, qp);
    if (locked) o.lost_counter = 0;
    else if (trk_give_up(cfg, s, locked)) { o.prn = 0; o.active = 0; o.lost_counter = 0; o.next_sample_index = 0; }
    else o.lost_counter = s.lost_counter + 1u;
    return o;
}

// Per-epoch scalar state update of one channel: phase advances of early_late_correlation (:240-242, :265-270) and, in
// DO_WORK mode, do_work + run_loop_filters (:183-210, :279-302).  v = the 2*ARMS correlator sums; n = samples of this
// epoch; s.num_samples_per_code already holds the length update() stored (:166).
template <int ARMS>
__device__ __forceinline__ void epoch_epilogue(const TrkDevCfg& cfg, gm_trk_state& s, const float (&v)[2 * ARMS],
                                               uint64_t n, int mode, uint8_t& lst, uint8_t& lprn) {
    const CarrierHalf a = carrier_half<ARMS>(cfg, s, v, n, mode);
    const CodeHalf b = code_half<ARMS>(cfg, s, v, n, s.num_samples_per_code, mode);
    s.prn = a.prn; s.active = a.active; s.lost_counter = a.lost_counter; s.next_sample_index = a.next_sample_index;
    s.carrier_freq = a.carrier_freq; s.carrier_phase = a.carrier_phase; s.carrier_error = a.carrier_error;
    s.carrier_nco = a.carrier_nco; s.i_prompt = a.i_prompt; s.q_prompt = a.q_prompt;
    s.num_samples_per_code = b.num_samples_per_code; s.code_phase = b.code_phase; s.code_error = b.code_error;
    s.code_nco = b.code_nco; s.code_rate = b.code_rate;
    lst = a.lst; lprn = a.lprn;
}

template <int ARMS> __device__ __forceinline__ gm_trk_out make_out(const float (&v)[2 * ARMS]) {
    gm_trk_out o;
    o.ip = v[0]; o.qp = v[1]; o.ie = v[2]; o.qe = v[3]; o.il = v[4]; o.ql = v[5];
    o.ive = o.qve = o.ivl = o.qvl = 0.0f;
    if constexpr (ARMS == 5) { o.ive = v[6]; o.qve = v[7]; o.ivl = v[8]; o.qvl = v[9]; }
    return o;
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
    float v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = 0.0f;
    uint8_t did = 0, lst = 0, lprn = 0;
    if (ready[ch]) {
        gm_trk_state s = states[ch];
        const float* p = partials + size_t(ch) * slices * NV;
        for (int sl = 0; sl < slices; ++sl)
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] += p[sl * NV + k];
        uint64_t n = s.num_samples_per_code;
        if (!linear) { n = samples_per_code(cfg.fs, s.code_rate, cfg.code_len_f); s.num_samples_per_code = n; }  // (:166)
        epoch_epilogue<ARMS>(cfg, s, v, n, mode, lst, lprn);
        did = 1;
        states[ch] = s;
    }
    if (outs) outs[ch] = make_out<ARMS>(v);
    if (processed) processed[ch] = did;
    if (lost) lost[ch] = lst;
    if (lost_prn) lost_prn[ch] = lprn;
}

// ------------------------------------------------------------------------------------ persistent tracking
// One launch runs `epochs` consecutive passes of process_channels (do_tracking.rs:364-371, 408-413).
// Grid: ceil(n_channels / 8) * 8 channel slots x G workgroups of TRK_PERSIST_THREADS lanes, all co-resident (two per CU);
// the G workgroups of a channel sit on one XCD.  Each correlates one slice of every code period; per epoch they exchange
// their partial sums through self-validating 8-byte {value, tag} granules (write-through stores and L1-bypassing loads
// in a launch's first epoch — MI355X_MICROARCH.md "Valid forms", R2 — and plain stores that stay in the XCD's L2 once the
// workgroups have confirmed from XCC_ID that they share it), every workgroup adds the G partials in the same order and
// runs the scalar update redundantly (carrier half on wave 0, code half on wave 1), so all of them hold bit-identical
// channel state without a broadcast.  The epoch-to-epoch dependence (carrier_freq/phase, code_rate/phase,
// next_sample_index) never leaves the chip.  DESIGN.md 4.3 walks through one epoch with its measured phases.
struct TrkPersistArgs {
    TrkDevCfg cfg;
    const int8_t* codes;
    gm_trk_state* states;
    const cf* ring; uint64_t mask, head;
    int G, epochs;
    int packed;                      // 0: channel slots x G (a channel's workgroups on one XCD); else the packed layout's workgroups per XCD
    int GS;                          // granules per arm in the exchange block: 16 for G <= 16 (a DPP row per arm, absent partners read as
                                     // +0.0: the totals then take the row-scan path for EVERY G up to 16), else G
    int stamp_block;                 // diagnostic: the workgroup whose phases are stamped (GM_TRK_STAMP_WG, default 0)
    int force_write_through;         // diagnostic (GM_TRK_FORCE_SC1=1): keep the cross-XCD exchange form even when a channel's workgroups share an XCD
    uint32_t per;                    // samples per workgroup slice (multiple of 64), fixed for the launch
    uint32_t stagger;                // 10 ns ticks a channel of stagger class 1 waits before its first epoch (class k: k times that; 0: none): see the kernel
    int fair_mode;                   // 1: raised for every other block; 2 (default): for three blocks of four; 3: for all of them (GM_TRK_FAIR, diagnostics)
    int fair_share;                  // CUs per XCD when the epochs are throughput-shaped (the five-arm loop's priority toggle), else 0
    uint32_t stagger_tab[2];         // the classes, 2 bits per channel of an XCD: [0] XCDs holding ceil(C / 8) channels, [1] the others
    uint32_t tag_base;               // unique per launch: tag = tag_base + epoch + 1
    unsigned long long* xchg;        // [2][n_channels][G][NV] granules of partials, then [n_channels][G] of XCC_IDs
    gm_trk_out* outs; uint8_t *processed, *lost, *lost_prn;   // [epochs][n_channels] (may be null)
    int* error_flag;                 // set to 1 if an exchange wait timed out (pinned host memory)
    int* error_flag_dev;             // the same in device memory: checked at the start of every launch
    long long* stamps;               // diagnostic only (may be null): [epochs][48] s_memtime stamps of workgroup 0 (8 phases, 16 waves x compute end, 16 x barrier arrival)
};

// diagnostic stamp (gm_trk_debug_stamps): one asm statement so the wait stays with the read, fenced against
// the scheduler on both sides (cdna_hip_programming.md §7 "In-kernel stamps")
// s_memrealtime: the constant 100 MHz counter (10 ns per unit).  s_memtime's rate is NOT constant on this part — the same epoch read
// 27 000 units with 432 workgroups resident and 44 000 with 240, in the same wall time (round 6) — so phase stamps in its units
// cannot be compared between configurations.
__device__ __forceinline__ long long stamp_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return (long long)t;
}

// Sum over the 64 lanes of a wavefront without LDS traffic: an inclusive scan inside each row of 16 lanes with
// DPP row_shr (lane 15 of a row ends with the row total), then the four row totals are read as scalars and
// added in row order.  Wave-uniform result; fixed order (deterministic).
__device__ __forceinline__ float wave_sum_dpp(float v) {
    int x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true));   // row_shr:1
    x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true));   // row_shr:2
    x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true));   // row_shr:4
    x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true));   // row_shr:8
    x = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(x, 15)), r1 = __int_as_float(__builtin_amdgcn_readlane(x, 31));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(x, 47)), r3 = __int_as_float(__builtin_amdgcn_readlane(x, 63));
    return ((r0 + r1) + r2) + r3;
}

// Sums over the 64 lanes of a wavefront of NV values at once (NV = 6 or 10), as a reduce-scatter: at each step a lane keeps
// half of the values it holds and hands the other half to its partner (DPP), so the work halves as the span doubles —
// 23 VALU instructions for six values (35 for ten) where six (ten) independent wave sums took ~100 (~170), and this runs on
// every wave of the workgroup right before the barrier the serial section waits behind.
// Returns X with: lane l (l & 15 < NV) holds the wave total of value l & 15.  Fixed order, deterministic.
template <int CTRL> __device__ __forceinline__ float dpp_take(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// lanes with `sel` keep hi and hand over lo, the others keep lo and hand over hi; result = kept + the partner's hand-over
template <int CTRL> __device__ __forceinline__ float keep_add(bool sel, float lo, float hi) {
    const float k = sel ? hi : lo, t = sel ? lo : hi;
    return k + dpp_take<CTRL>(t);
}
template <int NV> __device__ __forceinline__ float wave_sums_scatter(const float (&a)[NV], int lane) {
    static_assert(NV == 6 || NV == 10, "three or five arms");
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    constexpr int XOR1 = 0xB1, XOR2 = 0x4E, ROR4 = 0x124, ROR8 = 0x128;     // quad_perm [1,0,3,2], [2,3,0,1]; row_ror:4, :8
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    float x;
    if constexpr (NV == 6) {
        const float r01 = keep_add<XOR1>(b0, a[0], a[1]), r23 = keep_add<XOR1>(b0, a[2], a[3]), r45 = keep_add<XOR1>(b0, a[4], a[5]);
        const float q = keep_add<XOR2>(b1, r01, r23);            // value l & 3 over four lanes
        const float p = r45 + dpp_take<XOR2>(r45);               // value 4 + (l & 1) over four lanes
        x = keep_add<ROR4>(b2, q, p);                            // value l & 7 (6, 7 repeat 4, 5) over eight lanes
        x += dpp_take<ROR8>(x);                                  // ... over the row of sixteen
    } else {
        const float r01 = keep_add<XOR1>(b0, a[0], a[1]), r23 = keep_add<XOR1>(b0, a[2], a[3]), r45 = keep_add<XOR1>(b0, a[4], a[5]);
        const float r67 = keep_add<XOR1>(b0, a[6], a[7]), r89 = keep_add<XOR1>(b0, a[8], a[9]);
        const float q0 = keep_add<XOR2>(b1, r01, r23), q1 = keep_add<XOR2>(b1, r45, r67);
        float p = r89 + dpp_take<XOR2>(r89);
        const float o = keep_add<ROR4>(b2, q0, q1);              // value l & 7 over eight lanes
        p += dpp_take<ROR4>(p);                                  // value 8 + (l & 1) over eight lanes
        x = keep_add<ROR8>(b3, o, p);                            // value l & 15 (10.. repeat 8, 9) over the row
    }
    // the four rows: gfx950's lane swaps pair rows 0/1 and 2/3, then the two halves
    u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(r.x) + __uint_as_float(r.y);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}

// workgroup barrier that orders LDS traffic only: pending global loads (the sample prefetch) stay in flight
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// what wave 0 hands to the whole workgroup once per epoch
// (the channel state itself stays in wave 0's registers for the whole launch: only what the correlating waves need
// travels through LDS, and of the constants only the four that change from epoch to epoch)
struct EpochShared {
    uint64_t win;      // next_sample_index: first sample of the coming epoch's window
    uint32_t n;        // samples of the coming epoch (0: the data gate is closed)
    int alive;         // 0: the channel has just given up (do_work :183-210): the coming epoch does not run whatever n says
    int fast_car, fast_code;   // fast_car_ok / fast_code_ok for the coming epoch (both: the exact fast forms may be used)
    EpochConsts ec;
};

// n_known: the epilogue has just stored round(fs/(code_rate/len)) for the CURRENT code_rate in num_samples_per_code
// first: write every constant (launch start); later calls write the four per-epoch ones
__device__ __forceinline__ void prepare_epoch(const TrkDevCfg& cfg, uint64_t head, const gm_trk_state& s, EpochShared& sh,
                                              bool n_known, bool first, float n_cap) {
    const uint64_t n = n_known ? s.num_samples_per_code : samples_per_code(cfg, s.code_rate);     // update() :165-166
    bool run = s.active && n > 0 && n < (1ull << 31);
    if (run) run = (int64_t)(head - (s.next_sample_index + n)) >= 0;              // :170-172
    sh.win = s.next_sample_index;
    sh.n = run ? uint32_t(n) : 0u;
    sh.alive = 1;
    const EpochConsts ec = epoch_consts(cfg, s);
    if (first) sh.ec = ec;
    else { sh.ec.carrier_phase = ec.carrier_phase; sh.ec.two_pi_f = ec.two_pi_f; sh.ec.code_phase = ec.code_phase; sh.ec.step = ec.step; }
    sh.fast_car = fast_car_ok(ec, n_cap) ? 1 : 0;
    sh.fast_code = (fast_code_ok(ec, n) && float(uint32_t(n)) <= n_cap) ? 1 : 0;
}

// FAIR3: the three-arm correlation loop carries the co-tenant priority toggle of the five-arm one.  A template parameter, not a
// launch-uniform flag: the toggle's test inside the loop costs the 3 us latency chain of 32 channels x 25 Msps 75 ns per epoch even
// when it never fires, and two copies of the loop in one kernel cost it 9 % (DESIGN_HISTORY R6.6); launches shaped for throughput
// (a.fair_share != 0: eight or more rows per lane) take the FAIR3 instantiation, every other launch the plain one.
template <int ARMS, int MODE_T, int BOC_T, int T, int STRICT, bool FAIR3 = false>
// second launch bound = waves per SIMD with TRK_PERSIST_WG_PER_CU workgroups resident: the co-residency the exchange relies
// on must not be lost to a register count above 512 / that
__global__ __launch_bounds__(T, TRK_PERSIST_WG_PER_CU * T / 256) void trk_persistent_kernel(TrkPersistArgs a) {
    constexpr int NV = 2 * ARMS, NW = T / 64, KPF = 4;
    const TrkDevCfg& cfg = a.cfg;
    // workgroup -> (channel, slice).  Blocks b and b + 8 share an XCD (round-robin dispatch), so the grid is laid out as
    // ceil(C/8)*8 channel slots x G slices with slot = 8*(j/G) + xcd: all G workgroups of a channel sit on ONE XCD and the
    // partial sums they exchange every epoch can travel through that XCD's L2.  Slots beyond C are empty workgroups (they
    // leave at once).  The placement is only the expectation: what the exchange does is decided from the XCC_ID each
    // workgroup reads from the hardware (see the exchange below), never from this arithmetic.
    const int C = cfg.n_channels;
    int ch, g;
    if (a.packed) {
        // Packed layout (throughput-shaped launches whose channel count does not fill the eight XCDs evenly, e.g. 36): the C * G
        // workgroups form ONE run, dealt to the XCDs in eight equal pieces — workgroup o of XCD x is number x * per_xcd + o of the
        // run, channel = number / G.  Every place of every CU is used (36 channels: 14 workgroups each, 504 of 512 places, where the
        // slot layout gave 12 each on 432); a channel at a piece's edge has workgroups on two XCDs and exchanges through the
        // fabric (the XCC_ID hand-shake below decides that per channel, as always).
        const int num = int(blockIdx.x & 7) * a.packed + int(blockIdx.x >> 3);
        if (int(blockIdx.x >> 3) >= a.packed || num >= C * a.G) return;
        ch = num / a.G; g = num - ch * a.G;
    } else {
        ch = (int(blockIdx.x >> 3) / a.G) * 8 + int(blockIdx.x & 7);
        g = int(blockIdx.x >> 3) % a.G;
        if (ch >= C) return;
    }
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    __shared__ float wsum[NW][NV];
    __shared__ EpochShared sh;          // wave 0 -> everyone, once per epoch
    __shared__ int ctl;                 // 0 continue, 1 exchange timed out
    __shared__ float pre_phase[4];      // the carrier / code phase the coming state will hold, and the code-rate interval that keeps n (waves 2 / 3, every epoch)
    // The channel state between epochs: wave 0's copy (it owns the carrier fields and the bookkeeping) and wave 1's (code
    // fields).  Parked in LDS rather than in registers while the workgroup correlates: held in registers it put the hot loop
    // over the 128-VGPR budget of two workgroups per CU, and the spilled values came back through scratch memory in the
    // middle of the serial section.
    __shared__ gm_trk_state st_sh[2];
    static_assert(NW >= 4, "waves 0/1 run the two halves of the serial section, waves 2/3 the phase advances");
    __shared__ float gathered[2][256];  // the G*NV partials of one epoch (G other than 16 / 32): one staging area per gathering wave
    extern __shared__ float chips_pad[];   // the channel's chip row as floats, one guard entry at each end (correlate_sample_fast);
    // BOC: the half-chip table (with the sub-carrier sign) in FRONT of it (boc_plain_offset)
    float* const chips = BOC_T == 1 ? chips_pad + boc_plain_offset(cfg.code_len) : chips_pad;
    const float* const fast_tab = chips_pad;

    // an earlier launch of the same call (more than 4095 passes are several launches) has timed out: do nothing, the host
    // reports the error after it synchronises
    // (a device-memory twin of the flag, one lane per workgroup: reading the pinned host word itself cost 0.4 us per epoch
    // even from one lane per workgroup, 0.7 ms per launch from every lane)
    __shared__ int s_abort;
    if (threadIdx.x == 0) s_abort = __hip_atomic_load(a.error_flag_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_abort != 0) return;
    gm_trk_state s0 = a.states[ch];
    const int row = code_row(cfg, s0);
    const bool leader = (g == 0 && tid == 0);
    int e = 0;
    bool timed_out = false;
    const bool ran = s0.active && row >= 0 && row < cfg.n_codes;
    // Same-XCD hand-shake, once per launch: every workgroup publishes the XCC_ID it runs on (write-through, like the first
    // epoch's partials); after the first epoch's exchange each one has read all G of them and, when they agree, later epochs
    // publish with PLAIN stores, which stay in the XCD's L2 where the partners' L1-bypassing polls find them (a write-through
    // store drops the line, so the poll goes out to the fabric: 2050 -> 550 cycles of waiting per epoch at 32 channels).
    unsigned long long* const xcc_slot = a.xchg + size_t(2) * C * a.GS * NV + size_t(ch) * a.G;
    uint32_t my_xcc = 0;
    bool same_xcd = false;
    if (ran) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
        my_xcc &= 0xfu;
        if (tid == 0)
            __hip_atomic_store(&xcc_slot[g], (unsigned long long)my_xcc | ((unsigned long long)(a.tag_base + 1u) << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int8_t* crow = a.codes + size_t(row) * cfg.code_len;
        for (int i = tid; i < cfg.code_len; i += T) chips[i + 1] = float(crow[i]);
        if (tid == 0) {   // what floor(phase) = -1 and = len read (get_ca_chip :275: `as usize` saturates / FIXED wraps; `% len`)
            chips[0] = float(crow[cfg.code_index_mode == GM_CODE_INDEX_FAITHFUL ? 0 : cfg.code_len - 1]);
            chips[cfg.code_len + 1] = float(crow[0]);
        }
        if constexpr (BOC_T == 1) {   // the half-chip table of correlate_block_fast: entry j + 2 <-> half chip j = -2 .. 2 len + 1
            float* const t2 = chips_pad;
            for (int j = tid; j < 2 * cfg.code_len; j += T) t2[j + 2] = (j & 1) ? -float(crow[j >> 1]) : float(crow[j >> 1]);
            if (tid == 0) {
                const float lo = float(crow[cfg.code_index_mode == GM_CODE_INDEX_FAITHFUL ? 0 : cfg.code_len - 1]), hi = float(crow[0]);
                t2[0] = lo; t2[1] = -lo; t2[2 * cfg.code_len + 2] = hi; t2[2 * cfg.code_len + 3] = -hi;
            }
        }
        // an upper bound of any epoch's sample count in this launch, for the fast forms' range checks
        // (wave-uniform floats that live for the whole launch are pinned to scalar registers: left in VGPRs they were the
        // values the register allocator spilled, and their reloads from scratch sat in the middle of the serial section)
        const float n_cap = uniform_f32(2.0f * float(a.per) * float(a.G));
        if (tid == 0) { ctl = 0; prepare_epoch(cfg, a.head, s0, sh, false, true, n_cap); }
        if (lane == 0 && wave < 2) st_sh[wave] = s0;
        __syncthreads();
        // slice geometry, fixed for the launch: `per` samples per workgroup (multiple of 64 lanes); the last
        // workgroup also takes whatever a longer code period adds beyond G*per
        const uint32_t per = a.per;
        const uint32_t i0 = uint32_t(g) * per;
        // software prefetch: the first KPF strided samples of the NEXT epoch are requested while this
        // epoch's partial sums travel between workgroups (its window start is known: next + n)
        // (four named values, not an array: hipcc kept `cf pf[4]` in scratch memory, and the scratch store behind
        // each prefetch made every wave wait for its loads right there instead of at first use)
        static_assert(KPF == 4, "prefetch depth is spelled out below");
        cf pf0 = a.ring[(s0.next_sample_index + i0 + tid) & a.mask];
        cf pf1 = a.ring[(s0.next_sample_index + i0 + tid + T) & a.mask];
        cf pf2 = a.ring[(s0.next_sample_index + i0 + tid + 2 * T) & a.mask];
        cf pf3 = a.ring[(s0.next_sample_index + i0 + tid + 3 * T) & a.mask];

        // Throughput-shaped epochs (many samples per lane): the two workgroups a CU holds belong to different channels, start
        // together and do the same work, so they stay in lock-step — both correlate at once (sharing the issue slots), then both
        // sit in their serial sections (exchange + scalar update: the CU idles).  Channels that share CUs therefore start a third
        // (or two thirds) of an epoch apart, once per launch: from then on one tenant's serial section runs under the other's
        // correlation.  All G workgroups of a channel take the same decision (they meet in every epoch's exchange).  A hint
        // only: wherever the dispatcher puts the workgroups, the results are the same.
        if (a.stamps && a.stamp_block == -2 && tid == 0) {        // diagnostic (GM_TRK_STAMP_WG=-2): where the dispatcher put every workgroup
            uint32_t hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            a.stamps[blockIdx.x] = (long long)(((unsigned long long)my_xcc << 32) | hw);
            a.stamps[512 + blockIdx.x] = stamp_now();            // ... and when it got going (absolute, 10 ns); its end at [1024 + b] below
        }
        // (see the five-arm correlation loop) second round of the dispatcher over this XCD's CUs: the younger tenant of its CU
        const bool younger = a.fair_share && int(blockIdx.x >> 3) >= a.fair_share;
        if (a.stagger) {
            // class of this channel (2 bits per channel of the XCD, by its index there; one table for the XCDs that hold the larger
            // channel count, one for the others): a colouring, made on the host, of "shares CUs with" — see launch_trk_persistent
            const int xcd = int(blockIdx.x & 7), local = int(blockIdx.x >> 3) / a.G;
            const int count = (C - xcd + 7) >> 3, most = (C + 7) >> 3;
            const uint32_t cls = ((count == most ? a.stagger_tab[0] : a.stagger_tab[1]) >> (2 * (local & 7))) & 3u;
            if (cls) {
                if (tid == 0) {
                    const long long t0 = wall_clock64();                       // 100 MHz, whatever the shader clock does
                    const long long until = (long long)(a.stagger * cls);
                    while (wall_clock64() - t0 < until) __builtin_amdgcn_s_sleep(32);
                }
                __syncthreads();
            }
        }
        // which granules of the exchange block this lane gathers (lane + 64 q: arm-major, GS per arm) is fixed for the launch — the
        // places of partners that do not exist (GS = 16 > G) are never polled.  Worked out HERE, not in the epoch's serial section.
        bool mine[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) mine[q] = lane + q * 64 < a.GS * NV && (a.GS == a.G || ((lane + q * 64) & 15) < a.G);
        for (; e < a.epochs; ++e) {
            const uint32_t n = sh.alive ? sh.n : 0u;
            if (n == 0) break;                     // state is identical in the G workgroups: they all leave
            const EpochConsts ec = sh.ec;
            const uint64_t win = sh.win;
            const bool st_on = a.stamps && int(blockIdx.x) == a.stamp_block && tid == 0;
            long long* stp = a.stamps + size_t(e) * 48;
            if (st_on) stp[0] = stamp_now();
            const uint32_t i1 = (g == a.G - 1 || i0 + per > n) ? n : i0 + per;
            const uint32_t cnt = i1 > i0 ? i1 - i0 : 0u;     // samples of this slice (workgroup-uniform)
            const uint32_t full = cnt / T;                     // strided passes in which every lane has a sample
            float acc[NV], acc2[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) { acc[k] = 0.0f; acc2[k] = 0.0f; }
            if (!STRICT && (sh.fast_car & sh.fast_code)) {
                // A lane's samples are b0 + j*T, j < tot (tot = full, or full + 1 on the lanes of the ragged last pass).  They
                // are processed FOUR at a time as one straight-line block with four accumulator sets, so the scheduler
                // interleaves four independent dependent chains (division, f64 reduction, LDS look-ups): the phase is bound by
                // the latency of one wave's instruction stream, not by issue slots (one workgroup alone on a CU takes as long
                // as two).  A slot without a sample takes a zero sample, which adds exactly nothing to the sums; blocks are
                // sized per wave (a wave whose lanes all have three samples runs a block of three).
                const uint32_t b0 = i0 + tid;
                const uint32_t tot = full + ((b0 + full * T) < i1 ? 1u : 0u);
                const uint32_t wtot = full + (__any((b0 + full * T) < i1) ? 1u : 0u);       // wave-uniform
                const cf zero = cf_make(0.0f, 0.0f);
                const float fb0 = float(b0);                             // sample indices as floats: exact below 2^24
                constexpr float TF = float(T);
                // block width: four samples with three arms; two with five (ten accumulators per sample: four spill)
                constexpr uint32_t IL = ARMS == 3 ? 4 : 2;
                if constexpr (ARMS == 5) {
                    // Five arms: throughput-shaped epochs (BASELINE configs[4]: 200 000 samples per channel-epoch, ~33 per lane), so
                    // the samples travel TWO blocks ahead of their use in a rotating pair of named register pairs — (pf0, pf1)
                    // feeds the blocks at j = 0, 4, 8 .. and is refilled with the samples of j + 4 the moment it has been read,
                    // (pf2, pf3) likewise for j = 2, 6, 10 .. — instead of being loaded at the head of the block that consumes them
                    // (every wave then sat out most of a memory round trip per block: 45 % of the wave cycles parked, round 5).
                    // At the loop's entry the four registers hold samples 0 .. 3 (requested during the previous epoch's exchange).
#if GM_TRK_IL5 == 4
                    // blocks of FOUR samples (four interleaved dependent chains per wave: the phase is bound by the latency of a
                    // wave's instruction stream), the NEXT block's samples requested the moment this block's have been taken out
                    // of (pf0 .. pf3); sample q of a block adds into the same set of sums as in blocks of two (even -> acc, odd -> acc2)
                    for (uint32_t j = 0; j < wtot; j += 4) {
                        const uint32_t left = wtot - j;
                        cf dd[4];
                        float ff[4];
                        dd[0] = pf0; dd[1] = pf1; dd[2] = pf2; dd[3] = pf3;
                        if (j + 4 < wtot) {          // (wave-uniform; a lane past its own count reads a sample it will not use: the ring is masked)
                            pf0 = a.ring[(win + b0 + (j + 4) * T) & a.mask];
                            pf1 = a.ring[(win + b0 + (j + 5) * T) & a.mask];
                            pf2 = a.ring[(win + b0 + (j + 6) * T) & a.mask];
                            pf3 = a.ring[(win + b0 + (j + 7) * T) & a.mask];
                        }
                        if (j + 4 > full) {          // wave-uniform: this block touches the ragged last row
#pragma unroll
                            for (uint32_t q = 0; q < 4; ++q) dd[q] = j + q < tot ? dd[q] : zero;
                        }
                        ff[0] = fb0 + float(j * T);
#pragma unroll
                        for (uint32_t q = 1; q < 4; ++q) ff[q] = ff[0] + float(q) * TF;
                        if (left >= 4) correlate_block_fast<ARMS, BOC_T, 4>(ec, fast_tab, dd, ff, acc, acc2);
                        else if (left == 3)
                            correlate_block_fast<ARMS, BOC_T, 3>(ec, fast_tab, reinterpret_cast<const cf(&)[3]>(dd), reinterpret_cast<const float(&)[3]>(ff), acc, acc2);
                        else if (left == 2)
                            correlate_block_fast<ARMS, BOC_T, 2>(ec, fast_tab, reinterpret_cast<const cf(&)[2]>(dd), reinterpret_cast<const float(&)[2]>(ff), acc, acc2);
                        else
                            correlate_block_fast<ARMS, BOC_T, 1>(ec, fast_tab, reinterpret_cast<const cf(&)[1]>(dd), reinterpret_cast<const float(&)[1]>(ff), acc, acc2);
                    }
#else
                    // `full` rows hold a sample for EVERY lane of the workgroup: their blocks take the registers as they are; only the
                    // ragged last row (wtot = full + 1) zeroes the lanes beyond their count (a compare + two selects per sample
                    // otherwise: half-rate instructions on every sample of the epoch)
                    const auto run2 = [&](uint32_t j, cf s0, cf s1) {
                        const uint32_t left = wtot - j;
                        cf dd[2];
                        float ff[2];
                        dd[0] = s0; dd[1] = s1;
                        if (j + 2 > full) {                         // wave-uniform: this block touches the ragged row
                            dd[0] = j < tot ? s0 : zero;
                            dd[1] = j + 1 < tot ? s1 : zero;
                        }
                        ff[0] = fb0 + float(j * T);
                        ff[1] = ff[0] + TF;
                        if (left >= 2) correlate_block_fast<ARMS, BOC_T, 2>(ec, fast_tab, dd, ff, acc, acc2);
                        else correlate_block_fast<ARMS, BOC_T, 1>(ec, fast_tab, reinterpret_cast<const cf(&)[1]>(dd), reinterpret_cast<const float(&)[1]>(ff), acc, acc2);
                    };
                    for (uint32_t j = 0; j < wtot; j += 4) {
                        const cf u0 = pf0, u1 = pf1;
                        if (j + 4 < wtot) {          // (wave-uniform; a lane past its own count reads a sample it will not use: the ring is masked)
                            pf0 = a.ring[(win + b0 + (j + 4) * T) & a.mask];
                            pf1 = a.ring[(win + b0 + (j + 5) * T) & a.mask];
                        }
                        // Fair shares for the two tenants of a CU: a SIMD issues for its OLDEST ready wave, so the workgroup that got to
                        // the CU first (the dispatcher's first round over the XCD) ran a whole epoch in 13.5 us while its co-tenant of
                        // the second round took 20 .. 24 us (per-workgroup start / end times, tools/trk_placement.py) — and the launch
                        // lasts as long as its slowest channel.  The second-round workgroup raises its priority for three blocks of four
                        // (fair_mode 2; every other block, mode 1, still leaves it behind — it spends most of its TIME in the blocks it
                        // runs unraised; always, mode 3, starves the other tenant instead): channel durations within 6 % of each other,
                        // 21.6 -> 19.7 us per code period on one box.  Rotating four priority levels over the four waves of a SIMD, or
                        // raising the upper half of each workgroup's waves as well, did worse (20.4 / 20.2 us).
                        if (younger) __builtin_amdgcn_s_setprio(1);
                        run2(j, u0, u1);
                        if (younger && !(a.fair_mode >= 3 || (a.fair_mode == 2 && (j & 4)))) __builtin_amdgcn_s_setprio(0);
                        if (j + 2 >= wtot) break;
                        const cf u2 = pf2, u3 = pf3;
                        if (j + 6 < wtot) {
                            pf2 = a.ring[(win + b0 + (j + 6) * T) & a.mask];
                            pf3 = a.ring[(win + b0 + (j + 7) * T) & a.mask];
                        }
                        run2(j + 2, u2, u3);
                    }
                    if (younger) __builtin_amdgcn_s_setprio(0);
#endif
                } else
                for (uint32_t j = 0; j < wtot; j += IL) {                // the first block(s) come from the prefetched registers
                    const uint32_t left = wtot - j;
                    if constexpr (FAIR3) {                               // fair issue shares for a CU's two tenants (see the five-arm loop)
                        if (younger) {
                            if (a.fair_mode >= 3 || (j & (a.fair_mode == 2 ? 12u : 4u))) __builtin_amdgcn_s_setprio(1);
                            else __builtin_amdgcn_s_setprio(0);
                        }
                    }
                    cf dd[IL];
                    float ff[IL];
#pragma unroll
                    for (uint32_t q = 0; q < IL; ++q) dd[q] = zero;
                    if (j == 0) {
                        if (tot > 0) dd[0] = pf0;
                        if (tot > 1) dd[1] = pf1;
                        if constexpr (IL == 4) {
                            if (tot > 2) dd[2] = pf2;
                            if (tot > 3) dd[3] = pf3;
                        }
                    } else {
#pragma unroll
                        for (uint32_t q = 0; q < IL; ++q)
                            if (j + q < tot) dd[q] = a.ring[(win + b0 + (j + q) * T) & a.mask];
                    }
                    ff[0] = fb0 + float(j * T);
#pragma unroll
                    for (uint32_t q = 1; q < IL; ++q) ff[q] = ff[0] + float(q) * TF;
                    // one instantiation per block size; a slot beyond the wave's count (left < IL) is not computed at all
                    if (left >= IL) correlate_block_fast<ARMS, BOC_T, IL>(ec, fast_tab, dd, ff, acc, acc2);
                    else if (IL == 4 && left == 3)
                        correlate_block_fast<ARMS, BOC_T, 3>(ec, fast_tab, reinterpret_cast<const cf(&)[3]>(dd), reinterpret_cast<const float(&)[3]>(ff), acc, acc2);
                    else if (IL == 4 && left == 2)
                        correlate_block_fast<ARMS, BOC_T, 2>(ec, fast_tab, reinterpret_cast<const cf(&)[2]>(dd), reinterpret_cast<const float(&)[2]>(ff), acc, acc2);
                    else
                        correlate_block_fast<ARMS, BOC_T, 1>(ec, fast_tab, reinterpret_cast<const cf(&)[1]>(dd), reinterpret_cast<const float(&)[1]>(ff), acc, acc2);
                }
                if (FAIR3 && younger) __builtin_amdgcn_s_setprio(0);
            } else if (STRICT && (sh.fast_car & sh.fast_code)) {   // strict_libm: the exact fast forms of the code phase and of
                // x / fs, the carrier's cos / sin by sincosf_glibc (f64) — sample by sample, no prefetch use
                for (uint32_t i = i0 + tid; i < i1; i += T)
                    correlate_sample<ARMS, true, MODE_T, BOC_T, float, STRICT>(ec, chips, a.ring[(win + i) & a.mask], i, acc);
            } else {   // out-of-family state (e.g. set by the caller): general fmodf, no prefetch use
                for (uint32_t i = i0 + tid; i < i1; i += T)
                    correlate_sample<ARMS, false, MODE_T, BOC_T, float, STRICT>(ec, chips, a.ring[(win + i) & a.mask], i, acc);
            }
            if (st_on) stp[1] = stamp_now();
            if (a.stamps && int(blockIdx.x) == a.stamp_block && lane == 0) stp[8 + wave] = stamp_now();        // per-wave compute end
            {   // request the next epoch's samples now; they land during the exchange below
                const uint64_t nb = win + n;
                pf0 = a.ring[(nb + i0 + tid) & a.mask];
                pf1 = a.ring[(nb + i0 + tid + T) & a.mask];
                pf2 = a.ring[(nb + i0 + tid + 2 * T) & a.mask];
                pf3 = a.ring[(nb + i0 + tid + 3 * T) & a.mask];
            }
            // the phases the coming state will hold are functions of the OLD state and n alone (:240-242, :265-267): waves 2
            // and 3 work them out here, off the serial chain that follows the exchange
            if (wave == 2) {
                const float ph = advance_carrier_phase(cfg, ec.carrier_phase, ec.two_pi_f, float(n));
                if (lane == 0) pre_phase[0] = ph;
            } else if (wave == 3) {
                const float ph = advance_code_phase(cfg, ec.code_phase, ec.step, float(n));
                // ... and the code rates for which the NEXT epoch provably keeps this epoch's length n (the rate moves by
                // parts in 1e7 per epoch; the length changes when it crosses a half-integer quotient)
                float rlo = 1.0f, rhi = 0.0f;
                if (n < (1u << 23)) spc_rate_bounds(cfg.fs, cfg.code_len_f, float(n), rlo, rhi);
                if (lane == 0) { pre_phase[1] = ph; pre_phase[2] = rlo; pre_phase[3] = rhi; }
            }
#pragma unroll
            for (int k = 0; k < NV; ++k) acc[k] += acc2[k];
            const float wtotal = wave_sums_scatter<NV>(acc, lane);     // lane k < NV: this wave's total of value k
            if (lane < NV) wsum[wave][lane] = wtotal;
            if (a.stamps && int(blockIdx.x) == a.stamp_block && lane == 0) stp[24 + wave] = stamp_now();       // per-wave barrier arrival
            lds_barrier();    // NOT __syncthreads(): its fence would wait for the prefetch loads (vmcnt(0))
            if (st_on) stp[2] = stamp_now();
            gm_trk_state st;      // waves 0 / 1: this wave's copy of the channel state, from LDS and back (after the barrier below)
            if (wave < 2) {
                st = st_sh[wave];
                // The serial section: wave 0 publishes this workgroup's partial; waves 0 AND 1 each gather the G partials and
                // form the totals (same loads, same order: identical values), then wave 0 runs the carrier half of the scalar
                // update and wave 1 the code half, side by side (a lone wave issues one instruction per ~5 cycles whatever
                // the dependencies, so two waves halve the section).  The other workgroup of this CU is usually correlating
                // on the same SIMDs: raised issue priority keeps the chains from queueing behind it.
                __builtin_amdgcn_s_setprio(3);
                const bool st1_on = a.stamps && int(blockIdx.x) == a.stamp_block && tid == 64;
                // this workgroup's partial (waves added in a fixed order), published as {value, tag} granules
                const uint32_t tag = a.tag_base + uint32_t(e) + 1u;
                unsigned long long* slot = a.xchg + (size_t(e & 1) * C + ch) * a.GS * NV;
                if (wave == 0 && lane < NV) {
                    float pw[NW];
#pragma unroll
                    for (int w = 0; w < NW; ++w) pw[w] = wsum[w][lane];      // all reads in flight, then the fixed-order sum
                    float p = pw[0];
#pragma unroll
                    for (int w = 1; w < NW; ++w) p += pw[w];
                    const unsigned long long gran =
                        (unsigned long long)__float_as_uint(p) | ((unsigned long long)tag << 32);
                    if (same_xcd)      // plain: the line stays in this XCD's L2 (see the hand-shake above)
                        asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(&slot[lane * a.GS + g]), "v"(gran) : "memory");
                    else
                        __hip_atomic_store(&slot[lane * a.GS + g], gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // arm-major: [k][g], GS granules per arm
                }
                if (st_on) stp[3] = stamp_now();
                // gather the G partials: lane l polls granules l, l+64, ... (GS*NV <= 256); with GS = 16 > G the places of partners
                // that do not exist are never polled and count as +0.0
                const int ng = a.GS * NV;
                float val[4] = {0.f, 0.f, 0.f, 0.f};
                const long long t0 = wall_clock64();
                bool to = false;
                // first sweep: every granule this lane is responsible for is requested before any is examined, so the
                // common case (all partners already published) costs ONE memory round trip, not one per slot
                unsigned long long gr[4] = {0ull, 0ull, 0ull, 0ull};
                bool pending[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pending[q] = mine[q];
                    if (pending[q]) gr[q] = __hip_atomic_load(&slot[lane + q * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (pending[q] && uint32_t(gr[q] >> 32) == tag) pending[q] = false;
                // stragglers: re-poll only what is still missing; the wall clock (bounded wait) is read every 64th round
                uint32_t rounds = 0;
                while ((pending[0] | pending[1] | pending[2] | pending[3]) && !to) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (pending[q]) gr[q] = __hip_atomic_load(&slot[lane + q * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (pending[q] && uint32_t(gr[q] >> 32) == tag) pending[q] = false;
                    // (no s_sleep here: measured +0.5 us per epoch at 32 channels — the poll IS the critical path)
                    if ((++rounds & 63u) == 0u && wall_clock64() - t0 > 20000000ll) to = true;   // 0.2 s at 100 MHz
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (lane + q * 64 < ng) val[q] = __uint_as_float(uint32_t(gr[q]));       // (never polled: gr = 0 = +0.0f)
                if (e == 0 && wave == 0) {   // the hand-shake: the partners' XCC_IDs (published before their first partials were)
                    const unsigned long long want = (unsigned long long)(a.tag_base + 1u);
                    unsigned long long xg = want << 32 | my_xcc;
                    bool pend = lane < a.G;
                    while (pend && !to) {
                        xg = __hip_atomic_load(&xcc_slot[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pend = (xg >> 32) != want;
                        if ((++rounds & 63u) == 0u && wall_clock64() - t0 > 20000000ll) to = true;
                    }
                    same_xcd = __all(!pend && uint32_t(xg) == my_xcc) != 0 && !a.force_write_through;
                }
                to = __any(to);
                if (st_on) { stp[4] = stamp_now(); stp[45] = (long long)rounds; }
                if (st1_on) { stp[40] = stamp_now(); stp[46] = (long long)rounds; }
                // totals of the G partials of every arm, identical in all lanes and in all workgroups of the channel.
                // Granules are arm-major ([k][g]).  G <= 16 (GS == 16): arm k's partials sit in one DPP row of 16 lanes (arms 0-3 in
                // the first sweep's registers, 4.. in the next), so four row_shr adds leave the arm total in the row's last
                // lane and a readlane broadcasts it — no LDS, no fences.  Other G: staged in LDS (DS operations of one wave
                // complete in order) and added by lane k in workgroup order.
                float v[NV];
                if (a.GS == 16) {
                    float r[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) r[q] = val[q];
#pragma unroll
                    for (int q = 0; q < (NV + 3) / 4; ++q) {
                        int x = __float_as_int(r[q]);
                        r[q] += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true));   // row_shr:1
                        x = __float_as_int(r[q]);
                        r[q] += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true));   // row_shr:2
                        x = __float_as_int(r[q]);
                        r[q] += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true));   // row_shr:4
                        x = __float_as_int(r[q]);
                        r[q] += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true));   // row_shr:8
                    }
#pragma unroll
                    for (int k = 0; k < NV; ++k)
                        v[k] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(r[k / 4]), (k % 4) * 16 + 15));
                } else if (a.G == 32 && NV <= 8) {   // arm k = two rows of sweep k/2: row totals by row_shr, pair added as scalars
                    float r[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) r[q] = val[q];
#pragma unroll
                    for (int q = 0; q < (NV + 1) / 2; ++q) {
                        int x = __float_as_int(r[q]);
                        r[q] += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true));
                        x = __float_as_int(r[q]);
                        r[q] += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true));
                        x = __float_as_int(r[q]);
                        r[q] += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true));
                        x = __float_as_int(r[q]);
                        r[q] += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true));
                    }
#pragma unroll
                    for (int k = 0; k < NV; ++k) {
                        const uint32_t u = __float_as_uint(r[k / 2]);
                        v[k] = __uint_as_float(__builtin_amdgcn_readlane(u, (k % 2) * 32 + 15)) +
                               __uint_as_float(__builtin_amdgcn_readlane(u, (k % 2) * 32 + 31));
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int idx = lane + q * 64;
                        if (idx < ng) gathered[wave][idx] = val[q];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    float tk = 0.0f;
                    if (lane < NV)
                        for (int gg = 0; gg < a.G; ++gg) tk += gathered[wave][lane * a.G + gg];
#pragma unroll
                    for (int k = 0; k < NV; ++k) v[k] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(tk), k));
                }
                if (st_on) stp[5] = stamp_now();
                if (st1_on) stp[43] = stamp_now();
                if (wave == 0) {          // ---- carrier half + the bookkeeping (do_work :183-210, run_loop_filters' PLL :279-290)
                    uint8_t lst = 0, lprn = 0;
                    if (!to) {
                        const CarrierHalf h = carrier_half<ARMS>(cfg, st, v, n, TRK_MODE_DO_WORK, &pre_phase[0]);
                        st.prn = h.prn; st.active = h.active; st.lost_counter = h.lost_counter; st.next_sample_index = h.next_sample_index;
                        st.carrier_freq = h.carrier_freq; st.carrier_phase = h.carrier_phase; st.carrier_error = h.carrier_error;
                        st.carrier_nco = h.carrier_nco; st.i_prompt = h.i_prompt; st.q_prompt = h.q_prompt;
                        lst = h.lst; lprn = h.lprn;
                    }
                    if (st_on) stp[6] = stamp_now();
                    if (lane == 0) {      // the carrier's share of the NEXT epoch's constants
                        const EpochConsts nx = epoch_consts(cfg, st);
                        sh.ec.carrier_phase = nx.carrier_phase; sh.ec.two_pi_f = nx.two_pi_f;
                        sh.fast_car = fast_car_ok(nx, n_cap) ? 1 : 0;
                        sh.win = st.next_sample_index;           // this wave owns the bookkeeping
                        sh.alive = st.active ? 1 : 0;
                        if (to) ctl = 1;
                        if (g == 0 && !to) {
                            const size_t o = size_t(e) * C + ch;
                            if (a.outs) {
                                if constexpr (ARMS == 5) a.outs[o] = make_out<ARMS>(v);
                                else {   // ive..qvl are zero since the buffer was allocated (gm_api.hip trk_reserve_epochs): the six
                                         // live sums only — the four zeros were values the register allocator kept (and spilled)
                                    float* po = reinterpret_cast<float*>(&a.outs[o]);
#pragma unroll
                                    for (int k = 0; k < NV; ++k) po[k] = v[k];
                                }
                            }
                            if (a.processed) a.processed[o] = 1;
                            if (a.lost) a.lost[o] = lst;
                            if (a.lost_prn) a.lost_prn[o] = lprn;
                        }
                    }
                } else {                  // ---- code half (DLL :291-302, :265-270) + the gate and length of the next epoch
                    if (!to) {
                        st.num_samples_per_code = n;                 // update() stores the length it used (:166)
                        // (the lost counter this half's give-up test reads is kept here; prn / active / next_sample_index are
                        // wave 0's: this wave only needs to know whether the channel goes on, which wave 0 tells through sh.alive)
                        const bool locked = trk_locked(cfg, v[0], v[1]);
                        const uint32_t lost_next = locked ? 0u : (trk_give_up(cfg, st, locked) ? 0u : st.lost_counter + 1u);
                        if (st1_on) stp[44] = stamp_now();
                        const CodeHalf h = code_half<ARMS>(cfg, st, v, n, n, TRK_MODE_DO_WORK, &pre_phase[1], pre_phase[2], pre_phase[3]);
                        st.lost_counter = lost_next;
                        st.num_samples_per_code = h.num_samples_per_code; st.code_phase = h.code_phase; st.code_error = h.code_error;
                        st.code_nco = h.code_nco; st.code_rate = h.code_rate;
                    }
                    if (st1_on) stp[41] = stamp_now();
                    if (lane == 0) {
                        // the epilogue has just stored round(fs/(code_rate/len)) for the new code_rate (update() :165-166)
                        const uint64_t nn = st.num_samples_per_code;
                        bool run = nn > 0 && nn < (1ull << 31);
                        if (run) run = (int64_t)(a.head - (win + uint64_t(n) + nn)) >= 0;                 // :170-172 (next_sample_index = win + n while the channel lives)
                        const EpochConsts nx = epoch_consts(cfg, st);
                        sh.n = run ? uint32_t(nn) : 0u;
                        sh.ec.code_phase = nx.code_phase; sh.ec.step = nx.step;
                        sh.fast_code = (fast_code_ok(nx, nn) && float(uint32_t(nn)) <= n_cap) ? 1 : 0;
                        if (to) ctl = 1;
                    }
                    if (st1_on) stp[42] = stamp_now();
                }
                __builtin_amdgcn_s_setprio(0);
            }
            lds_barrier();
            if (ctl) { timed_out = true; break; }
            if (wave < 2 && lane == 0) st_sh[wave] = st;      // off the serial chain: the other waves are already correlating
            // no third barrier: wsum is rewritten only after every wave has passed the NEXT epoch's correlation, and
            // `sh` only after the next epoch's first barrier, which no wave reaches before reading it above
            if (st_on) stp[7] = stamp_now();
        }
        if (a.stamps && a.stamp_block == -2 && tid == 0) a.stamps[1024 + blockIdx.x] = stamp_now();
        if (tid == 0 && timed_out) {
            *a.error_flag = 1;                                                                       // host's copy (pinned)
            __hip_atomic_store(a.error_flag_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // later launches' copy
        }
    }
    if (ran) {      // carrier fields and bookkeeping from wave 0's copy, code fields from wave 1's
        __syncthreads();
        if (leader) {   // (the leader is lane 0 of wave 0) an idle channel's state is left as it was
            gm_trk_state st = st_sh[0];
            st.num_samples_per_code = st_sh[1].num_samples_per_code; st.code_phase = st_sh[1].code_phase;
            st.code_error = st_sh[1].code_error; st.code_nco = st_sh[1].code_nco; st.code_rate = st_sh[1].code_rate;
            a.states[ch] = st;
        }
    }
    if (leader) {
        gm_trk_out z;
        z.ip = z.qp = z.ie = z.qe = z.il = z.ql = z.ive = z.qve = z.ivl = z.qvl = 0.0f;
        for (int r = e; r < a.epochs; ++r) {   // passes in which this channel did not run
            const size_t o = size_t(r) * C + ch;
            if (a.outs) a.outs[o] = z;
            if (a.processed) a.processed[o] = 0;
            if (a.lost) a.lost[o] = 0;
            if (a.lost_prn) a.lost_prn[o] = 0;
        }
    }
}

// granules per arm in the exchange block (TrkPersistArgs::GS); the block is 2 * n_channels * GS * (2 * arms) granules + n_channels * G
int trk_persistent_granule_stride(int G) { return G <= 16 ? 16 : G; }

// dynamic LDS of the persistent kernel: the chip row as floats with one guard entry at each end, rounded up to 16 B
static size_t trk_persistent_lds(const TrkDevCfg& cfg) {
    size_t floats = size_t(cfg.code_len + 2);                                    // the padded chip row
    if (cfg.boc11) floats += size_t(boc_plain_offset(cfg.code_len));             // + the half-chip table of the BOC fast path in front of it
    return (floats * sizeof(float) + 15) & ~size_t(15);
}

void launch_trk_persistent(hipStream_t st, const TrkDevCfg& cfg, const int8_t* d_codes, gm_trk_state* d_states,
                           const cf* ring, uint64_t mask, uint64_t head, int G, int packed, int epochs, uint32_t tag_base,
                           unsigned long long* d_xchg, gm_trk_out* d_outs, uint8_t* d_processed, uint8_t* d_lost,
                           uint8_t* d_lost_prn, int* d_error, int* d_error_dev, long long* d_stamps) {
    TrkPersistArgs a;
    a.error_flag_dev = d_error_dev;
    a.stamps = d_stamps;
    a.cfg = cfg; a.codes = d_codes; a.states = d_states; a.ring = ring; a.mask = mask; a.head = head;
    a.G = G; a.GS = trk_persistent_granule_stride(G); a.epochs = epochs; a.tag_base = tag_base;
    a.packed = packed ? (cfg.n_channels * G + 7) / 8 : 0;
    a.stamp_block = diag_int("GM_TRK_STAMP_WG", 0);
    a.force_write_through = diag_int("GM_TRK_FORCE_SC1", 0) != 0 ? 1 : 0;
    {   // slice length from the nominal code period (+1 % margin; +0.2 % where an epoch is many samples per lane: there the margin is
        // rows of work on every workgroup but the last, which takes whatever a longer period adds anyway), whole wavefronts
        const float nn = roundf(cfg.fs / (cfg.nominal_code_rate / cfg.code_len_f));
        const float margin = (nn > 0 && nn / float(G) / float(TRK_PERSIST_THREADS) >= 8.0f) ? 1.002f : 1.01f;
        const uint64_t n_nom = nn > 0 ? uint64_t(nn * margin) + 64 : 64;
        a.per = uint32_t(((n_nom + G - 1) / G + 63) / 64 * 64);
    }
    {   // stagger (see the kernel): only when an epoch is many samples per lane, i.e. when correlation, not the serial chain, fills it.
        // The dispatcher deals an XCD's workgroups round-robin over its CUs (measured: tools/trk_placement.py), so ordinals o and
        // o + CUs-per-XCD share a CU; channel `l` of an XCD owns ordinals [l G, (l + 1) G).  Channels that share a CU get different
        // classes (0, 1, 2: a greedy colouring — the conflict graph of five channels of twelve workgroups on 32 CUs is a 5-cycle,
        // so two classes do not do), class k starts k thirds of an epoch late.
        const uint32_t rows = a.per / uint32_t(TRK_PERSIST_THREADS);                 // samples per lane and epoch
        // (measured, round 6: the stagger alone buys NOTHING — 21.2 us per code period at BASELINE configs[4] with and without, plus
        // its own delay — because the co-tenants are not symmetric: see `fair_share`.  Off unless asked for; kept for experiments.)
        const int k = diag_int("GM_TRK_STAGGER_K", 0);                               // 10 ns ticks per row of samples and class
        a.stagger = rows >= 8 && G >= 2 && !packed ? rows * uint32_t(k < 0 ? 0 : k) : 0u;
        a.stagger_tab[0] = a.stagger_tab[1] = 0u;
        static int cus_cache[16] = {0};                     // per device, asked once (hipGetDeviceProperties is not a per-launch call)
        int dev = 0, cus = 32;
        if (hipGetDevice(&dev) == hipSuccess) {
            int& c = cus_cache[dev & 15];
            if (c == 0) {
                hipDeviceProp_t prop;
                c = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8) ? prop.multiProcessorCount / 8 : 32;
            }
            cus = c;
        }
        a.fair_mode = diag_int("GM_TRK_FAIR", 2);
        a.fair_share = (rows >= 8 && a.fair_mode != 0) ? cus : 0;
        if (a.stagger) {
            const int most = (cfg.n_channels + 7) / 8;
            for (int v = 0; v < 2; ++v) {
                const int count = most - v;
                if (count < 1 || count > 8) continue;
                int cls[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int l = 0; l < count; ++l) {           // greedy: the smallest class no earlier co-tenant has
                    bool used[4] = {false, false, false, false};
                    for (int m = 0; m < l; ++m) {
                        bool share = false;                  // some ordinal of l and some ordinal of m differ by a multiple of `cus`
                        for (int o = l * G; o < (l + 1) * G && !share; ++o)
                            for (int q = m * G; q < (m + 1) * G; ++q)
                                if ((o - q) % cus == 0) { share = true; break; }
                        if (share) used[cls[m]] = true;
                    }
                    cls[l] = !used[0] ? 0 : !used[1] ? 1 : !used[2] ? 2 : 3;
                    a.stagger_tab[v] |= uint32_t(cls[l]) << (2 * l);
                }
            }
        }
    }
    a.xchg = d_xchg; a.outs = d_outs;
    a.processed = d_processed; a.lost = d_lost; a.lost_prn = d_lost_prn; a.error_flag = d_error;
    const size_t lds = trk_persistent_lds(cfg);
    const dim3 grid(a.packed ? 8 * a.packed : trk_persistent_slots(cfg.n_channels) * G);     // channel slots: n_channels rounded up to the eight XCDs
    // compile-time arms / code-index mode / BOC: straight-line sample code
    const int key = (cfg.n_arms == 5 ? 4 : 0) | (cfg.code_index_mode == GM_CODE_INDEX_FIXED ? 2 : 0) | (cfg.boc11 ? 1 : 0);
    constexpr int T = TRK_PERSIST_THREADS;
#define GM_TRK_LAUNCH(A, M, B) \
    if (cfg.strict_libm) hipLaunchKernelGGL((trk_persistent_kernel<A, M, B, T, 1>), grid, dim3(T), lds, st, a); \
    else if (A == 3 && a.fair_share) hipLaunchKernelGGL((trk_persistent_kernel<A, M, B, T, 0, A == 3>), grid, dim3(T), lds, st, a); \
    else hipLaunchKernelGGL((trk_persistent_kernel<A, M, B, T, 0>), grid, dim3(T), lds, st, a); \
    break
    switch (key) {
        case 0: GM_TRK_LAUNCH(3, 0, 0);
        case 1: GM_TRK_LAUNCH(3, 0, 1);
        case 2: GM_TRK_LAUNCH(3, 1, 0);
        case 3: GM_TRK_LAUNCH(3, 1, 1);
        case 4: GM_TRK_LAUNCH(5, 0, 0);
        case 5: GM_TRK_LAUNCH(5, 0, 1);
        case 6: GM_TRK_LAUNCH(5, 1, 0);
        default: GM_TRK_LAUNCH(5, 1, 1);
    }
#undef GM_TRK_LAUNCH
}

// Workgroups of the persistent kernel one CU can hold at once (occupancy API for THIS instantiation and its dynamic LDS,
// capped by the design's TRK_PERSIST_WG_PER_CU): the exchange between the G workgroups of a channel needs all
// n_channels * G of them resident together.
int trk_persistent_blocks_per_cu(const TrkDevCfg& cfg) {
    const size_t lds = trk_persistent_lds(cfg);
    const int key = (cfg.n_arms == 5 ? 4 : 0) | (cfg.code_index_mode == GM_CODE_INDEX_FIXED ? 2 : 0) | (cfg.boc11 ? 1 : 0);
    constexpr int T = TRK_PERSIST_THREADS;
    int n = 0;
    hipError_t e = hipSuccess;
#define GM_TRK_OCC(A, M, B) \
    e = cfg.strict_libm ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trk_persistent_kernel<A, M, B, T, 1>, T, lds) \
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trk_persistent_kernel<A, M, B, T, 0>, T, lds); \
    break
    switch (key) {
        case 0: GM_TRK_OCC(3, 0, 0);
        case 1: GM_TRK_OCC(3, 0, 1);
        case 2: GM_TRK_OCC(3, 1, 0);
        case 3: GM_TRK_OCC(3, 1, 1);
        case 4: GM_TRK_OCC(5, 0, 0);
        case 5: GM_TRK_OCC(5, 0, 1);
        case 6: GM_TRK_OCC(5, 1, 0);
        default: GM_TRK_OCC(5, 1, 1);
    }
#undef GM_TRK_OCC
    if (e != hipSuccess || n < 1) n = 1;
    return n < TRK_PERSIST_WG_PER_CU ? n : TRK_PERSIST_WG_PER_CU;
}

template <int ARMS>
static void launch_trk_epoch_serial(hipStream_t st, const TrkDevCfg& cfg, const int8_t* d_codes, gm_trk_state* d_states,
                                    const TrkSrc& src, int slices, float* d_partials, uint8_t* d_ready, int mode,
                                    gm_trk_out* d_outs, uint8_t* d_processed, uint8_t* d_lost, uint8_t* d_lost_prn,
                                    float* d_terms, size_t terms_cap, int* d_error) {
    const int nch = src.only_channel >= 0 ? 1 : cfg.n_channels;
    const int ub = 64, ug = src.only_channel >= 0 ? 1 : (cfg.n_channels + ub - 1) / ub;
    hipLaunchKernelGGL(trk_terms_kernel<ARMS>, dim3(slices, nch), dim3(256), size_t(cfg.code_len), st, cfg, d_codes, d_states, src,
                       slices, d_terms, (unsigned long long)terms_cap, d_ready, d_error);
    hipLaunchKernelGGL(trk_serial_sum_kernel<ARMS>, dim3(nch), dim3(256), 0, st,
                       cfg, d_states, src, d_terms, (unsigned long long)terms_cap, d_ready, d_partials);
    hipLaunchKernelGGL(trk_update_kernel<ARMS>, dim3(ug), dim3(ub), 0, st, cfg, d_states, d_partials, d_ready,
                       1, mode, src.only_channel, src.linear, d_outs, d_processed, d_lost, d_lost_prn);
}

void launch_trk_epoch(hipStream_t st, const TrkDevCfg& cfg, const int8_t* d_codes, gm_trk_state* d_states,
                      const TrkSrc& src, int slices, float* d_partials, uint8_t* d_ready, int mode,
                      gm_trk_out* d_outs, uint8_t* d_processed, uint8_t* d_lost, uint8_t* d_lost_prn,
                      float* d_terms, size_t terms_cap, int* d_error) {
    if (cfg.strict_sum_order) {      // the reference's sequential sums (see trk_terms_kernel)
        if (cfg.n_arms == 5) launch_trk_epoch_serial<5>(st, cfg, d_codes, d_states, src, slices, d_partials, d_ready, mode, d_outs,
                                                        d_processed, d_lost, d_lost_prn, d_terms, terms_cap, d_error);
        else launch_trk_epoch_serial<3>(st, cfg, d_codes, d_states, src, slices, d_partials, d_ready, mode, d_outs, d_processed,
                                        d_lost, d_lost_prn, d_terms, terms_cap, d_error);
        return;
    }
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

// Results of a tracking call (a few KB) from the device block to PINNED HOST memory by a kernel of the library's own: gfx950 stores to
// host-mapped memory straight over the fabric, one launch, a fixed cost.  (The runtime's hipMemcpyAsync of the same bytes stalled the
// calling thread ~7 ms five times in a 100-block receiver loop on the ROCm 7.0 runtime a PyTorch process brings along —
// tools/receiver_time.py GM_TRK_TRACE_SLOW — and took three submissions where this takes one.)
__global__ __launch_bounds__(256) void trk_results_to_host_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t words,
                                                                  const uint32_t* __restrict__ src2, uint32_t* __restrict__ dst2, size_t words2) {
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < words + words2; i += size_t(gridDim.x) * 256) {
        if (i < words) __builtin_nontemporal_store(src[i], dst + i);
        else __builtin_nontemporal_store(src2[i - words], dst2 + (i - words));
    }
}
// a second range (the channel states behind the call's passes, ABI 7) travels in the same launch; d_src2 may be NULL
void launch_trk_results_to_host(hipStream_t st, const void* d_src, void* h_dst_pinned, size_t bytes, const void* d_src2, void* h_dst2_pinned, size_t bytes2) {
    const size_t words = (bytes + 3) / 4;                      // (the blocks are allocated in multiples of 4 bytes)
    const size_t words2 = d_src2 ? (bytes2 + 3) / 4 : 0;
    if (!(words + words2)) return;
    const size_t wg = (words + words2 + 255) / 256;
    const int blocks = int(wg < 64 ? wg : 64);
    hipLaunchKernelGGL(trk_results_to_host_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const uint32_t*>(d_src),
                       static_cast<uint32_t*>(h_dst_pinned), words, static_cast<const uint32_t*>(d_src2), static_cast<uint32_t*>(h_dst2_pinned), words2);
}

}  // namespace gm
