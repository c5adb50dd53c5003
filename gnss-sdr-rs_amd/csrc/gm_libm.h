// gm_libm.h — the libm functions of the reference's scalar tracking epilogue, restated so that the device rounds like
// the host the reference runs on.
//
// run_loop_filters (src/tracking/do_tracking.rs:279-302) calls f32::atan, which Rust's std forwards to the platform
// libm's atanf: on linux-gnu that is glibc's sysdeps/ieee754/flt-32/s_atanf.c (the fdlibm float kernel: four-interval
// argument reduction + an 11-term odd/even polynomial, plain f32 multiplies and adds, no FMA; glibc 2.35 ships no
// multiarch variant of it).  It is accurate to < 1 ulp but NOT correctly rounded, so another < 1 ulp implementation (the
// device's ocml atanf) differs from it by an ulp on a fraction of arguments — enough to make carrier_error / carrier_nco /
// carrier_freq drift apart from the reference's by a few ulps per epoch.  The function below performs glibc's operations
// in glibc's order (compile with -ffp-contract=off; divisions are IEEE), so loop state stays bit-identical.
// tests/cpu/test_libm.cpp checks it against the host's atanf on 2^26 arguments, bit for bit.
//
// Host/device portable on purpose (same idea as fft_core.h): g++ validates it without a GPU.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GM_LIBM_HD __host__ __device__ __forceinline__
#else
#define GM_LIBM_HD inline __attribute__((always_inline))
#endif

namespace gm {

GM_LIBM_HD uint32_t f32_bits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
GM_LIBM_HD float f32_from_bits(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

// IEEE-754 correctly rounded f32 division on both sides
GM_LIBM_HD float div_rn(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fdiv_rn(a, b);
#else
    return a / b;
#endif
}

// IEEE-754 correctly rounded f32 square root on both sides.  NOT __fsqrt_rn: without OCML_BASIC_ROUNDED_OPERATIONS hipcc's
// header maps that to __ocml_native_sqrt_f32 (v_sqrt_f32, 1 ulp); __builtin_sqrtf is lowered to the correctly rounded
// sequence under -fhip-fp32-correctly-rounded-divide-sqrt (hipcc's default, also passed explicitly by build.py).
GM_LIBM_HD float sqrt_rn(float a) { return __builtin_sqrtf(a); }

GM_LIBM_HD float atanf_glibc(float x) {
    // coefficients of s_atanf.c (atanhi / atanlo / aT[0..10]) as the bit patterns its decimal literals round to (aT[0] =
    // 3.3333334327e-01 is 0x3eaaaaab: the 0x3eaaaaaa in that file's comment is off by one)
    const float hi0 = f32_from_bits(0x3eed6338u), hi1 = f32_from_bits(0x3f490fdau), hi2 = f32_from_bits(0x3f7b985eu),
                hi3 = f32_from_bits(0x3fc90fdau);
    const float lo0 = f32_from_bits(0x31ac3769u), lo1 = f32_from_bits(0x33222168u), lo2 = f32_from_bits(0x33140fb4u),
                lo3 = f32_from_bits(0x33a22168u);
    const float a0 = f32_from_bits(0x3eaaaaabu), a1 = f32_from_bits(0xbe4ccccdu), a2 = f32_from_bits(0x3e124925u),
                a3 = f32_from_bits(0xbde38e38u), a4 = f32_from_bits(0x3dba2e6eu), a5 = f32_from_bits(0xbd9d8795u),
                a6 = f32_from_bits(0x3d886b35u), a7 = f32_from_bits(0xbd6ef16bu), a8 = f32_from_bits(0x3d4bda59u),
                a9 = f32_from_bits(0xbd15a221u), a10 = f32_from_bits(0x3c8569d7u);
    const uint32_t hx = f32_bits(x), ix = hx & 0x7fffffffu;
    const bool neg = (hx >> 31) != 0;
    if (ix >= 0x4c000000u) {                         // |x| >= 2^25
        if (ix > 0x7f800000u) return x + x;          // NaN
        return neg ? -hi3 - lo3 : hi3 + lo3;
    }
    int id;
    float ahi = 0.0f, alo = 0.0f;
    if (ix < 0x3ee00000u) {                          // |x| < 0.4375
        if (ix < 0x31000000u) return x;              // |x| < 2^-29
        id = -1;
    } else {
        x = f32_from_bits(ix);                       // fabsf
        if (ix < 0x3f980000u) {                      // |x| < 1.1875
            if (ix < 0x3f300000u) { id = 0; ahi = hi0; alo = lo0; x = div_rn(2.0f * x - 1.0f, 2.0f + x); }   // 7/16 <= |x| < 11/16
            else { id = 1; ahi = hi1; alo = lo1; x = div_rn(x - 1.0f, x + 1.0f); }                           // 11/16 <= |x| < 19/16
        } else {
            if (ix < 0x401c0000u) { id = 2; ahi = hi2; alo = lo2; x = div_rn(x - 1.5f, 1.0f + 1.5f * x); }   // |x| < 2.4375
            else { id = 3; ahi = hi3; alo = lo3; x = div_rn(-1.0f, x); }                                     // 2.4375 <= |x| < 2^25
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
    const float s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = ahi - ((x * (s1 + s2) - alo) - x);
    return neg ? -r : r;
}


// x / y for a divisor whose correctly rounded reciprocal inv = RN(1/y) is known: q0 = x*inv,
// q = fma(fma(-q0, y, x), inv, q0) is the correctly rounded quotient (Markstein) while the quotient stays in the normal
// range and y's significand is not all ones.  Used where the divisor is a configuration constant (fs, the code length,
// 2*pi) and the dividend a sample count, a code rate or an angle; tests/cpu/test_libm.cpp compares it with IEEE division
// on 2.6e8 operands (21 sample rates, 5 code lengths, 2*pi), bit for bit.
GM_LIBM_HD float div_const(float x, float y, float inv) {
    const float q0 = x * inv;
    return __builtin_fmaf(__builtin_fmaf(-q0, y, x), inv, q0);
}

// fmodf(x, y) for a positive constant y with inv = RN(1/y); exact, as fmodf is (the remainder is representable, so every
// correct algorithm returns the same bits).  For y <= |x| < 4096 y: q = rint(|x| * inv) is the integer quotient or one
// more, r = fma(-q, y, |x|) is exact (a multiple of ulp(y) of magnitude below y), one conditional + y brings it into [0, y),
// and the sign is x's.  Outside that range (and for inf / NaN): the library's fmodf.  (The library form is ~35 instructions
// with frexp/ldexp and three branches on the device; this is 9.)  tests/cpu/test_libm.cpp: equal to fmodf on 1.6e8 operands.
GM_LIBM_HD float fmod_bounded(float x, float y, float inv) {
    const float ax = __builtin_fabsf(x);
    if (!(ax < 4096.0f * y)) {
#if defined(__HIP_DEVICE_COMPILE__)
        // y through a scalar register, opaque: keeps the library form's loop-invariant set-up inside this (rare) branch
        // instead of in VGPRs for the whole kernel
        uint32_t u = __builtin_amdgcn_readfirstlane(f32_bits(y));
        asm volatile("" : "+s"(u));
        return fmodf(x, f32_from_bits(u));
#else
        return __builtin_fmodf(x, y);
#endif
    }
    const float q = __builtin_rintf(ax * inv);
    float r = __builtin_fmaf(-q, y, ax);
    r = r < 0.0f ? r + y : r;
    r = ax < y ? ax : r;
    return __builtin_copysignf(r, x);
}

// The tracking epoch length n = round(fs / (code_rate / len)) (ca_code.rs:13-16 through do_tracking.rs:165-166) changes
// only when the code rate crosses one of the rates at which the quotient passes a half-integer; between two epochs it
// almost never does.  spc_rate_bounds gives an interval of rates [lo, hi] that PROVABLY keeps the f32 evaluation at n
// (n < 2^23): the exact quotient F / r, F = fs * len, lies in [(n - 0.5)(1 + 9e-7), (n + 0.5)(1 - 9e-7)] for every r in
// the interval (the bounds are formed in f32 with a 2e-6 inward margin against their own < 3e-7 of rounding), and the
// reference's two correctly rounded divisions move it by < 1.3e-7 relative — so roundf lands on n.  A rate outside the
// interval proves nothing: the caller then evaluates the definition.  tests/cpu/test_libm.cpp checks the claim on the
// end points, their float neighbours inside, and random rates, for the library's sample rates and code lengths.
GM_LIBM_HD void spc_rate_bounds(float fs, float lenf, float n, float& lo, float& hi) {
    const float F = fs * lenf;
    lo = div_rn(F, n + 0.5f) * 1.000002f;
    hi = div_rn(F, n - 0.5f) * 0.999998f;
}
// the definition, in the reference's order (f32, two IEEE divisions, round half away from zero)
GM_LIBM_HD float spc_definition(float fs, float code_rate, float lenf) { return __builtin_roundf(div_rn(fs, div_rn(code_rate, lenf))); }

// sin and cos of an f32 phase of moderate size (|x| < 1.3e5 rad: k = rint(x * 2/pi) < 2^17), for the carrier wipe-off of the
// tracking correlators (do_tracking.rs:243-246 calls f32::cos / f32::sin = glibc's cosf / sinf, < 1 ulp, not correctly
// rounded).  Cody-Waite reduction in f32 with the first product split exactly: k*c1 = ph + pl (one multiply, one fma),
// x - ph is exact (Sterbenz: ph is within a factor two of x whenever k != 0), and the small terms pl + k*c2 are gathered
// before the single rounding that forms r — the reduced argument is as good as one rounded from f64 (pi/2 = c1 + c2 to 48
// bits; k*c3 < 2e-10 is dropped).  Then the Cephes sinf / cosf minimax cores on |r| <= pi/4, which set the accuracy:
// tests/cpu/test_libm.cpp measures max |error| = 1.56 * 2^-24 against the f64 functions on 6.7e7 arguments in |x| <= 131072 (the same as with
// the argument reduced in f64, which this replaces: 99.999 % of results are identical) and a last-bit difference from the
// host's sinf / cosf on 25 % of arguments.
// The correlator sums these values enter are compared with the reference's under a tolerance; the loop state
// downstream is where a bit-exact libm matters, and that uses atanf_glibc above.
GM_LIBM_HD void sincos_cw(float x, float& s, float& c) {
    const float c1 = f32_from_bits(0x3fc90fdbu), c2 = f32_from_bits(0xb33bbd2eu);   // pi/2 = 1.5707963705 - 4.3711388e-8 - ...
    const float k = __builtin_rintf(x * 0.636619747f);
    const float ph = k * c1;
    const float pl = __builtin_fmaf(k, c1, -ph);          // k*c1 = ph + pl exactly
    const float r1 = x - ph;                              // exact
    const float r = r1 - __builtin_fmaf(k, c2, pl);
    const float z = r * r;
    float sp = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    sp = __builtin_fmaf(sp, z, -1.6666654611e-1f);
    const float sr = __builtin_fmaf(sp * z, r, r);
    float cp = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    cp = __builtin_fmaf(cp, z, 4.166664568298827e-2f);
    const float cr = __builtin_fmaf(cp * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    // quadrant: swap on odd k, then the signs as bit flips (bit 1 of k for sin, of k + 1 for cos) — integer operations
    // issue at the full rate on gfx950, compares and selects at half
    const uint32_t q = uint32_t(int(k));
    const float sv = (q & 1u) ? cr : sr, cv = (q & 1u) ? sr : cr;
    s = f32_from_bits(f32_bits(sv) ^ ((q << 30) & 0x80000000u));
    c = f32_from_bits(f32_bits(cv) ^ (((q + 1u) << 30) & 0x80000000u));
}

}  // namespace gm
