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

// sinf(y) and cosf(y) as glibc 2.35 computes them on an x86-64 host with FMA (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c and
// sincosf.h built as the multiarch *_fma variants, which the ifunc resolver of libm.so.6 picks on every CPU with FMA and
// AVX2 — both boxes this repository runs on): the reference's `phase.cos()` / `phase.sin()` (do_tracking.rs:234-235) are
// these two functions.  All arithmetic is f64: the argument is reduced to |x| <= pi/4 with a quadrant count n — below 120
// by one multiply and one fused multiply-subtract (reduce_fast), above by the 96-bit fixed-point product with 4/pi
// (reduce_large) — then an odd (sine) or even (cosine) polynomial in x^2, and ONE rounding to f32 at the end.  The fused
// multiply-adds below are the ones the compiler contracted in the shipped binary (read from its disassembly: every
// `a + b*c` of the source became one FMA, the plain products stayed products); a restatement without them differs from
// the host's functions in the last bit of the f64 result, which now and then decides the f32 rounding.
// tests/cpu/test_libm.cpp compares both results with the host's sinf / cosf bit for bit (all three argument ranges).
// Constants: __sincosf_table[0] / [1] and __inv_pio4 of sincosf_data.c, as hexadecimal literals.
struct SincosfPoly { double c0, c1, c2, c3, c4, s1, s2, s3; };
GM_LIBM_HD double sincosf_sinpoly(double xs, double x2, const SincosfPoly& p) {
    const double x3 = x2 * xs;
    const double s1 = __builtin_fma(p.s3, x2, p.s2);
    const double x7 = x2 * x3;
    const double s = __builtin_fma(x3, p.s1, xs);
    return __builtin_fma(s1, x7, s);
}
GM_LIBM_HD double sincosf_cospoly(double x2, const SincosfPoly& p) {
    const double x4 = x2 * x2;
    const double c1 = __builtin_fma(p.c1, x2, p.c0);
    const double c2 = __builtin_fma(p.c4, x2, p.c3);
    const double x6 = x2 * x4;
    const double c = __builtin_fma(x4, p.c2, c1);
    return __builtin_fma(c2, x6, c);
}
GM_LIBM_HD void sincosf_glibc(float y, float& sn, float& cs) {
    static constexpr uint32_t inv_pio4[24] = {0xa2u,       0xa2f9u,     0xa2f983u,   0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u,
                                              0x6e4e4415u, 0x4e441529u, 0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u,
                                              0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u, 0x34ddc0dbu, 0xddc0db62u,
                                              0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                 C4 = 0x1.99343027bf8c3p-16, S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    const uint32_t yi = f32_bits(y), top = (yi >> 20) & 0x7ffu;      // abstop12
    double x = double(y);
    if (top < 0x3f4u) {                                               // |y| < pi/4
        const double x2 = x * x;
        if (top < 0x398u) { sn = y; cs = 1.0f; return; }              // |y| < 2^-12
        const SincosfPoly p = {C0, C1, C2, C3, C4, S1, S2, S3};
        sn = float(sincosf_sinpoly(x, x2, p));
        cs = float(sincosf_cospoly(x2, p));
        return;
    }
    int n, q;                                                         // quadrant count (parity picks the polynomial), sign / table selector
    if (top < 0x42fu) {                                               // |y| < 120: reduce_fast
        const double r = x * 0x1.45F306DC9C883p+23;                   // 2/pi * 2^24
        n = (int32_t(r) + 0x800000) >> 24;                            // truncating conversion, arithmetic shift
        x = __builtin_fma(-double(n), 0x1.921FB54442D18p0, x);
        q = n;
    } else if (top < 0x7f8u) {                                        // finite: reduce_large
        const uint32_t* arr = &inv_pio4[(yi >> 26) & 15u];
        const int shift = int((yi >> 23) & 7u);
        uint32_t m = (yi & 0x7fffffu) | 0x800000u;
        m <<= shift;
        uint64_t res0 = uint64_t(uint32_t(m * arr[0]));
        const uint64_t res1 = uint64_t(m) * arr[4], res2 = uint64_t(m) * arr[8];
        res0 = (res2 >> 32) | (res0 << 32);
        res0 += res1;
        const uint64_t nn = (res0 + (1ull << 61)) >> 62;
        res0 -= nn << 62;
        x = double(int64_t(res0)) * 0x1.921FB54442D18p-62;
        n = int(nn);
        q = n + int(yi >> 31);
    } else {                                                          // inf / NaN
        sn = cs = y - y;
        return;
    }
    const double sg = ((q + 1) & 2) ? -1.0 : 1.0;                     // sign[q & 3] = {1, -1, -1, 1}
    const double k = (q & 2) ? -1.0 : 1.0;                            // __sincosf_table[1]: the cosine coefficients negated
    const SincosfPoly p = {k * C0, k * C1, k * C2, k * C3, k * C4, S1, S2, S3};
    const double x2 = x * x;
    const double a = sincosf_sinpoly(x * sg, x2, p), b = sincosf_cospoly(x2, p);
    sn = float((n & 1) ? b : a);                                      // sinf: sinf_poly(x * s, x * x, p, n)
    cs = float((n & 1) ? a : b);                                      // cosf: sinf_poly(x * s, x * x, p, n ^ 1)
}


}  // namespace gm
