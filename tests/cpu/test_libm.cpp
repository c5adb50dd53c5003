// tests/cpu/test_libm.cpp — gnss-sdr-rs_amd/csrc/gm_libm.h against the host libm (glibc), bit for bit.
// The reference's f32::atan (do_tracking.rs:280) is this host function; the device runs gm::atanf_glibc.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "gm_libm.h"

int main() {
    unsigned long long bad = 0, n = 0;
    // every 61st f32 bit pattern (odd stride: all exponents, both signs, ~7.0e7 values) ...
    for (unsigned long long u = 0; u < 0x100000000ull; u += 61) {
        const float x = gm::f32_from_bits(uint32_t(u));
        const float a = gm::atanf_glibc(x), b = atanf(x);
        const bool same = gm::f32_bits(a) == gm::f32_bits(b) || (a != a && b != b);
        if (!same && bad++ < 10) printf("MISMATCH x=%a ours=%a libm=%a\n", x, a, b);
        ++n;
    }
    // ... and a dense sweep of the PLL's operating range |q/i| <= 4 (2^24 values)
    for (unsigned i = 0; i <= (1u << 24); ++i) {
        const float x = -4.0f + 8.0f * float(i) / float(1u << 24);
        const float a = gm::atanf_glibc(x), b = atanf(x);
        if (gm::f32_bits(a) != gm::f32_bits(b) && bad++ < 10) printf("MISMATCH x=%a ours=%a libm=%a\n", x, a, b);
        ++n;
    }
    printf("atanf_glibc: %llu arguments, %llu mismatches\n", n, bad);
    // div_const against IEEE division: divisors = the sample rates / code lengths / 2*pi the library divides by; dividends =
    // sample counts, code rates, angles (a deterministic LCG walk over their ranges)
    {
        const float divisors[] = {2.0e6f, 2.046e6f, 2.048e6f, 4.0e6f, 4.092e6f, 4.096e6f, 5.0e6f, 5.456e6f, 6.0e6f, 8.0e6f, 8.184e6f,
                                  8.192e6f, 1.0e7f, 1.2e7f, 1.5e7f, 1.6e7f, 1.6368e7f, 1.63676e7f, 2.5e7f, 3.8192e7f, 5.0e7f,
                                  1023.0f, 2046.0f, 4092.0f, 10230.0f, 511.0f, 6.28318530717958647692f};
        unsigned long long nd = 0, badd = 0;
        unsigned long long st = 0x9E3779B97F4A7C15ull;
        for (float y : divisors) {
            const float inv = 1.0f / y;
            for (int i = 0; i < 10000000; ++i) {
                st = st * 6364136223846793005ull + 1442695040888963407ull;
                const unsigned r = unsigned(st >> 33);
                float x;
                switch (i & 3) {
                    case 0: x = float(r % 300001u); break;                                  // sample counts
                    case 1: x = 1.0e6f + float(r % 100000u) * 0.5f; break;                  // code rates around 1.023 / 2.046 / 10.23 Mcps
                    case 2: x = gm::f32_from_bits(0x3f800000u + (r % 0x0c000000u)); break;  // 1 .. 1e7, every exponent
                    default: x = -3.2f + float(r % 6400001u) * 1.0e-6f; break;              // angles (atan's range, both signs)
                }
                const float a = gm::div_const(x, y, inv), b = x / y;
                const bool same = gm::f32_bits(a) == gm::f32_bits(b) || (a == 0.0f && b == 0.0f);   // only a zero's sign may differ
                if (!same && badd++ < 10) printf("div_const MISMATCH x=%a y=%a ours=%a ieee=%a\n", x, y, a, b);
                ++nd;
            }
        }
        printf("div_const: %llu operands, %llu mismatches\n", nd, badd);
        bad += badd;
    }
    // fmod_bounded against fmodf: y in {2*pi, code lengths}, x over (-4200 y, 4200 y) incl. the fallback range, both signs
    {
        const float ys[] = {6.28318530717958647692f, 1023.0f, 2046.0f, 4092.0f, 10230.0f, 511.0f};
        unsigned long long nf = 0, badf = 0;
        unsigned long long st = 0xD1B54A32D192ED03ull;
        for (float y : ys) {
            const float inv = 1.0f / y;
            for (int i = 0; i < 26000000; ++i) {
                st = st * 6364136223846793005ull + 1442695040888963407ull;
                const unsigned r = unsigned(st >> 32);
                float x;
                switch (i & 3) {
                    case 0: x = (float(r) * (1.0f / 4294967296.0f) * 2.0f - 1.0f) * 4200.0f * y; break;    // the whole admitted range and a little beyond
                    case 1: x = (float(r) * (1.0f / 4294967296.0f) * 2.0f - 1.0f) * 3.0f * y; break;       // the operating range (a few periods)
                    case 2: x = float(int(r % 8001u) - 4000) * y + (float(r >> 13) * (1.0f / 524288.0f) - 0.5f) * 1.0e-3f * y; break;   // around multiples of y
                    default: x = gm::f32_from_bits(r); break;                                             // any bit pattern (inf, NaN, denormals)
                }
                const float a = gm::fmod_bounded(x, y, inv), b = fmodf(x, y);
                const bool same = gm::f32_bits(a) == gm::f32_bits(b) || (a != a && b != b);
                if (!same && badf++ < 10) printf("fmod_bounded MISMATCH x=%a y=%a ours=%a libm=%a\n", x, y, a, b);
                ++nf;
            }
        }
        printf("fmod_bounded: %llu operands, %llu mismatches\n", nf, badf);
        bad += badf;
    }
    // spc_rate_bounds: every rate inside the interval must give n by the definition
    {
        const float fss[] = {2.048e6f, 4.0e6f, 4.096e6f, 5.0e6f, 8.0e6f, 1.0e7f, 1.6368e7f, 1.63676e7f, 2.5e7f, 3.8192e7f, 5.0e7f};
        const struct { float len, rate; } codes[] = {{1023.0f, 1.023e6f}, {2046.0f, 2.046e6f}, {4092.0f, 1.023e6f}, {10230.0f, 1.023e7f}};
        unsigned long long ns = 0, bads = 0, empty = 0;
        unsigned long long st = 0xA0761D6478BD642Full;
        for (float fs : fss)
            for (auto cd : codes) {
                const float n0 = gm::spc_definition(fs, cd.rate, cd.len);
                for (int dn = -40; dn <= 40; ++dn) {
                    const float n = n0 + float(dn);
                    if (!(n >= 2.0f && n < 8388608.0f)) continue;
                    float lo, hi;
                    gm::spc_rate_bounds(fs, cd.len, n, lo, hi);
                    if (!(lo <= hi)) { ++empty; continue; }       // (n so large that the margins meet: the caller falls back)
                    auto check = [&](float r) {
                        if (r < lo || r > hi) return;
                        ++ns;
                        if (gm::spc_definition(fs, r, cd.len) != n && bads++ < 10)
                            printf("spc_rate_bounds MISMATCH fs=%a len=%a n=%a rate=%a lo=%a hi=%a -> %a\n", fs, cd.len, n, r, lo, hi,
                                   gm::spc_definition(fs, r, cd.len));
                    };
                    check(lo); check(hi);
                    for (int k = 1; k <= 64; ++k) { check(gm::f32_from_bits(gm::f32_bits(lo) + k)); check(gm::f32_from_bits(gm::f32_bits(hi) - k)); }
                    for (int i = 0; i < 2000; ++i) {
                        st = st * 6364136223846793005ull + 1442695040888963407ull;
                        check(lo + (hi - lo) * (float(unsigned(st >> 40)) * (1.0f / 16777216.0f)));
                    }
                }
            }
        printf("spc_rate_bounds: %llu rates inside their intervals, %llu mismatches (%llu empty intervals)\n", ns, bads, empty);
        bad += bads;
    }
    // sincosf_glibc against the host's sinf / cosf, bit for bit: every 127th f32 bit pattern (all exponents, both signs, inf / NaN),
    // a dense sweep of the carrier's operating range (|phase| up to 2.7e5 rad: both reductions), and the neighbourhood of the
    // range boundaries (2^-12, pi/4, 120)
    {
        unsigned long long nsc = 0, badsc = 0;
        auto check = [&](float x) {
            float s, c;
            gm::sincosf_glibc(x, s, c);
            const float hs = sinf(x), hc = cosf(x);
            const bool same = (gm::f32_bits(s) == gm::f32_bits(hs) || (s != s && hs != hs)) && (gm::f32_bits(c) == gm::f32_bits(hc) || (c != c && hc != hc));
            if (!same && badsc++ < 10) printf("sincosf_glibc MISMATCH x=%a ours=(%a, %a) libm=(%a, %a)\n", x, s, c, hs, hc);
            ++nsc;
        };
        for (unsigned long long u = 0; u < 0x100000000ull; u += 127) check(gm::f32_from_bits(uint32_t(u)));
        for (unsigned i = 0; i <= (1u << 26); ++i) check(float(-270000.0 + 540000.0 * double(i) / double(1u << 26)));
        const float edges[] = {0x1p-12f, 0x1.921fb6p-1f, 120.0f, 0x1p-126f, 1.0f, 8.0f, 0x1p23f, 0x1p24f, 0x1p31f, 0x1p100f};
        for (float e : edges)
            for (int k = -2000; k <= 2000; ++k) { const float v = gm::f32_from_bits(gm::f32_bits(e) + k); check(v); check(-v); }
        printf("sincosf_glibc: %llu arguments, %llu mismatches\n", nsc, badsc);
        bad += badsc;
    }
    // sincos_cw against the f64 functions of the same f32 argument: |x| <= 131072 (the fast path admits 1e5), 2^25 arguments
    double worst = 0.0;
    unsigned long long ns = 0, lastbit = 0;
    for (unsigned i = 0; i <= (1u << 25); ++i) {
        const float x = float(-131072.0 + 262144.0 * double(i) / double(1u << 25));
        float sv, cv;
        gm::sincos_cw(x, sv, cv);
        const double es = fabs(double(sv) - sin(double(x))), ec = fabs(double(cv) - cos(double(x)));
        if (es > worst) worst = es;
        if (ec > worst) worst = ec;
        if (gm::f32_bits(sv) != gm::f32_bits(sinf(x))) ++lastbit;
        if (gm::f32_bits(cv) != gm::f32_bits(cosf(x))) ++lastbit;
        ns += 2;
    }
    const double ulp1 = 5.9604644775390625e-08;   // 2^-24: ulp of results in [0.5, 1)
    printf("sincos_cw: %llu values, max |error| = %.3g = %.2f ulp(1); differs from the host's sinf/cosf in the last bit on %.2f %%\n",
           ns, worst, worst / ulp1, 100.0 * double(lastbit) / double(ns));
    if (worst > 1.6 * ulp1) { printf("sincos_cw: error bound exceeded\n"); bad++; }
    return bad ? 1 : 0;
}
