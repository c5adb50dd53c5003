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
