// synth_xoshiro.cpp — the deterministic synthetic-scene generator of SURVEY.md §8 d2, in C++: splitmix64-seeded
// xoshiro256**, Box-Muller normals, the acquisition / tracking signal model of gnss-sdr-rs_amd/synth.py.  Host code only
// (g++; no HIP): it makes INPUTS for bench.py and the tests, it is not part of the product library and computes nothing of
// the hot path.  Built as gnss-sdr-rs_amd/lib/libgm_synth.so by build.py; Python binding: synth.py (generator="xoshiro").
//
//   seed   = 0x6E5553445200 + config_id                      
//   stream = a small integer per use (noise, satellite parameters, stand-in codes): the four state words of a stream are
//            four splitmix64 outputs from  seed + stream * 0x9E3779B97F4A7C15
//   normal = Box-Muller on two 53-bit uniforms, u1 in (0, 1], u2 in [0, 1): r = sqrt(-2 ln u1), (r cos 2 pi u2, r sin 2 pi u2)
//   x[n]   = sum_s A_s * c_s(floor((n - k_s) * code_rate / fs) mod L) * d_s * exp(+j (2 pi (f_if + f_s) n / fs + phi_s)) + w[n],
//            w ~ CN(0, 2 sigma^2);  real_only: sigma * N(0, 1) + sum_s A_s sqrt(2) c_s cos(.)
//   quantise: round half to even (rint), clip to [-127, 127]
// Everything is evaluated in double with the C library's log / sqrt / sin / cos, one sample after the other: the bytes depend
// on the seed and on libm alone (both boxes run one image), not on a numpy version.
#include <cmath>
#include <cstddef>
#include <cstdint>

extern "C" {

uint64_t gs_splitmix64_next(uint64_t* state) {
    uint64_t z = (*state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

uint64_t gs_xoshiro_next(uint64_t s[4]) {      // xoshiro256** 1.0 (Blackman & Vigna)
    const uint64_t result = rotl(s[1] * 5, 7) * 9;
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return result;
}

void gs_stream_seed(uint64_t seed, uint64_t stream, uint64_t s[4]) {
    uint64_t sm = seed + stream * 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 4; ++i) s[i] = gs_splitmix64_next(&sm);
}

double gs_uniform(uint64_t s[4]) { return double(gs_xoshiro_next(s) >> 11) * 0x1p-53; }                   // [0, 1)

// integer in [lo, hi): multiply-shift on the top 32 bits (hi - lo < 2^32); the tiny bias is irrelevant for scene parameters
int64_t gs_integer(uint64_t s[4], int64_t lo, int64_t hi) {
    const uint64_t span = uint64_t(hi - lo);
    return lo + int64_t(((gs_xoshiro_next(s) >> 32) * span) >> 32);
}

static inline void normal_pair(uint64_t s[4], double& a, double& b) {
    const double u1 = double((gs_xoshiro_next(s) >> 11) + 1) * 0x1p-53;   // (0, 1]
    const double u2 = double(gs_xoshiro_next(s) >> 11) * 0x1p-53;         // [0, 1)
    const double r = std::sqrt(-2.0 * std::log(u1)), th = 6.283185307179586476925286766559 * u2;
    a = r * std::cos(th);
    b = r * std::sin(th);
}

// n standard normals from the stream's current state (pairs: an odd n drops the last value's partner)
void gs_fill_normal(uint64_t s[4], double* out, size_t n) {
    for (size_t i = 0; i < n; i += 2) {
        double a, b;
        normal_pair(s, a, b);
        out[i] = a;
        if (i + 1 < n) out[i + 1] = b;
    }
}

typedef struct {
    int32_t prn_row;
    int32_t has_bits;          // data_bits given
    double cn0_dbhz, doppler_hz, code_start, phase;
    const double* data_bits;   // +-1 per 20 code periods (50 bit/s), or NULL
    int64_t n_bits, bit_edge_ms;
} gs_sat;

// The scene of synth.make_scene.  Noise: stream `noise_stream` of `seed`, one pair per complex sample (real_only: one value
// per sample).  out_re / out_im: n_samples doubles each (out_im all zero when real_only); quantised when `quantize`.
int gs_make_scene(const int8_t* code_table, int32_t n_rows, int32_t code_len, double fs, double f_if, size_t n_samples,
                  const gs_sat* sats, int32_t n_sats, double sigma, uint64_t seed, uint64_t noise_stream, int32_t real_only,
                  double code_rate, int32_t quantize, int64_t bit_flip_at, double* out_re, double* out_im) {
    if (!code_table || !out_re || !out_im || code_len <= 0 || !(fs > 0)) return -1;
    for (int32_t k = 0; k < n_sats; ++k)
        if (sats[k].prn_row < 0 || sats[k].prn_row >= n_rows) return -2;
    uint64_t s[4];
    gs_stream_seed(seed, noise_stream, s);
    if (real_only) {
        gs_fill_normal(s, out_re, n_samples);
        for (size_t i = 0; i < n_samples; ++i) { out_re[i] *= sigma; out_im[i] = 0.0; }
    } else {
        for (size_t i = 0; i < n_samples; ++i) {
            double a, b;
            normal_pair(s, a, b);
            out_re[i] = sigma * a;
            out_im[i] = sigma * b;
        }
    }
    const double two_pi = 6.283185307179586476925286766559;
    for (int32_t k = 0; k < n_sats; ++k) {
        const gs_sat& sv = sats[k];
        const double amp = sigma * std::sqrt(2.0 * std::pow(10.0, sv.cn0_dbhz / 10.0) / fs);
        const int8_t* row = code_table + size_t(sv.prn_row) * size_t(code_len);
#pragma omp parallel for schedule(static)      // every sample is independent: the bytes do not depend on the thread count
        for (size_t i = 0; i < n_samples; ++i) {
            const double t = double(i);
            const double cp = std::floor((t - sv.code_start) * code_rate / fs);
            int64_t chip = int64_t(cp) % code_len;
            if (chip < 0) chip += code_len;
            double c = double(row[chip]);
            if (bit_flip_at >= 0 && int64_t(i) >= bit_flip_at) c = -c;
            if (sv.has_bits && sv.data_bits && sv.n_bits > 0) {
                const int64_t period = int64_t(std::floor((t - sv.code_start) * (code_rate / double(code_len)) / fs));
                int64_t q = period - sv.bit_edge_ms;
                int64_t kb = q >= 0 ? q / 20 : -((-q + 19) / 20);      // floor division
                int64_t idx = kb % sv.n_bits;
                if (idx < 0) idx += sv.n_bits;
                c *= sv.data_bits[idx];
            }
            const double ph = two_pi * (f_if + sv.doppler_hz) * t / fs + sv.phase;
            if (real_only) out_re[i] += amp * 1.4142135623730951 * c * std::cos(ph);
            else { out_re[i] += amp * c * std::cos(ph); out_im[i] += amp * c * std::sin(ph); }
        }
    }
    if (quantize)
        for (size_t i = 0; i < n_samples; ++i) {
            double r = std::nearbyint(out_re[i]), q = std::nearbyint(out_im[i]);
            out_re[i] = r < -127.0 ? -127.0 : (r > 127.0 ? 127.0 : r);
            out_im[i] = q < -127.0 ? -127.0 : (q > 127.0 ? 127.0 : q);
        }
    return 0;
}

}  // extern "C"
