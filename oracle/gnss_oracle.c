/*
 * gnss_oracle.c — CPU ORACLE (test infrastructure, NOT product code).  See gnss_oracle.h.
 *
 * Restates, in plain C with f32 arithmetic in the reference's operation order:
 *   src/utilities/ca_code.rs:12-27, src/constants/gps_ca_constants.rs (regenerated),
 *   src/acquisition/doppler_shift.rs:10-58, src/acquisition/do_acquisition.rs:39-74,130-238,302-313,
 *   src/tracking/do_tracking.rs:52-71,118-327, src/utilities/multicast_ring_buffer.rs:46-129,
 *   src/fft.rs:5-56.
 * Compile with -ffp-contract=off: rustc never fuses a*b+c.
 * libm calls (cosf/sinf/atanf/sqrtf/fmodf/roundf/floorf) are glibc's, which is what Rust's
 * f32::{cos,sin,atan,sqrt,%,round,floor} lower to on x86_64-unknown-linux-gnu.
 */
#include "gnss_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI_F 3.14159265358979323846f /* std::f32::consts::PI */

/* ------------------------------------------------------------------------------------------
 * C/A code: IS-GPS-200 G1/G2 generator.  Reproduces GPS_CA_CODE_32_PRN
 * (src/constants/gps_ca_constants.rs:1-1346): row r <-> PRN r+1, chip bit 1 -> +1, 0 -> -1.
 * ------------------------------------------------------------------------------------------ */
static const uint16_t G2_DELAY[32] = { /* IS-GPS-200 Table 3-Ia, code delay chips */
    5,   6,   7,   8,   17,  18,  139, 140, 141, 251, 252, 254, 255, 256, 257, 258,
    469, 470, 471, 472, 473, 474, 509, 512, 513, 514, 515, 516, 859, 860, 861, 862};

static void lfsr1023(const int *taps, int ntaps, uint8_t out[1023]) {
    uint8_t reg[10];
    for (int i = 0; i < 10; i++) reg[i] = 1;
    for (int i = 0; i < 1023; i++) {
        out[i] = reg[9];
        uint8_t fb = 0;
        for (int t = 0; t < ntaps; t++) fb ^= reg[taps[t] - 1];
        for (int j = 9; j > 0; j--) reg[j] = reg[j - 1];
        reg[0] = fb;
    }
}

int orc_ca_code_row(int row, int8_t out[1023]) {
    if (row < 0 || row > 31) return -1;
    static const int g1_taps[] = {3, 10};
    static const int g2_taps[] = {2, 3, 6, 8, 9, 10};
    uint8_t g1[1023], g2[1023];
    lfsr1023(g1_taps, 2, g1);
    lfsr1023(g2_taps, 6, g2);
    int delay = G2_DELAY[row];
    for (int i = 0; i < 1023; i++) {
        uint8_t bit = g1[i] ^ g2[(i + 1023 - delay) % 1023];
        out[i] = bit ? 1 : -1;
    }
    return 0;
}

/* src/utilities/ca_code.rs:13-16 */
size_t orc_num_samples_per_code(float code_rate, float fs) {
    float v = roundf(fs / (code_rate / ORC_GPS_L1_CA_CODE_LENGTH_CHIPS));
    if (!(v > 0.0f)) return 0; /* `as usize` saturates negatives/NaN to 0 */
    return (size_t)v;
}

/* src/utilities/ca_code.rs:12-27 */
long orc_generate_ca_code_samples(int prn, float code_rate, float fs, int8_t *out, size_t cap) {
    if (prn < 1 || prn > 32) return -1; /* `prn as usize - 1` underflow / table OOB -> panic */
    int8_t code[1023];
    orc_ca_code_row(prn - 1, code);
    size_t n = orc_num_samples_per_code(code_rate, fs);
    for (size_t i = 0; i < n; i++) {
        float f = floorf(((float)i * code_rate) / fs); /* :19  x as f32 * code_rate / f_sampling */
        size_t ind = (f > 0.0f) ? (size_t)f : 0;
        if (ind >= 1023) return -1; /* ca_code[ind] out of bounds -> panic */
        if (i < cap) out[i] = code[ind];
    }
    return (long)n;
}

/* ------------------------------------------------------------------------------------------
 * Doppler wipe-off table and multiply (src/acquisition/doppler_shift.rs)
 * ------------------------------------------------------------------------------------------ */
float orc_doppler_table_new(float f_if, float doppler_hz, float fs, size_t n, orc_c32 *table) {
    float carr_freq = f_if + doppler_hz;               /* :13 */
    float phase_step = 2.0f * PI_F * carr_freq / fs;   /* :14  ((2*PI)*carr)/fs */
    for (size_t i = 0; i < n; i++) {
        float phase = (float)i * phase_step;           /* :17 */
        table[i].re = cosf(phase);                     /* :18 */
        table[i].im = -sinf(phase);
    }
    return carr_freq;                                   /* :20 stores IF + Doppler */
}

void orc_apply_doppler_shift(const orc_c32 *s, const orc_c32 *t, orc_c32 *out, size_t n) {
    size_t chunks = n / 4;                              /* :26 — the tail n%4 is left untouched */
    for (size_t i = 0; i < chunks * 4; i++) {
        float a = s[i].re, b = s[i].im, c = t[i].re, d = t[i].im;
        /* multiply_simd_block :43-58: first = [a*c, a*d]; second = [b*d*(-1), b*c*(+1)] */
        float re1 = a * c, im1 = a * d;
        float re2 = (b * d) * -1.0f, im2 = (b * c) * 1.0f;
        out[i].re = re1 + re2;
        out[i].im = im1 + im2;
    }
}

/* ------------------------------------------------------------------------------------------
 * FFT: f32 mixed-radix Stockham autosort, unnormalised, any n (prime factors handled by an
 * O(r^2) butterfly).  Restates the DFT that rustfft 6.1.0 computes at do_acquisition.rs:137,182,188
 * (forward = sum x[n] e^{-j2pi kn/N}; inverse = e^{+...}; no 1/N).  Not bit-identical to rustfft.
 * ------------------------------------------------------------------------------------------ */
#define ORC_MAX_PASS 40
struct orc_fft_plan {
    size_t n;
    int inverse;
    int npass;
    int radix[ORC_MAX_PASS];
    orc_c32 *tw[ORC_MAX_PASS];    /* per pass: p*(R-1) twiddles, tw[k*(R-1)+r-1] = w_{pR}^{r k} */
    orc_c32 *roots[ORC_MAX_PASS]; /* per pass: R roots w_R^j (generic radix) */
    orc_c32 *scratch;
};

static inline orc_c32 cmul(orc_c32 a, orc_c32 b) {
    orc_c32 r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
    return r;
}
static inline orc_c32 cadd(orc_c32 a, orc_c32 b) { orc_c32 r = {a.re + b.re, a.im + b.im}; return r; }
static inline orc_c32 csub(orc_c32 a, orc_c32 b) { orc_c32 r = {a.re - b.re, a.im - b.im}; return r; }
/* multiply by -j (forward) or +j (inverse) */
static inline orc_c32 cmulj(orc_c32 a, int inverse) {
    orc_c32 r;
    if (!inverse) { r.re = a.im; r.im = -a.re; } else { r.re = -a.im; r.im = a.re; }
    return r;
}

orc_fft_plan *orc_fft_plan_create(size_t n, int inverse) {
    if (n == 0) return NULL;
    orc_fft_plan *pl = (orc_fft_plan *)calloc(1, sizeof(*pl));
    pl->n = n;
    pl->inverse = inverse ? 1 : 0;
    size_t m = n;
    int np = 0, e2 = 0;
    while ((m & 1) == 0) { m >>= 1; e2++; }
    /* odd radices first (ascending primes), then 4s, then a final 2 */
    for (size_t f = 3; f * f <= m; f += 2)
        while (m % f == 0) { pl->radix[np++] = (int)f; m /= f; }
    if (m > 1) pl->radix[np++] = (int)m;
    for (int i = 0; i < e2 / 2; i++) pl->radix[np++] = 4;
    if (e2 & 1) pl->radix[np++] = 2;
    if (np == 0) pl->radix[np++] = 1; /* n == 1 */
    pl->npass = np;
    const double sign = inverse ? 1.0 : -1.0;
    size_t p = 1;
    for (int s = 0; s < np; s++) {
        int R = pl->radix[s];
        pl->tw[s] = (orc_c32 *)malloc(sizeof(orc_c32) * p * (size_t)(R - 1) + 8);
        for (size_t k = 0; k < p; k++)
            for (int r = 1; r < R; r++) {
                double ang = sign * 2.0 * M_PI * (double)((size_t)r * k) / (double)(p * (size_t)R);
                pl->tw[s][k * (size_t)(R - 1) + (size_t)(r - 1)].re = (float)cos(ang);
                pl->tw[s][k * (size_t)(R - 1) + (size_t)(r - 1)].im = (float)sin(ang);
            }
        pl->roots[s] = (orc_c32 *)malloc(sizeof(orc_c32) * (size_t)R);
        for (int j = 0; j < R; j++) {
            double ang = sign * 2.0 * M_PI * (double)j / (double)R;
            pl->roots[s][j].re = (float)cos(ang);
            pl->roots[s][j].im = (float)sin(ang);
        }
        p *= (size_t)R;
    }
    pl->scratch = (orc_c32 *)malloc(sizeof(orc_c32) * n);
    return pl;
}

void orc_fft_plan_destroy(orc_fft_plan *p) {
    if (!p) return;
    for (int s = 0; s < p->npass; s++) { free(p->tw[s]); free(p->roots[s]); }
    free(p->scratch);
    free(p);
}

static void butterfly(int R, orc_c32 *u, const orc_c32 *roots, int inverse) {
    switch (R) {
    case 2: {
        orc_c32 a = u[0], b = u[1];
        u[0] = cadd(a, b); u[1] = csub(a, b);
        return;
    }
    case 3: {
        const float s3 = 0.86602540378443864676f;
        orc_c32 t1 = cadd(u[1], u[2]);
        orc_c32 m = {u[0].re - 0.5f * t1.re, u[0].im - 0.5f * t1.im};
        orc_c32 d = csub(u[1], u[2]);
        orc_c32 sd = {s3 * d.re, s3 * d.im};
        orc_c32 js = cmulj(sd, inverse);
        u[0] = cadd(u[0], t1); u[1] = cadd(m, js); u[2] = csub(m, js);
        return;
    }
    case 4: {
        orc_c32 t0 = cadd(u[0], u[2]), t1 = csub(u[0], u[2]);
        orc_c32 t2 = cadd(u[1], u[3]), t3 = cmulj(csub(u[1], u[3]), inverse);
        u[0] = cadd(t0, t2); u[1] = cadd(t1, t3); u[2] = csub(t0, t2); u[3] = csub(t1, t3);
        return;
    }
    case 5: {
        const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
        const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
        orc_c32 t1 = cadd(u[1], u[4]), t2 = cadd(u[2], u[3]);
        orc_c32 t3 = csub(u[1], u[4]), t4 = csub(u[2], u[3]);
        orc_c32 a1 = {u[0].re + c1 * t1.re + c2 * t2.re, u[0].im + c1 * t1.im + c2 * t2.im};
        orc_c32 a2 = {u[0].re + c2 * t1.re + c1 * t2.re, u[0].im + c2 * t1.im + c1 * t2.im};
        orc_c32 b1 = {s1 * t3.re + s2 * t4.re, s1 * t3.im + s2 * t4.im};
        orc_c32 b2 = {s2 * t3.re - s1 * t4.re, s2 * t3.im - s1 * t4.im};
        orc_c32 jb1 = cmulj(b1, inverse), jb2 = cmulj(b2, inverse);
        orc_c32 y0 = {u[0].re + t1.re + t2.re, u[0].im + t1.im + t2.im};
        u[0] = y0; u[1] = cadd(a1, jb1); u[4] = csub(a1, jb1); u[2] = cadd(a2, jb2); u[3] = csub(a2, jb2);
        return;
    }
    default: {
        orc_c32 y[64];
        orc_c32 *yy = y, *heap = NULL;
        if (R > 64) yy = heap = (orc_c32 *)malloc(sizeof(orc_c32) * (size_t)R);
        for (int q = 0; q < R; q++) {
            orc_c32 acc = u[0];
            for (int r = 1; r < R; r++) acc = cadd(acc, cmul(u[r], roots[(int)(((long)r * q) % R)]));
            yy[q] = acc;
        }
        memcpy(u, yy, sizeof(orc_c32) * (size_t)R);
        free(heap);
        return;
    }
    }
}

void orc_fft_exec(orc_fft_plan *pl, orc_c32 *data) {
    const size_t n = pl->n;
    orc_c32 *x = data, *y = pl->scratch;
    size_t p = 1;
    orc_c32 ubuf[64];
    for (int s = 0; s < pl->npass; s++) {
        const int R = pl->radix[s];
        const size_t t = n / (size_t)R;
        const orc_c32 *tw = pl->tw[s];
        orc_c32 *u = ubuf, *heap = NULL;
        if (R > 64) u = heap = (orc_c32 *)malloc(sizeof(orc_c32) * (size_t)R);
        for (size_t blk = 0; blk < t / p; blk++) {
            for (size_t k = 0; k < p; k++) {
                const size_t i = blk * p + k;
                u[0] = x[i];
                if (k == 0) {
                    for (int r = 1; r < R; r++) u[r] = x[i + (size_t)r * t];
                } else {
                    const orc_c32 *twk = tw + k * (size_t)(R - 1);
                    for (int r = 1; r < R; r++) u[r] = cmul(x[i + (size_t)r * t], twk[r - 1]);
                }
                butterfly(R, u, pl->roots[s], pl->inverse);
                const size_t j = blk * p * (size_t)R + k;
                for (int r = 0; r < R; r++) y[j + (size_t)r * p] = u[r];
            }
        }
        free(heap);
        orc_c32 *tmp = x; x = y; y = tmp;
        p *= (size_t)R;
    }
    if (x != data) memcpy(data, x, sizeof(orc_c32) * n);
}

int orc_fft_forward(orc_c32 *data, size_t n) { /* src/fft.rs:21-25 */
    orc_fft_plan *p = orc_fft_plan_create(n, 0);
    if (!p) return -1;
    orc_fft_exec(p, data);
    orc_fft_plan_destroy(p);
    return 0;
}

int orc_fft_power_spectrum(orc_c32 *data, size_t n, float *power) { /* src/fft.rs:27-29 */
    if (orc_fft_forward(data, n)) return -1;
    for (size_t i = 0; i < n; i++) power[i] = data[i].re * data[i].re + data[i].im * data[i].im;
    return 0;
}

int orc_rfft_forward(const float *in, size_t n, orc_c32 *out) { /* src/fft.rs:45-49 (realfft 3.3.0) */
    orc_c32 *buf = (orc_c32 *)malloc(sizeof(orc_c32) * n);
    if (!buf) return -1;
    for (size_t i = 0; i < n; i++) { buf[i].re = in[i]; buf[i].im = 0.0f; }
    int rc = orc_fft_forward(buf, n);
    if (!rc) memcpy(out, buf, sizeof(orc_c32) * (n / 2 + 1));
    free(buf);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Acquisition (src/acquisition/do_acquisition.rs)
 * ------------------------------------------------------------------------------------------ */
struct orc_acq_worker {
    uint8_t prn;
    size_t fft_size;
    float fs;
    orc_fft_plan *fft, *ifft;
    orc_c32 *code_fft;    /* ca_code_samples_fft :126 */
    orc_c32 *result_buf;  /* :127 */
    float *best_power, *acc_power;
};

static orc_acq_worker *worker_alloc(uint8_t prn, size_t fft_size, float fs) {
    orc_acq_worker *w = (orc_acq_worker *)calloc(1, sizeof(*w));
    w->prn = prn; w->fft_size = fft_size; w->fs = fs;
    w->fft = orc_fft_plan_create(fft_size, 0);
    w->ifft = orc_fft_plan_create(fft_size, 1);
    w->code_fft = (orc_c32 *)calloc(fft_size, sizeof(orc_c32));
    w->result_buf = (orc_c32 *)calloc(fft_size, sizeof(orc_c32)); /* vec![0; fft_size] :154 */
    w->best_power = (float *)calloc(fft_size, sizeof(float));
    w->acc_power = (float *)calloc(fft_size, sizeof(float));
    return w;
}

orc_acq_worker *orc_acq_worker_new(uint8_t prn, size_t fft_size, float fs) { /* :130-156 */
    int8_t *code = (int8_t *)malloc(fft_size + 16);
    long n = orc_generate_ca_code_samples(prn, ORC_GPS_L1_CA_CODE_RATE_CHIPS_PER_S, fs, code, fft_size);
    /* rustfft's process() panics unless buffer length is a multiple of fft_size (:135-137) */
    if (n < 0 || (size_t)n != fft_size) { free(code); return NULL; }
    orc_acq_worker *w = worker_alloc(prn, fft_size, fs);
    for (size_t i = 0; i < fft_size; i++) { w->code_fft[i].re = (float)code[i]; w->code_fft[i].im = 0.0f; }
    orc_fft_exec(w->fft, w->code_fft);
    free(code);
    return w;
}

orc_acq_worker *orc_acq_worker_new_custom(uint8_t prn, size_t fft_size, float fs, const int8_t *code,
                                          size_t code_len, float code_rate) {
    /* generalisation of ca_code.rs:12-27 to an arbitrary +-1 chip sequence (no reference code) */
    orc_acq_worker *w = worker_alloc(prn, fft_size, fs);
    for (size_t i = 0; i < fft_size; i++) {
        float f = floorf(((float)i * code_rate) / fs);
        size_t ind = (f > 0.0f) ? (size_t)f : 0;
        w->code_fft[i].re = (float)code[ind % code_len];
        w->code_fft[i].im = 0.0f;
    }
    orc_fft_exec(w->fft, w->code_fft);
    return w;
}

void orc_acq_worker_free(orc_acq_worker *w) {
    if (!w) return;
    orc_fft_plan_destroy(w->fft); orc_fft_plan_destroy(w->ifft);
    free(w->code_fft); free(w->result_buf); free(w->best_power); free(w->acc_power);
    free(w);
}

const orc_c32 *orc_acq_worker_code_fft(const orc_acq_worker *w) { return w->code_fft; }

int orc_is_good_satellite(const float *power, size_t n, float max_val, float *sum_out) { /* :229-238 */
    float lane[8] = {0, 0, 0, 0, 0, 0, 0, 0};       /* fold(f32x8::splat(0.0), |acc,x| acc + x) */
    size_t chunks = n / 8;                           /* chunks_exact(8) drops the tail */
    for (size_t c = 0; c < chunks; c++)
        for (int l = 0; l < 8; l++) lane[l] = lane[l] + power[c * 8 + (size_t)l];
    float sum = -0.0f;                               /* reduce_sum = simd_reduce_add_ordered(v, -0.0) */
    for (int l = 0; l < 8; l++) sum = sum + lane[l];
    if (sum_out) *sum_out = sum;
    float avg = (sum - max_val) / (float)(n - 1);    /* (self.fft_size - 1) as f32 */
    return (max_val / avg) > 7.0f;
}

int orc_search_satellite(orc_acq_worker *w, const orc_c32 *samples, size_t n_samples,
                         const orc_c32 *const *tables, const float *table_freq, size_t n_tables,
                         uint64_t local_tail, size_t num_integrations, orc_acq_result *out,
                         float *bin_max, uint32_t *bin_argmax, float *bin_sum, uint32_t *bins_done,
                         int no_early_exit) {
    const size_t N = w->fft_size;
    if (n_samples < N * num_integrations) return -1; /* slice index panic :176 */
    float global_max_val = 0.0f, best_doppler_freq = 0.0f;   /* :165-167 */
    size_t best_code_phase = 0;
    int32_t best_bin = -1;
    memset(w->best_power, 0, sizeof(float) * N);             /* :168 */
    int found = 0;
    uint32_t done = 0;
    for (size_t d = 0; d < n_tables; d++) {                  /* :171 */
        memset(w->acc_power, 0, sizeof(float) * N);          /* :172 */
        for (size_t c = 0; c < num_integrations; c++) {      /* :174 */
            const orc_c32 *chunk = samples + c * N;
            orc_apply_doppler_shift(chunk, tables[d], w->result_buf, N);  /* :177-181 */
            orc_fft_exec(w->fft, w->result_buf);                          /* :182 */
            for (size_t i = 0; i < N; i++) {                              /* :184-186  *= conj() */
                orc_c32 a = w->result_buf[i];
                orc_c32 b = {w->code_fft[i].re, -w->code_fft[i].im};
                w->result_buf[i] = cmul(a, b);
            }
            orc_fft_exec(w->ifft, w->result_buf);                         /* :188 */
            for (size_t i = 0; i < N; i++) {                              /* :190-192 norm_sqr */
                orc_c32 v = w->result_buf[i];
                w->acc_power[i] += v.re * v.re + v.im * v.im;
            }
        }
        float local_max = 0.0f;                              /* :195-202 first strict maximum */
        size_t local_best_phase = 0;
        for (size_t i = 0; i < N; i++)
            if (w->acc_power[i] > local_max) { local_max = w->acc_power[i]; local_best_phase = i; }
        if (bin_max) bin_max[d] = local_max;
        if (bin_argmax) bin_argmax[d] = (uint32_t)local_best_phase;
        if (bin_sum) orc_is_good_satellite(w->acc_power, N, local_max, &bin_sum[d]);
        done++;
        if (!found) {
            if (local_max > global_max_val) {                /* :204-209 */
                global_max_val = local_max;
                best_doppler_freq = table_freq[d];
                best_code_phase = local_best_phase;
                best_bin = (int32_t)d;
                memcpy(w->best_power, w->acc_power, sizeof(float) * N);
            }
            if (orc_is_good_satellite(w->best_power, N, global_max_val, NULL)) {  /* :211-222 */
                out->prn = w->prn;
                out->code_phase_samples = best_code_phase;
                out->code_phase_chips =
                    (float)best_code_phase * ORC_GPS_L1_CA_CODE_RATE_CHIPS_PER_S / w->fs;
                out->carrier_freq = best_doppler_freq;
                out->fs = w->fs;
                out->mag_relative = global_max_val;
                out->sample_global_index = local_tail + best_code_phase;
                out->doppler_bin = best_bin;
                found = 1;
                if (!no_early_exit) break;
            }
        }
    }
    if (bins_done) *bins_done = done;
    return found;                                            /* :225 None */
}

int orc_decide_from_metrics(const float *bin_max, const uint32_t *bin_argmax, const float *bin_sum,
                            const float *table_freq, size_t n_tables, size_t fft_size, uint8_t prn,
                            float fs, uint64_t local_tail, float threshold, orc_acq_result *out) {
    float global_max_val = 0.0f, best_freq = 0.0f, best_sum = 0.0f; /* best plane starts all-zero */
    size_t best_phase = 0;
    int32_t best_bin = -1;
    for (size_t d = 0; d < n_tables; d++) {
        if (bin_max[d] > global_max_val) {
            global_max_val = bin_max[d]; best_freq = table_freq[d];
            best_phase = bin_argmax[d]; best_sum = bin_sum[d]; best_bin = (int32_t)d;
        }
        float avg = (best_sum - global_max_val) / (float)(fft_size - 1);
        if (global_max_val / avg > threshold) {
            out->prn = prn; out->code_phase_samples = best_phase;
            out->code_phase_chips = (float)best_phase * ORC_GPS_L1_CA_CODE_RATE_CHIPS_PER_S / fs;
            out->carrier_freq = best_freq; out->fs = fs; out->mag_relative = global_max_val;
            out->sample_global_index = local_tail + best_phase; out->doppler_bin = best_bin;
            return 1;
        }
    }
    return 0;
}

int orc_acq_search_all(orc_acq_worker *const *workers, size_t n_workers, uint64_t mask,
                       const orc_c32 *samples, size_t n_samples, const orc_c32 *const *tables,
                       const float *table_freq, size_t n_tables, uint64_t local_tail,
                       size_t num_integrations, int n_threads, int no_early_exit,
                       orc_acq_result *results, uint8_t *found, uint64_t *cells_out) {
    uint64_t cells = 0;
    int err = 0;
    if (n_threads < 1) n_threads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads) reduction(+ : cells) reduction(| : err)
#endif
    for (long i = 0; i < (long)n_workers; i++) {     /* workers.par_iter_mut().enumerate() :302-313 */
        found[i] = 0;
        if (!((mask >> i) & 1)) continue;            /* (mask >> (prn - 1)) & 1 == 1 */
        uint32_t done = 0;
        int rc = orc_search_satellite(workers[i], samples, n_samples, tables, table_freq, n_tables,
                                      local_tail, num_integrations, &results[i], NULL, NULL, NULL,
                                      &done, no_early_exit);
        if (rc < 0) { err |= 1; continue; }
        found[i] = (uint8_t)rc;
        cells += (uint64_t)done * workers[i]->fft_size;
    }
    if (cells_out) *cells_out = cells;
    return err ? -1 : 0;
}

int orc_acq_mode_for(size_t n) { return n == 0 ? 0 : (n <= 4 ? 1 : 2); }  /* update_mode :50-56 */

void orc_acq_pacing_and_list(int mode, uint32_t active_mask, uint64_t *interval_ms, uint32_t *mask) {
    uint64_t interval; uint32_t search_size;                   /* get_pacing_and_list :58-73 */
    switch (mode) {
    case 0: interval = 500; search_size = ORC_PRN_SEARCH_ACQUISITION_TOTAL; break;
    case 1: interval = 1000; search_size = 8; break;
    default: interval = 2000; search_size = 5; break;
    }
    uint32_t m = 0, taken = 0;
    for (uint32_t prn = 1; prn <= ORC_PRN_SEARCH_ACQUISITION_TOTAL && taken < search_size; prn++)
        if (!((active_mask >> (prn - 1)) & 1)) { m |= (1u << (prn - 1)); taken++; }
    *interval_ms = interval; *mask = m;
}

/* ------------------------------------------------------------------------------------------
 * Tracking (src/tracking/do_tracking.rs)
 * ------------------------------------------------------------------------------------------ */
#define LOCK_THRESHOLD 15.0f          /* :16 */
#define MAX_LOST_EPOCHS 20u           /* :17 */
#define DLL_DUMPING_RATIO 0.7f        /* :19 */
#define PLL_DUMPING_RATIO 0.7f
#define PLL_GAIN 0.25f
#define DLL_NOISE_BANDWIDTH 2.0f
#define PLL_NOISE_BANDWIDTH 25.0f
#define DLL_GAIN 1.0f
#define PLL_SUM_CARR 0.001f
#define DLL_SUM_CODE 0.001f
#define EARLY_LATE_SPACE 0.5f         /* :28 */

orc_loop_filter orc_loop_filter_new(float noise_bw, float zeta, float gain) { /* :59-65 */
    float w = noise_bw * 8.0f * zeta / (4.0f * (zeta * zeta) + 1.0f); /* powf(2.0) == x*x */
    orc_loop_filter f;
    f.tau1 = gain / (w * w);
    f.tau2 = (2.0f * zeta) / w;
    return f;
}

float orc_loop_filter_update(const orc_loop_filter *f, float d_err, float err, float dt) { /* :68-70 */
    return d_err * (dt / f->tau1) + (d_err - err) * (f->tau2 / f->tau1);
}

void orc_trk_new(orc_trk_channel *c, uint8_t id, float fs) { /* :118-146 */
    memset(c, 0, sizeof(*c));
    c->id = id;
    c->state = ORC_STATE_IDLE;
    c->fs = fs;
    c->num_samples_per_code =
        orc_num_samples_per_code(ORC_GPS_L1_CA_CODE_RATE_CHIPS_PER_S, fs);
    c->code_rate = ORC_GPS_L1_CA_CODE_RATE_CHIPS_PER_S;
    c->pll_filter = orc_loop_filter_new(PLL_NOISE_BANDWIDTH, PLL_DUMPING_RATIO, PLL_GAIN);
    c->dll_filter = orc_loop_filter_new(DLL_NOISE_BANDWIDTH, DLL_DUMPING_RATIO, DLL_GAIN);
    c->code_index_mode = ORC_CODE_INDEX_FAITHFUL;
}

void orc_trk_start(orc_trk_channel *c, const orc_acq_result *r) { /* :148-154 */
    c->prn = r->prn;
    c->carrier_freq = r->carrier_freq;
    c->code_phase = r->code_phase_chips;
    c->next_sample_index = r->sample_global_index;
    /* FIXED (our documented deviation): sample_global_index already points at the code start,
     * so the replica starts at chip 0; FAITHFUL copies the acquisition delay like the reference */
    if (c->code_index_mode == ORC_CODE_INDEX_FIXED) c->code_phase = 0.0f;
    c->state = ORC_STATE_TRACKING;
    c->state_prn = r->prn;
}

int orc_trk_is_active(const orc_trk_channel *c) { /* :156-158 */
    return c->state == ORC_STATE_TRACKING && c->state_prn == c->prn;
}

void orc_trk_reset(orc_trk_channel *c) { /* :311-327 */
    c->prn = 0;
    c->state = ORC_STATE_IDLE;
    c->lost_counter = 0;
    c->next_sample_index = 0;
    c->carrier_freq = 0.0f; c->carrier_phase = 0.0f; c->carrier_error = 0.0f; c->carrier_nco = 0.0f;
    c->code_phase = 0.0f; c->code_error = 0.0f; c->code_nco = 0.0f;
    c->code_rate = 0.0f;   /* reference bug kept in the oracle: a restarted channel has code_rate 0 */
    c->i_prompt = 0.0f; c->q_prompt = 0.0f;
}

/* `x.floor() as usize` (saturating cast) then `% 1023` */
static inline size_t floor_as_usize_mod1023(float phase) {
    float f = floorf(phase);
    if (!(f > 0.0f)) return 0;                       /* negative, -0, NaN -> 0 */
    if (f >= 18446744073709551616.0f) return (size_t)(18446744073709551615ull % 1023ull);
    return (size_t)((uint64_t)f % 1023ull);
}

static const int8_t *ca_row_cached(int row) {
    static int8_t table[32][1023];
    static int ready = 0;
    if (!ready) {
#ifdef _OPENMP
#pragma omp critical(orc_ca_table)
#endif
        {
            if (!ready) { for (int r = 0; r < 32; r++) orc_ca_code_row(r, table[r]); ready = 1; }
        }
    }
    return table[row];
}

static int get_chip_general(const orc_trk_channel *c, float phase, float *chip) {
    /* custom code table (no reference): row prn-1, wrapping index, optional BOC(1,1) sign */
    const long len = (long)c->code_len;
    int row = (int)c->prn - 1;
    if (row < 0 || (uint32_t)row >= c->n_codes) return -1;
    float f = floorf(phase);
    long li;
    if (c->code_index_mode == ORC_CODE_INDEX_FAITHFUL) li = (f > 0.0f) ? ((long)f % len) : 0;
    else { li = (long)f % len; if (li < 0) li += len; }
    *chip = (float)c->custom_codes[(size_t)row * (size_t)len + (size_t)li];
    return 0;
}

int orc_trk_get_ca_chip(const orc_trk_channel *c, float phase, float *chip) { /* :274-277 */
    int row;
    size_t idx;
    if (c->custom_codes) return get_chip_general(c, phase, chip);
    if (c->code_index_mode == ORC_CODE_INDEX_FAITHFUL) {
        row = (int)c->prn;                           /* GPS_CA_CODE_32_PRN[self.prn as usize] */
        idx = floor_as_usize_mod1023(phase);
    } else {
        row = (int)c->prn - 1;                       /* fixed: the PRN's own code, wrapping late arm */
        float f = floorf(phase);
        long li = (long)f % 1023;
        if (li < 0) li += 1023;
        idx = (size_t)li;
    }
    if (row < 0 || row > 31) return -1;
    *chip = (float)ca_row_cached(row)[idx];
    return 0;
}

int orc_trk_early_late_correlation(orc_trk_channel *c, orc_c32 *data, float out6[6], double acc64[6]) {
    const size_t n = (size_t)c->num_samples_per_code;
    for (size_t i = 0; i < n; i++) {                 /* :232-238 */
        float phase = c->carrier_phase + (2.0f * PI_F * c->carrier_freq * (float)i / c->fs);
        float cos_p = cosf(phase);
        float sin_p = -sinf(phase);
        orc_c32 w = {cos_p, sin_p};
        data[i] = cmul(data[i], w);
    }
    c->carrier_phase = fmodf(c->carrier_phase +
                                 2.0f * PI_F * c->carrier_freq * ((float)n / c->fs),
                             2.0f * PI_F);           /* :240-242 */
    float i_p = 0.0f, q_p = 0.0f, i_e = 0.0f, q_e = 0.0f, i_l = 0.0f, q_l = 0.0f;
    double a64[6] = {0, 0, 0, 0, 0, 0};
    int rc = 0;
    for (size_t i = 0; i < n; i++) {                 /* :251-263 */
        float chip_idx = fmodf(c->code_phase + ((float)i * (c->code_rate / c->fs)), 1023.0f);
        float p_chip, e_chip, l_chip;
        if (orc_trk_get_ca_chip(c, chip_idx, &p_chip) ||
            orc_trk_get_ca_chip(c, chip_idx + EARLY_LATE_SPACE, &e_chip) ||
            orc_trk_get_ca_chip(c, chip_idx - EARLY_LATE_SPACE, &l_chip)) { rc = -1; break; }
        i_p += data[i].re * p_chip; q_p += data[i].im * p_chip;
        i_e += data[i].re * e_chip; q_e += data[i].im * e_chip;
        i_l += data[i].re * l_chip; q_l += data[i].im * l_chip;
        if (acc64) {
            a64[0] += (double)(data[i].re * p_chip); a64[1] += (double)(data[i].im * p_chip);
            a64[2] += (double)(data[i].re * e_chip); a64[3] += (double)(data[i].im * e_chip);
            a64[4] += (double)(data[i].re * l_chip); a64[5] += (double)(data[i].im * l_chip);
        }
    }
    if (rc) return rc;
    c->code_phase = fmodf(c->code_phase + (c->code_rate / c->fs) * (float)n, 1023.0f); /* :265-267 */
    c->i_prompt = i_p; c->q_prompt = q_p;            /* :269-270 */
    out6[0] = i_p; out6[1] = q_p; out6[2] = i_e; out6[3] = q_e; out6[4] = i_l; out6[5] = q_l;
    if (acc64) memcpy(acc64, a64, sizeof(a64));
    return 0;
}

int orc_trk_early_late_correlation_ex(orc_trk_channel *c, orc_c32 *data, float out10[10], double acc64[10]) {
    /* early_late_correlation :231-272 generalised: same per-sample f32 operations, code length / arm count /
     * BOC(1,1) from the channel.  With n_arms 3, no BOC, no custom code it equals the function above. */
    const size_t n = (size_t)c->num_samples_per_code;
    const float lenf = c->custom_codes ? (float)c->code_len : 1023.0f;
    const float el = c->el_space > 0.0f ? c->el_space : EARLY_LATE_SPACE;
    const float vel = c->vel_space > 0.0f ? c->vel_space : 1.0f;
    const int arms = c->n_arms == 5 ? 5 : 3;
    for (size_t i = 0; i < n; i++) {
        float phase = c->carrier_phase + (2.0f * PI_F * c->carrier_freq * (float)i / c->fs);
        orc_c32 w = {cosf(phase), -sinf(phase)};
        data[i] = cmul(data[i], w);
    }
    c->carrier_phase = fmodf(c->carrier_phase + 2.0f * PI_F * c->carrier_freq * ((float)n / c->fs), 2.0f * PI_F);
    float acc[10] = {0};
    double a64[10] = {0};
    for (size_t i = 0; i < n; i++) {
        float chip_idx = fmodf(c->code_phase + ((float)i * (c->code_rate / c->fs)), lenf);
        float ph[5] = {chip_idx, chip_idx + el, chip_idx - el, chip_idx + vel, chip_idx - vel};
        for (int a = 0; a < arms; a++) {
            float chip;
            if (orc_trk_get_ca_chip(c, ph[a], &chip)) return -1;
            if (c->boc11 && !((ph[a] - floorf(ph[a])) < 0.5f)) chip = -chip;
            acc[2 * a] += data[i].re * chip;
            acc[2 * a + 1] += data[i].im * chip;
            a64[2 * a] += (double)(data[i].re * chip);
            a64[2 * a + 1] += (double)(data[i].im * chip);
        }
    }
    c->code_phase = fmodf(c->code_phase + (c->code_rate / c->fs) * (float)n, lenf);
    c->i_prompt = acc[0]; c->q_prompt = acc[1];
    memcpy(out10, acc, sizeof(acc));
    if (acc64) memcpy(acc64, a64, sizeof(a64));
    return 0;
}

void orc_trk_run_loop_filters(orc_trk_channel *c, float i_p, float q_p, float i_e, float q_e,
                              float i_l, float q_l) { /* :279-302 */
    float pll_err = atanf(q_p / i_p) / (2.0f * PI_F);
    c->carrier_nco = orc_loop_filter_update(&c->pll_filter, pll_err, c->carrier_error, PLL_SUM_CARR);
    c->carrier_error = pll_err;
    c->carrier_freq += c->carrier_nco;
    float pow_e = sqrtf(i_e * i_e + q_e * q_e);      /* powi(2) == x*x */
    float pow_l = sqrtf(i_l * i_l + q_l * q_l);
    float dll_err = ((pow_e + pow_l) != 0.0f) ? (pow_e - pow_l) / (pow_e + pow_l) : 0.0f;
    c->code_nco = orc_loop_filter_update(&c->dll_filter, dll_err, c->code_error, DLL_SUM_CODE);
    c->code_error = dll_err;
    c->code_rate += c->code_nco;
}

int orc_trk_do_work(orc_trk_channel *c, orc_c32 *data, float out6[6], uint8_t *msg_prn) { /* :183-210 */
    if (orc_trk_early_late_correlation(c, data, out6, NULL)) return -1;
    float i_p = out6[0], q_p = out6[1];
    float power = i_p * i_p + q_p * q_p;
    if (power > LOCK_THRESHOLD) {
        c->lost_counter = 0;
        orc_trk_run_loop_filters(c, out6[0], out6[1], out6[2], out6[3], out6[4], out6[5]);
        c->next_sample_index += c->num_samples_per_code;
        c->num_samples_per_code = orc_num_samples_per_code(c->code_rate, c->fs);
        return 0;
    }
    c->lost_counter += 1;
    if (c->lost_counter >= MAX_LOST_EPOCHS) {
        orc_trk_reset(c);
        if (msg_prn) *msg_prn = c->prn;              /* built AFTER reset -> carries 0 (:199-201) */
        return 1;
    }
    c->next_sample_index += c->num_samples_per_code;
    c->num_samples_per_code = orc_num_samples_per_code(c->code_rate, c->fs);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * MulticastRingBuffer (src/utilities/multicast_ring_buffer.rs)
 * ------------------------------------------------------------------------------------------ */
int orc_ring_new(orc_ring *r, size_t buf_size) { /* :46-61 */
    if (buf_size == 0 || (buf_size & (buf_size - 1))) return -1; /* assert power of two */
    r->buffer = (orc_c32 *)calloc(buf_size, sizeof(orc_c32));
    r->buf_size = buf_size; r->mask = buf_size - 1; r->head = 0;
    return r->buffer ? 0 : -1;
}
void orc_ring_free(orc_ring *r) { free(r->buffer); r->buffer = NULL; }

void orc_ring_write_samples(orc_ring *r, const orc_c32 *s, size_t n) { /* :66-101 */
    size_t start = (size_t)(r->head & r->mask);
    if (start + n <= r->buf_size) {
        memcpy(r->buffer + start, s, n * sizeof(orc_c32));
    } else {
        size_t first = r->buf_size - start;
        memcpy(r->buffer + start, s, first * sizeof(orc_c32));
        memcpy(r->buffer, s + first, (n - first) * sizeof(orc_c32));
    }
    r->head += n;
}
uint64_t orc_ring_get_head(const orc_ring *r) { return r->head; }

void orc_ring_copy_to_slice(const orc_ring *r, uint64_t start, orc_c32 *dest, size_t n) { /* :107-129 */
    size_t ps = (size_t)(start & r->mask);
    if (ps + n <= r->buf_size) {
        memcpy(dest, r->buffer + ps, n * sizeof(orc_c32));
    } else {
        size_t first = r->buf_size - ps;
        memcpy(dest, r->buffer + ps, first * sizeof(orc_c32));
        memcpy(dest + first, r->buffer, (n - first) * sizeof(orc_c32));
    }
}

int orc_trk_update(orc_trk_channel *c, const orc_ring *ring, orc_c32 *scratch, float out6[6],
                   uint8_t *msg_prn) { /* :160-180 */
    if (!orc_trk_is_active(c)) return 0;
    /* generate_ca_code_samples(...).len() :165-166 — only the length is used */
    if (c->prn < 1 || c->prn > 32) return -1;
    c->num_samples_per_code = orc_num_samples_per_code(c->code_rate, c->fs);
    uint64_t head = orc_ring_get_head(ring);
    if ((int64_t)(head - (c->next_sample_index + c->num_samples_per_code)) < 0) return 0;
    orc_ring_copy_to_slice(ring, c->next_sample_index, scratch, (size_t)c->num_samples_per_code);
    int rc = orc_trk_do_work(c, scratch, out6, msg_prn);
    if (rc < 0) return -1;
    return rc == 1 ? 2 : 1;
}

/* update() :160-180 + do_work() :183-210 GENERALISED to the channel's code length / arm count / BOC flag (no reference
 * code for those, SURVEY §8c5: the same statements with 1023 -> code_len and the E/P/L correlator -> _ex).  With the
 * reference's GPS settings it performs exactly orc_trk_update's operations.  out10 as in _ex. */
static size_t num_samples_per_code_len(float code_rate, float fs, float lenf) {
    float v = roundf(fs / (code_rate / lenf));       /* ca_code.rs:13-16 with the code length as a parameter */
    if (!(v > 0.0f)) return 0;
    return (size_t)v;
}
int orc_trk_update_ex(orc_trk_channel *c, const orc_ring *ring, orc_c32 *scratch, float out10[10],
                      uint8_t *msg_prn) {
    if (!orc_trk_is_active(c)) return 0;
    const float lenf = c->custom_codes ? (float)c->code_len : 1023.0f;
    if (!c->custom_codes && (c->prn < 1 || c->prn > 32)) return -1;
    c->num_samples_per_code = num_samples_per_code_len(c->code_rate, c->fs, lenf);      /* :165-166 */
    uint64_t head = orc_ring_get_head(ring);
    if ((int64_t)(head - (c->next_sample_index + c->num_samples_per_code)) < 0) return 0; /* :170-172 */
    orc_ring_copy_to_slice(ring, c->next_sample_index, scratch, (size_t)c->num_samples_per_code);
    if (orc_trk_early_late_correlation_ex(c, scratch, out10, NULL)) return -1;            /* do_work :183-210 */
    float power = out10[0] * out10[0] + out10[1] * out10[1];
    if (power > LOCK_THRESHOLD) {
        c->lost_counter = 0;
        orc_trk_run_loop_filters(c, out10[0], out10[1], out10[2], out10[3], out10[4], out10[5]);
        c->next_sample_index += c->num_samples_per_code;
        c->num_samples_per_code = num_samples_per_code_len(c->code_rate, c->fs, lenf);
        return 1;
    }
    c->lost_counter += 1;
    if (c->lost_counter >= MAX_LOST_EPOCHS) {
        orc_trk_reset(c);
        if (msg_prn) *msg_prn = c->prn;
        return 2;
    }
    c->next_sample_index += c->num_samples_per_code;
    c->num_samples_per_code = num_samples_per_code_len(c->code_rate, c->fs, lenf);
    return 1;
}

/* Teacher-forced form of orc_trk_update_ex for the parity tests of multi-epoch device launches: the correlation is
 * computed from the channel's own state (-> computed10, what the device's correlator sums are compared with), but the
 * lock decision, the loop filters and i/q_prompt then use forced10 (the device's sums of that epoch).  By induction the
 * channel state entering every epoch is bit-for-bit what a device that does the reference's scalar arithmetic exactly
 * must hold, so (a) every epoch's sums are compared from IDENTICAL inputs, and (b) the final state must match exactly.
 * computed64 (may be NULL): the same per-sample f32 products accumulated in double. */
int orc_trk_update_forced(orc_trk_channel *c, const orc_ring *ring, orc_c32 *scratch, float computed10[10],
                          double computed64[10], const float forced10[10], uint8_t *msg_prn) {
    if (!orc_trk_is_active(c)) return 0;
    const float lenf = c->custom_codes ? (float)c->code_len : 1023.0f;
    if (!c->custom_codes && (c->prn < 1 || c->prn > 32)) return -1;
    c->num_samples_per_code = num_samples_per_code_len(c->code_rate, c->fs, lenf);
    uint64_t head = orc_ring_get_head(ring);
    if ((int64_t)(head - (c->next_sample_index + c->num_samples_per_code)) < 0) return 0;
    orc_ring_copy_to_slice(ring, c->next_sample_index, scratch, (size_t)c->num_samples_per_code);
    if (orc_trk_early_late_correlation_ex(c, scratch, computed10, computed64)) return -1;
    const float *f = forced10;
    c->i_prompt = f[0]; c->q_prompt = f[1];
    float power = f[0] * f[0] + f[1] * f[1];
    if (power > LOCK_THRESHOLD) {
        c->lost_counter = 0;
        orc_trk_run_loop_filters(c, f[0], f[1], f[2], f[3], f[4], f[5]);
        c->next_sample_index += c->num_samples_per_code;
        c->num_samples_per_code = num_samples_per_code_len(c->code_rate, c->fs, lenf);
        return 1;
    }
    c->lost_counter += 1;
    if (c->lost_counter >= MAX_LOST_EPOCHS) {
        orc_trk_reset(c);
        if (msg_prn) *msg_prn = c->prn;
        return 2;
    }
    c->next_sample_index += c->num_samples_per_code;
    c->num_samples_per_code = num_samples_per_code_len(c->code_rate, c->fs, lenf);
    return 1;
}

/* TrackingManager::process_channels :351-371 — par_iter_mut over the channels, one task per channel, repeated
 * while any channel still finds a whole code period in the ring (the run() loop of :384-415 without the Condvar).
 * scratch: n_channels blocks of scratch_stride samples.  Returns channel-epochs processed, -1 on OOB. */
int64_t orc_trk_process_channels(orc_trk_channel *ch, int n_channels, const orc_ring *ring, orc_c32 *scratch,
                                 size_t scratch_stride, int max_passes, int n_threads) {
    int64_t done = 0;
    int err = 0;
    for (int pass = 0; pass < max_passes; ++pass) {
        int64_t got = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads) reduction(+ : got) reduction(| : err)
        for (int i = 0; i < n_channels; ++i) {
            float out6[6];
            uint8_t prn = 255;
            int rc = orc_trk_update(&ch[i], ring, scratch + (size_t)i * scratch_stride, out6, &prn);
            if (rc < 0) err |= 1;
            else if (rc > 0) got += 1;
        }
        if (err) return -1;
        if (!got) break;
        done += got;
    }
    return done;
}

/* ------------------------------------------------------------------------------------------
 * Digital front-end (src/rf/frontend.rs, src/rf/nco_lut.rs, src/rf/dc_remove.rs)
 * ------------------------------------------------------------------------------------------ */
void orc_frontend_new(orc_frontend *fe, float f_if, float fs_in, float fs_out) { /* frontend.rs:19-30 */
    (void)fs_out;                                                  /* stored but unused by the reference (:12-15) */
    const float pi = 3.14159265358979323846f;                      /* std::f32::consts::PI */
    for (int i = 0; i < ORC_LUT_SIZE; ++i) {                       /* NcoLut::new nco_lut.rs:28-32 */
        const float angle = ((2.0f * pi) * (float)i) / (float)ORC_LUT_SIZE;
        fe->lut_re[i] = cosf(angle);
        fe->lut_im[i] = -sinf(angle);                              /* "Negative for downconversion" */
    }
    fe->phase_step = (f_if / fs_in) * (float)ORC_LUT_SIZE;         /* :34 */
    fe->phase_accumulator = 0.0f;
    fe->alpha = 0.001f;                                            /* DcRemoverSimd::new(0.001) frontend.rs:21 */
    fe->con = 1.0f - fe->alpha;                                    /* dc_remove.rs:12 */
    for (int j = 0; j < 8; ++j) fe->bias_re[j] = fe->bias_im[j] = 0.0f;
}

static size_t f32_as_usize(float v) {          /* Rust `as usize`: saturating, NaN -> 0 */
    if (!(v > 0.0f)) return 0;
    if (v >= 18446744073709551616.0f) return (size_t)-1;
    return (size_t)v;
}

void orc_frontend_process_block(orc_frontend *fe, float *raw, size_t n_floats) { /* frontend.rs:33-62 */
    for (size_t c = 0; c + 16 <= n_floats; c += 16) {              /* chunks_exact_mut(16) :35 */
        float *chunk = raw + c;
        float re[8], im[8];
        for (int j = 0; j < 8; ++j) { re[j] = chunk[2 * j]; im[j] = chunk[2 * j + 1]; }   /* deinterleave :39 */
        for (int j = 0; j < 8; ++j) {                              /* DcRemoverSimd::process_block dc_remove.rs:22-28 */
            fe->bias_re[j] = fe->bias_re[j] * fe->con + re[j] * fe->alpha;
            fe->bias_im[j] = fe->bias_im[j] * fe->con + im[j] * fe->alpha;
            re[j] = re[j] - fe->bias_re[j];
            im[j] = im[j] - fe->bias_im[j];
        }
        size_t idx[8];
        for (int j = 0; j < 8; ++j) {                              /* :47-52 */
            idx[j] = f32_as_usize(fe->phase_accumulator) % ORC_LUT_SIZE;
            fe->phase_accumulator = fmodf(fe->phase_accumulator + fe->phase_step, (float)ORC_LUT_SIZE);
        }
        for (int j = 0; j < 8; ++j) {                              /* mix_simd nco_lut.rs:8-15 */
            const float lc = fe->lut_re[idx[j]], ls = fe->lut_im[idx[j]];
            const float mi = re[j] * lc + im[j] * ls;
            const float mq = re[j] * ls - im[j] * lc;
            chunk[2 * j] = mi;                                     /* interleave :59-61 */
            chunk[2 * j + 1] = mq;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * finer_doppler (src/acquisition/acquisition_bk.rs:215-302, legacy)
 * ------------------------------------------------------------------------------------------ */
int orc_finer_doppler(const orc_c32 *samples, size_t n_samples, size_t code_phase, const int8_t *chips, size_t code_len,
                      float code_rate, float fs, size_t size_signal_use, uint64_t *peak_index, float *peak_mag,
                      float *freq_hz, int *upper_half, size_t *fft_size_out) {
    if (!size_signal_use || code_phase + size_signal_use > n_samples) return -1;   /* slice :260 would panic */
    float sum_re = 0.0f, sum_im = 0.0f;                                            /* iter().sum::<Complex32>() :236 */
    for (size_t i = 0; i < n_samples; ++i) { sum_re += samples[i].re; sum_im += samples[i].im; }
    const float mean_re = sum_re / (float)n_samples, mean_im = sum_im / (float)n_samples;
    size_t p2 = 1;
    while (p2 < size_signal_use) p2 <<= 1;                                         /* next_power_of_two :249 */
    const size_t fft_size = 8 * p2;
    orc_c32 *x = (orc_c32 *)calloc(fft_size, sizeof(orc_c32));                     /* zero padding :255,273 */
    if (!x) return -2;
    for (size_t n = 0; n < size_signal_use; ++n) {
        const size_t ind = (size_t)floorf(((float)n * code_rate) / fs) % code_len; /* :241-247 */
        const float c = (float)chips[ind];
        x[n].re = (samples[code_phase + n].re - mean_re) * c;                      /* :237, :266-272 */
        x[n].im = (samples[code_phase + n].im - mean_im) * c;
    }
    orc_fft_plan *pl = orc_fft_plan_create(fft_size, 0);
    if (!pl) { free(x); return -2; }
    orc_fft_exec(pl, x);                                                           /* :275 */
    orc_fft_plan_destroy(pl);
    float best = -1.0f;
    size_t idx = 0;
    for (size_t k = 0; k < fft_size; ++k) {                                        /* abs() :276, max :278, first equal :279-282 */
        const float m = hypotf(x[k].re, x[k].im);
        if (m > best) { best = m; idx = k; }
    }
    free(x);
    const size_t one_side = (size_t)ceilf(((float)fft_size + 1.0f) / 2.0f);        /* :250 */
    if (peak_index) *peak_index = idx;
    if (peak_mag) *peak_mag = best;
    if (fft_size_out) *fft_size_out = fft_size;
    if (idx > one_side) {                                                          /* :284-297 — the legacy panics here */
        if (upper_half) *upper_half = 1;
        if (freq_hz) *freq_hz = -(((float)(fft_size - idx) * fs) / (float)fft_size);
    } else {
        if (upper_half) *upper_half = 0;
        if (freq_hz) *freq_hz = ((float)idx * fs) / (float)fft_size;               /* :251-253 */
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Bit sync / nav-bit accumulation / preamble (src/decoding.rs, legacy)
 * ------------------------------------------------------------------------------------------ */
static const int8_t ORC_GPS_CA_PREAMBLE[8] = {1, -1, -1, -1, 1, -1, 1, 1};   /* gps_property_constants.rs:13 */
#define ORC_BIT_SYNC_THRESHOLD 30                                              /* decoding.rs:8 */

void orc_nav_sync_new(orc_nav_sync *s, int fixed) { /* :68-100 */
    memset(s, 0, sizeof(*s));
    s->fixed = fixed;
    s->polarity = -1;
}
void orc_nav_sync_free(orc_nav_sync *s) { free(s->frame_bits); s->frame_bits = NULL; }

static int orc_check_bit_sync(orc_nav_sync *s, float old_i, float i_p) { /* :164-182 */
    if (old_i * i_p < 0.0f) {
        s->bit_sync_buff[s->biti] += 1;
        size_t i_max = 0;
        uint64_t v_max = s->bit_sync_buff[0];
        for (size_t i = 1; i < 20; ++i)                 /* Iterator::max_by keeps the LAST of equal maxima */
            if (s->bit_sync_buff[i] >= v_max) { v_max = s->bit_sync_buff[i]; i_max = i; }
        s->frame_sync_ind = i_max;
        if (v_max == ORC_BIT_SYNC_THRESHOLD) return 1;
    }
    return 0;
}

static void orc_bit_accumulation(orc_nav_sync *s, float i_p, uint64_t loop_ms, uint64_t buff_loc) { /* :184-214 */
    (void)buff_loc;
    s->sync_sw = 0;
    if (s->biti == s->frame_sync_ind) { s->bit_code_cnt = 1; s->i_p = i_p; }
    else s->i_p += i_p;
    s->loop_sw = (s->bit_code_cnt % loop_ms) == 0;
    const uint64_t last = s->fixed ? (s->frame_sync_ind + 19) % 20 : s->frame_sync_ind + 19;   /* :203-205 */
    if (s->biti == last) {
        const int8_t bit = s->i_p > 0.0f ? 1 : -1;
        if (s->n_frame_bits == s->cap_frame_bits) {
            s->cap_frame_bits = s->cap_frame_bits ? 2 * s->cap_frame_bits : 512;
            s->frame_bits = (int8_t *)realloc(s->frame_bits, s->cap_frame_bits);
        }
        s->frame_bits[s->n_frame_bits++] = bit;
        s->last_bit = bit;
        s->sync_sw = 1;
        if (!s->flag_frame_sync) {                      /* buff_preamble.push_back(bit) :210-212 */
            if (s->fixed) {
                memmove(s->buff_preamble, s->buff_preamble + 1, 7);
                s->buff_preamble[7] = bit;
                if (s->n_preamble < 8) s->n_preamble++;
            } else {
                if (s->n_preamble < 8) s->buff_preamble[s->n_preamble] = bit;
                s->n_preamble++;
            }
        }
    }
    s->bit_code_cnt += 1;
}

static int orc_check_preamble_syn(orc_nav_sync *s) { /* :216-227 */
    int corr = 0;
    for (int x = 0; x < 8; ++x) corr += s->buff_preamble[x] * ORC_GPS_CA_PREAMBLE[x % 8];
    if (abs(corr) == 8) { s->polarity = (int8_t)(corr > 0 ? 1 : -1); return 1; }
    return 0;
}

int orc_nav_sync_update(orc_nav_sync *s, float old_i, float i_p, uint64_t cnt, uint64_t buff_loc) { /* :102-162 */
    s->biti = cnt % 20;                                                        /* :114 */
    if (!s->flag_bit_sync && cnt > (uint64_t)(1.0f / 1.0e-3f))                 /* :115 */
        s->flag_bit_sync = orc_check_bit_sync(s, old_i, i_p);
    if (s->flag_bit_sync) orc_bit_accumulation(s, i_p, 10 /* tracking::LOOP_MS */, buff_loc);
    if (s->sync_sw) {                                                          /* :129-145 */
        if (!s->flag_frame_sync && s->n_preamble == 8) s->flag_frame_sync = orc_check_preamble_syn(s);
        if (s->flag_frame_sync) {
            s->sf_buffer_loc = buff_loc;
            s->sf_cnt = cnt;
            s->sf_start_biti = s->n_frame_bits - 8;
            s->tow_expected_ind = cnt + 30 * 20;
        }
    }
    return s->sync_sw;
}

int orc_nav_parity_check(const int8_t b[32], int *ref_sum_zero) { /* :259-352 */
    static const int8_t idx[6][16] = {
        {0, 2, 3, 4, 6, 7, 11, 12, 13, 14, 15, 18, 19, 21, 24, -1},
        {1, 3, 4, 5, 7, 8, 12, 13, 14, 15, 16, 19, 20, 22, 25, -1},
        {0, 2, 4, 5, 6, 8, 9, 13, 14, 15, 16, 17, 20, 21, 23, -1},
        {1, 3, 5, 6, 7, 9, 10, 14, 15, 16, 17, 18, 21, 22, 24, -1},
        {1, 2, 4, 6, 7, 8, 10, 11, 15, 16, 17, 18, 19, 22, 23, 25},
        {0, 4, 6, 7, 9, 10, 11, 12, 14, 16, 20, 23, 24, 25, -1, -1}};
    int all = 1, sum = 0;
    for (int k = 0; k < 6; ++k) {
        int p = 1;
        for (int j = 0; j < 16 && idx[k][j] >= 0; ++j) p *= b[idx[k][j]];
        if (p != b[26 + k]) all = 0;
        sum += p - b[26 + k];
    }
    if (ref_sum_zero) *ref_sum_zero = ((int8_t)sum) == 0;
    return all;
}

