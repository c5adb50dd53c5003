// fe_kernels.hip — the digital front-end in front of the acquisition / tracking path:
// DigitalFrontend::process_block (src/rf/frontend.rs:33-62) = DcRemoverSimd (src/rf/dc_remove.rs:22-28, eight
// independent one-pole IIR lanes per I and per Q) + NcoLut phase accumulator and LUT gather (src/rf/nco_lut.rs:17-42,
// frontend.rs:47-55) + mix_simd (nco_lut.rs:8-15), with the int8 -> f32 conversion and the ring write of
// rf_thread.rs:43-48 fused in.
//
// Both recurrences round in f32 at every step, so they are evaluated in the reference's order to stay bit-exact:
//   * the NCO phase chain (data-independent) runs on ONE lane of wave 0,
//   * the 16 DC-bias chains run on 16 lanes of wave 1, concurrently with it,
//   * everything else (load/convert, LUT gather, complex mix, store) is spread over the workgroup.
// One workgroup per stream; a segment of 2048 samples costs ~the NCO chain (2 dependent VALU ops per sample, measured
// 9.1 ns: a lone wave issues a DEPENDENT VALU op every ~4.5 ns) + 3.5 us of load/mix: 22 us, 93 Msps per stream.
// Compiled with -ffp-contract=off: a*b + c*d must round like rustc's (no FMA).
#include "gm_internal.h"

namespace gm {

namespace {
constexpr int FE_T = 256;
constexpr int FE_SEG = 2048;
constexpr int LUT = 2048;

__device__ __forceinline__ uint32_t as_usize_mod_lut(float p) {   // `phase_accumulator as usize % LUT_SIZE` (:49)
    // Rust's float -> usize cast saturates: negative and NaN -> 0; |p| < 2^24 here on the fast path
    if (!(p > 0.0f)) return 0u;
    if (p < 4.0e9f) return uint32_t(p) & (LUT - 1);
    if (p >= 1.8446744e19f) return uint32_t(0xFFFFFFFFFFFFFFFFull % LUT);
    return uint32_t(static_cast<unsigned long long>(p) % LUT);
}

template <int FMT>
__device__ __forceinline__ void load_sample(const void* in, size_t i, float& re, float& im) {
    if (FMT == GM_FMT_C32) {
        const float2 v = reinterpret_cast<const float2*>(in)[i];
        re = v.x; im = v.y;
    } else {   // GM_FMT_I8_IQ
        const char2 v = reinterpret_cast<const char2*>(in)[i];
        re = float(v.x); im = float(v.y);
    }
}

// One segment of the phase chain (frontend.rs:47-52), fast form.  Scaling by 2^-11 commutes with f32 rounding, so the
// chain is kept in revolutions r = |phase| / 2048 in [0, 1):  r' = fract(fl(r + |step|/2048)) is bit-for-bit
// |fmodf(fl(phase + step), 2048)| / 2048 when phase and step have the same sign (fmodf keeps the dividend's sign; RNE is
// symmetric under negation; for r + s in [1, 2) the subtraction of 1 is exact).  Two dependent VALU ops per sample.
__device__ __forceinline__ void nco_segment_fast(float& r, float s_abs, float* s_ph, int L) {
#pragma unroll 8
    for (int k = 0; k < L; ++k) {
        s_ph[k] = r;
        r = __builtin_amdgcn_fractf(r + s_abs);
    }
}

template <int FMT>
__global__ __launch_bounds__(FE_T) void frontend_kernel(FrontendArgs a) {
    __shared__ float s_re[FE_SEG], s_im[FE_SEG];
    __shared__ float s_lre[LUT], s_lim[LUT];
    __shared__ float s_ph[FE_SEG];                         // phase_accumulator before each sample (x s_scale)
    __shared__ float s_scale;
    const int tid = threadIdx.x;
    const FrontendArgs::Stream st = a.streams ? a.streams[blockIdx.x] : a.one;
    for (int i = tid; i < LUT; i += FE_T) { s_lre[i] = a.lut[i]; s_lim[i] = a.lut[LUT + i]; }

    const size_t n8 = st.n_samples & ~size_t(7);          // chunks_exact_mut(16 floats) (:35)
    float phase = 0.0f, bias = 0.0f, nco_scale = 1.0f;
    const bool nco_lane = tid == 0;
    const bool dc_lane = tid >= 64 && tid < 80;           // wave 1, lanes 0..15: lane j -> re lane j, 8+j -> im lane j
    const int dl = tid - 64;
    if (nco_lane) phase = st.state->phase_accumulator;
    if (dc_lane) bias = dl < 8 ? st.state->bias_re[dl] : st.state->bias_im[dl - 8];
    const float step = st.phase_step, alpha = a.alpha, con = a.con;
    const bool fast = st.fast_fmod != 0;

    for (size_t seg = 0; seg < n8; seg += FE_SEG) {
        const int L = int(n8 - seg < size_t(FE_SEG) ? n8 - seg : size_t(FE_SEG));
        for (int i = tid; i < L; i += FE_T) load_sample<FMT>(st.in, seg + i, s_re[i], s_im[i]);
        __syncthreads();
        if (nco_lane) {                                   // frontend.rs:47-52
            // fast form: |phase|, |step| < 2048, same sign (or zero), step not so small that r + s could be subnormal
            const bool neg = step < 0.0f;
            const bool same_sign = neg ? !(phase > 0.0f) : !(phase < 0.0f);
            if (fast && fabsf(phase) < 2048.0f && same_sign && (step == 0.0f || fabsf(step) > 1.0e-20f)) {
                float r = fabsf(phase) * (1.0f / 2048.0f);
                nco_segment_fast(r, fabsf(step) * (1.0f / 2048.0f), s_ph, L);
                nco_scale = neg ? -2048.0f : 2048.0f;
                phase = r * nco_scale;                      // -0.0 for a negative chain that landed on -2048
            } else {
                for (int k = 0; k < L; ++k) {
                    s_ph[k] = phase;
                    phase = fmodf(phase + step, 2048.0f);
                }
                nco_scale = 1.0f;
            }
            s_scale = nco_scale;
        } else if (dc_lane) {                             // dc_remove.rs:22-28, lane j of the f32x8
            float* buf = dl < 8 ? s_re : s_im;
            const int j = dl & 7;
            int k = j;
            for (; k + 56 < L; k += 64) {                 // 8 steps per trip: LDS reads issued together, then the chain
                float x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = buf[k + 8 * u];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    bias = bias * con + x[u] * alpha;
                    x[u] = x[u] - bias;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) buf[k + 8 * u] = x[u];
            }
            for (; k < L; k += 8) {
                const float x = buf[k];
                bias = bias * con + x * alpha;
                buf[k] = x - bias;
            }
        }
        __syncthreads();
        for (int i = tid; i < L; i += FE_T) {             // mix_simd nco_lut.rs:8-15, interleave :59-61
            const float re = s_re[i], im = s_im[i];
            const uint32_t idx = as_usize_mod_lut(s_ph[i] * s_scale);   // `phase_accumulator as usize % LUT_SIZE` (:49)
            const float lc = s_lre[idx], ls = s_lim[idx];
            float2 o;
            o.x = re * lc + im * ls;
            o.y = re * ls - im * lc;
            reinterpret_cast<float2*>(st.out)[(st.out_start + seg + i) & st.out_mask] = o;
        }
        __syncthreads();
    }
    // the tail that chunks_exact leaves untouched: converted, not processed
    for (size_t i = n8 + tid; i < st.n_samples; i += FE_T) {
        float2 o;
        load_sample<FMT>(st.in, i, o.x, o.y);
        reinterpret_cast<float2*>(st.out)[(st.out_start + i) & st.out_mask] = o;
    }
    if (nco_lane) st.state->phase_accumulator = phase;
    if (dc_lane) { if (dl < 8) st.state->bias_re[dl] = bias; else st.state->bias_im[dl - 8] = bias; }
}
}  // namespace

void launch_frontend(hipStream_t s, const FrontendArgs& a, int n_streams, int fmt) {
    if (fmt == GM_FMT_C32) frontend_kernel<GM_FMT_C32><<<n_streams, FE_T, 0, s>>>(a);
    else frontend_kernel<GM_FMT_I8_IQ><<<n_streams, FE_T, 0, s>>>(a);
}

}  // namespace gm
