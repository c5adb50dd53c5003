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
// ------------------------------------------------------------------------------------ table-driven fast form
// The NCO phase chain is data-independent: from phase 0 it is ONE fixed orbit of r -> fract(fl(r + s)), and that orbit is
// periodic after a few steps with a period of at most 2^23 (measured: 4 ... 8 388 607 steps for front-end settings from
// 2 to 50 Msps).  gm_frontend builds it once on the host (exactly the f32 operations of the sequential form) and the
// kernel looks phases up instead of running the chain.  What stays sequential is the DC remover: 16 one-pole recurrences
// (eight SIMD lanes x I/Q, dc_remove.rs:22-28) that round at every step and depend on the data.  They run on 16 lanes of
// ONE wave — two dependent VALU operations per 8 samples — and that wave does nothing else:
//   stage A (waves 1..):  load + convert segment t, xa = x * alpha               -> LDS xa[t & 1][lane][step]
//   stage B (wave 0):     bias = bias * con + xa, step after step; the bias after every 4th step -> LDS ckpt[(t-1) & 1]
//   stage C (waves 1..):  segment t-2: each lane takes 4 consecutive steps of one SIMD lane j (re and im), replays them from
//                         the checkpoint with the same two operations, x - bias, table phase -> LUT index -> gather,
//                         mix_simd, store
// One workgroup barrier per segment; the stages of three different segments overlap.  Writing only checkpoints matters:
// a 16-byte LDS store costs the lone chain wave ~26 cycles (its address and data cross to the LDS at 2 cycles per dword
// on one half of the path), as much as four chain steps; per 16 steps it now issues 4 reads, 32 VALU and 1 store.
// LDS rows are padded by 4 floats: the 16 chain lanes read 16 bytes each at a stride of one row (conflict-free over 64 banks).
constexpr int FF_T = 1024;
constexpr int FF_SEG = FE_FAST_SEG, FF_STEPS = FF_SEG / 8, FF_ROW = FF_STEPS + 4;
constexpr int FF_QUADS = FF_STEPS / 4, FF_QROW = FF_QUADS + 4;

template <int FMT>
__global__ __launch_bounds__(FF_T) void frontend_fast_kernel(FrontendArgs a) {
    __shared__ __attribute__((aligned(16))) float s_xa[2][16][FF_ROW];
    __shared__ __attribute__((aligned(16))) float s_ckpt[2][16][FF_QROW];     // bias after quad q of the segment
    __shared__ float s_start[2][16];                                            // bias entering the segment
    __shared__ float s_lre[LUT], s_lim[LUT];
    const int tid = threadIdx.x;
    const FrontendArgs::Stream st = a.streams ? a.streams[blockIdx.x] : a.one;
    for (int i = tid; i < LUT; i += FF_T) { s_lre[i] = a.lut[i]; s_lim[i] = a.lut[LUT + i]; }
    const size_t n8 = st.n_samples & ~size_t(7);          // chunks_exact_mut(16 floats) (frontend.rs:35)
    const int nseg = int((n8 + FF_SEG - 1) / FF_SEG);
    const float alpha = a.alpha, con = a.con;
    const bool chain_wave = tid < 64;
    const int hl = tid - 64, HT = FF_T - 64;               // helper lane index / count
    float bias = 0.0f;
    if (tid < 16) bias = tid < 8 ? st.state->bias_re[tid] : st.state->bias_im[tid - 8];

    // stage A operands of the next iteration, requested one iteration ahead (their latency overlaps work and barrier)
    constexpr int UA = (FF_SEG + (FF_T - 64) - 1) / (FF_T - 64);
    float nar[UA], nai[UA];
    auto stage_a_loads = [&](int tt) {
        const bool dA = tt < nseg;
        const size_t sA = size_t(dA ? tt : 0) * FF_SEG;
        const int lA = dA ? int(n8 - sA < size_t(FF_SEG) ? n8 - sA : size_t(FF_SEG)) : 0;
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const int i = hl + u * HT;
            nar[u] = nai[u] = 0.0f;
            if (i < lA) load_sample<FMT>(st.in, sA + i, nar[u], nai[u]);
        }
    };
    // stage C operands likewise: item = (quad q, SIMD lane j) of segment tt - 2; one item per helper lane (FE_FAST_SEG)
    static_assert(FF_QUADS * 8 <= FF_T - 64, "one stage-C item per helper lane");
    float nxr[4], nxi[4], nph[4];
    auto stage_c_loads = [&](int tt) {
        const int scn = tt - 2;
        const bool dC = scn >= 0 && scn < nseg;
        const size_t sC = size_t(dC ? scn : 0) * FF_SEG;
        const int stepsC = dC ? int((n8 - sC < size_t(FF_SEG) ? n8 - sC : size_t(FF_SEG)) / 8) : 0;
        // table index of the segment's first sample: tab_pos + sC, folded back into [0, tab_len)
        uint64_t p0 = uint64_t(st.tab_pos) + sC;
        if (p0 >= st.tab_len) p0 = (st.tab_len - st.tab_lambda) + (p0 - (st.tab_len - st.tab_lambda)) % st.tab_lambda;
        const int j = hl & 7, q = hl >> 3;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * q + k, i = 8 * c + j;
            nxr[k] = nxi[k] = nph[k] = 0.0f;
            if (c < stepsC) {
                load_sample<FMT>(st.in, sC + i, nxr[k], nxi[k]);
                uint32_t p = uint32_t(p0) + uint32_t(i);              // i < FF_SEG <= tab_lambda: one fold at most
                if (p >= st.tab_len) p -= st.tab_lambda;
                nph[k] = st.ph_table[p];
            }
        }
    };
    if (!chain_wave) { stage_a_loads(0); stage_c_loads(0); }
    __syncthreads();

    for (int t = 0; t < nseg + 2; ++t) {
        if (chain_wave) {
            const int sb = t - 1;                          // stage B: the 16 chains over segment sb
            if (sb >= 0 && sb < nseg && tid < 16) {
                __builtin_amdgcn_s_setprio(3);             // the chain is the critical path: never queue behind a helper wave
                const size_t seg = size_t(sb) * FF_SEG;
                const int steps = int((n8 - seg < size_t(FF_SEG) ? n8 - seg : size_t(FF_SEG)) / 8);
                const float4* xa4 = reinterpret_cast<const float4*>(&s_xa[sb & 1][tid][0]);
                float4* ck4 = reinterpret_cast<float4*>(&s_ckpt[sb & 1][tid][0]);
                s_start[sb & 1][tid] = bias;
                // con in a VGPR: a VALU instruction with an SGPR source runs at half rate on gfx950 (tools/ubench/valu_forms)
                float conv = con;
                asm volatile("" : "+v"(conv));
                // a lone wave issues one instruction per ~4.5-6 cycles whatever it is: fewest instructions per step
                auto quad = [&](const float4 v) {
                    bias = bias * conv + v.x;              // bias = bias * con + input * alpha  (dc_remove.rs:24-25)
                    bias = bias * conv + v.y;
                    bias = bias * conv + v.z;
                    bias = bias * conv + v.w;
                    return bias;
                };
                const int q4 = steps / 4, q16 = q4 / 4;
                for (int g = 0; g < q16; ++g) {
                    const float4 v0 = xa4[4 * g], v1 = xa4[4 * g + 1], v2 = xa4[4 * g + 2], v3 = xa4[4 * g + 3];
                    float4 o;
                    o.x = quad(v0); o.y = quad(v1); o.z = quad(v2); o.w = quad(v3);
                    ck4[g] = o;
                }
                for (int q = q16 * 4; q < q4; ++q) s_ckpt[sb & 1][tid][q] = quad(xa4[q]);
                for (int k = q4 * 4; k < steps; ++k) bias = bias * conv + s_xa[sb & 1][tid][k];
                __builtin_amdgcn_s_setprio(0);
            }
        } else {
            float ar[UA], ai[UA];
#pragma unroll
            for (int u = 0; u < UA; ++u) { ar[u] = nar[u]; ai[u] = nai[u]; }
            const int sc = t - 2;
            const bool doA = t < nseg, doC = sc >= 0 && sc < nseg;
            const size_t segA = size_t(doA ? t : 0) * FF_SEG, segC = size_t(doC ? sc : 0) * FF_SEG;
            const int LA = doA ? int(n8 - segA < size_t(FF_SEG) ? n8 - segA : size_t(FF_SEG)) : 0;
            const int LC = doC ? int(n8 - segC < size_t(FF_SEG) ? n8 - segC : size_t(FF_SEG)) : 0;
            stage_a_loads(t + 1);
#pragma unroll
            for (int u = 0; u < UA; ++u) {                 // stage A: segment t
                const int i = hl + u * HT;
                if (i < LA) {
                    s_xa[t & 1][i & 7][i >> 3] = ar[u] * alpha;
                    s_xa[t & 1][8 + (i & 7)][i >> 3] = ai[u] * alpha;
                }
            }
            float xr[4], xi[4], ph[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { xr[k] = nxr[k]; xi[k] = nxi[k]; ph[k] = nph[k]; }
            stage_c_loads(t + 1);
            if (doC) {                                     // stage C: segment t - 2, this lane's item (quad q, SIMD lane j)
                const int stepsC = LC / 8;
                const int j = hl & 7, q = hl >> 3;
                if (4 * q < stepsC) {
                    float bre = q ? s_ckpt[sc & 1][j][q - 1] : s_start[sc & 1][j];
                    float bim = q ? s_ckpt[sc & 1][8 + j][q - 1] : s_start[sc & 1][8 + j];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int c = 4 * q + k, i = 8 * c + j;
                        if (c < stepsC) {
                            bre = bre * con + xr[k] * alpha;               // the chain's own two operations, replayed
                            bim = bim * con + xi[k] * alpha;
                            const float re = xr[k] - bre, im = xi[k] - bim;               // input - bias (dc_remove.rs:27)
                            const uint32_t idx = as_usize_mod_lut(ph[k] * st.tab_scale);   // `phase_accumulator as usize % LUT_SIZE` (:49)
                            const float lc = s_lre[idx], ls = s_lim[idx];
                            float2 o;
                            o.x = re * lc + im * ls;                       // mix_simd nco_lut.rs:8-15
                            o.y = re * ls - im * lc;
                            reinterpret_cast<float2*>(st.out)[(st.out_start + segC + i) & st.out_mask] = o;
                        }
                    }
                }
            }
        }
        // workgroup barrier that orders LDS traffic only: the next iteration's global loads (and this one's output stores)
        // stay in flight across it (__syncthreads() would wait for vmcnt(0))
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    // the tail that chunks_exact leaves untouched: converted, not processed
    for (size_t i = n8 + tid; i < st.n_samples; i += FF_T) {
        float2 o;
        load_sample<FMT>(st.in, i, o.x, o.y);
        reinterpret_cast<float2*>(st.out)[(st.out_start + i) & st.out_mask] = o;
    }
    if (tid == 0) st.state->phase_accumulator = st.ph_table[st.tab_pos_end] * st.tab_scale;   // -0.0 for a negative chain on 0
    if (tid < 16) { if (tid < 8) st.state->bias_re[tid] = bias; else st.state->bias_im[tid - 8] = bias; }
}

// ------------------------------------------------------------------------------------ speculative form: one block on K workgroups
// What stays sequential in frontend_fast_kernel is the DC remover's sixteen f32 recurrences bias' = fl(fl(bias * con) + x * alpha):
// 0.55 ms per block of 2^19 samples on ONE wave, the bound of the whole receiver chain.  They forget their start: two chains
// fed the same samples contract by con = 0.999 per step and, once within an ulp or two, land on the same float and stay together —
// measured on the CPU: identical from 1 ulp apart after ~300 steps (worst of 12: 2 400), from 25 ulps after ~1 900 (worst of 42: 5 300).
// So a block is cut into K runs of pipeline segments, one workgroup each.  Workgroup s
//   1. GUESSES the sixteen biases WARM segments in front of its run: the recurrence in exact arithmetic (f64) over the <= 16 384 steps
//      before that point, from the block's true initial state — within ~25 ulps of the f32 chain (its own roundings, a random walk);
//   2. runs the ordinary pipeline (stage A / chain / stage C) from there, with outputs suppressed through the warm-up: 8 640 steps
//      in which the guessed chains fall onto the true ones;
//   3. records the state it ENTERED its run with and the state it left it with.
// A second, one-workgroup kernel then walks the runs in order: run s stands if the state it entered with is bit for bit the state
// run s - 1 (already known to be right) left — all sixteen words; else that run alone is done again from the right state (its
// outputs rewritten), which is the sequential kernel on 1 / K of the block.  EXACT by construction: a guess that did not converge
// costs time, never a wrong sample.  Workgroups whose warm-up would start before the block simply start AT the block's first
// sample with the true state (no guess).  The input must not alias the output (the warm-up re-reads samples of earlier runs).
constexpr int FS_WARM_SEGS = 18;                 // 8 640 chain steps of warm-up
constexpr int FS_GUESS_STEPS = 16384;            // con^16384 = 7.6e-8: what is left of the prior's error is far below an ulp
constexpr int FS_PARTS = 128;
static_assert(FS_PARTS * 8 == FF_T, "fs_guess: lane = (part of the window, SIMD lane j)");
static_assert(FE_SPEC_K_MAX * 16 <= FF_T, "the walk verifies every run at once: lane = (run, chain)");

struct __attribute__((aligned(16))) FeSharedMem {
    float xa[2][16][FF_ROW];
    float ckpt[2][16][FF_QROW];
    float start[2][16];
    float lre[LUT], lim[LUT];
    double parts[FS_PARTS][16];
    int flag;
};

__device__ __forceinline__ double fs_pow(double c, uint32_t m) {          // c^m by squaring
    double r = 1.0;
    while (m) { if (m & 1u) r *= c; c *= c; m >>= 1; }
    return r;
}

// lanes tid < 16: the bias of chain tid at sample p (a multiple of 8) by the exact-arithmetic recurrence from `b_init` (the block's
// initial state); other lanes: unspecified.  All FF_T lanes take part.
template <int FMT>
__device__ __forceinline__ float fs_guess(const FrontendArgs& a, const FrontendArgs::Stream& st, FeSharedMem& S, size_t p, float b_init) {
    const int tid = threadIdx.x;
    const uint32_t steps_before = uint32_t(p / 8);
    const uint32_t nw = steps_before < uint32_t(FS_GUESS_STEPS) ? steps_before : uint32_t(FS_GUESS_STEPS);
    const uint32_t k0 = steps_before - nw, pl = (nw + FS_PARTS - 1) / FS_PARTS;
    const int j = tid & 7, part = tid >> 3;                   // SIMD lane j (its re AND im chain), part of the window
    const double c = double(a.con);
    const uint32_t ka = k0 + uint32_t(part) * pl, kb = (ka + pl < k0 + nw) ? ka + pl : k0 + nw;
    double are = 0.0, aim = 0.0;
    uint32_t k = ka;
    for (; k + 16 <= kb; k += 16) {                              // sixteen loads requested before the first is used: the chain below
        float re[16], im[16];                                    // is dependent, the loads are not (one at a time each paid a round trip)
#pragma unroll
        for (int u = 0; u < 16; ++u) load_sample<FMT>(st.in, size_t(k + u) * 8 + j, re[u], im[u]);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            are = are * c + double(re[u] * a.alpha);             // x * alpha rounded to f32, as stage A hands it to the chain
            aim = aim * c + double(im[u] * a.alpha);
        }
    }
    for (; k < kb; ++k) {
        float re, im;
        load_sample<FMT>(st.in, size_t(k) * 8 + j, re, im);
        are = are * c + double(re * a.alpha);
        aim = aim * c + double(im * a.alpha);
    }
    const double w = ka < kb ? fs_pow(c, (k0 + nw) - kb) : 0.0;
    S.parts[part][j] = are * w;
    S.parts[part][8 + j] = aim * w;
    __syncthreads();
    float g = 0.0f;
    if (tid < 16) {
        double t = double(b_init) * fs_pow(c, nw);
        for (int q = 0; q < FS_PARTS; ++q) t += S.parts[q][tid];
        g = float(t);
    }
    __syncthreads();
    return g;
}

// The pipeline of frontend_fast_kernel over the pipeline segments [seg_begin, seg_end) of the block, outputs for the segments
// >= seg_out only.  bias (lanes tid < 16): in = the state entering seg_begin, out = the state leaving seg_end - 1; bias_at_out =
// the state entering seg_out.  Every lane of the workgroup calls it (it holds workgroup barriers).
template <int FMT>
__device__ __forceinline__ void fs_pipeline(const FrontendArgs& a, const FrontendArgs::Stream& st, FeSharedMem& S, int seg_begin, int seg_out,
                                            int seg_end, size_t n8, float& bias, float& bias_at_out) {
    const int tid = threadIdx.x;
    const float alpha = a.alpha, con = a.con;
    const bool chain_wave = tid < 64;
    const int hl = tid - 64, HT = FF_T - 64;
    constexpr int UA = (FF_SEG + (FF_T - 64) - 1) / (FF_T - 64);
    float nar[UA], nai[UA];
    auto seg_len = [&](int sg) { const size_t o = size_t(sg) * FF_SEG; return int(n8 - o < size_t(FF_SEG) ? n8 - o : size_t(FF_SEG)); };
    auto stage_a_loads = [&](int tt) {
        const bool dA = tt < seg_end;
        const size_t sA = size_t(dA ? tt : 0) * FF_SEG;
        const int lA = dA ? seg_len(tt) : 0;
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const int i = hl + u * HT;
            nar[u] = nai[u] = 0.0f;
            if (i < lA) load_sample<FMT>(st.in, sA + i, nar[u], nai[u]);
        }
    };
    float nxr[4], nxi[4], nph[4];
    auto stage_c_loads = [&](int tt) {
        const int scn = tt - 2;
        const bool dC = scn >= seg_out && scn < seg_end;
        const size_t sC = size_t(dC ? scn : 0) * FF_SEG;
        const int stepsC = dC ? seg_len(scn) / 8 : 0;
        uint64_t p0 = uint64_t(st.tab_pos) + sC;
        if (p0 >= st.tab_len) p0 = (st.tab_len - st.tab_lambda) + (p0 - (st.tab_len - st.tab_lambda)) % st.tab_lambda;
        const int j = hl & 7, q = hl >> 3;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * q + k, i = 8 * c + j;
            nxr[k] = nxi[k] = nph[k] = 0.0f;
            if (c < stepsC) {
                load_sample<FMT>(st.in, sC + i, nxr[k], nxi[k]);
                uint32_t p = uint32_t(p0) + uint32_t(i);
                if (p >= st.tab_len) p -= st.tab_lambda;
                nph[k] = st.ph_table[p];
            }
        }
    };
    if (!chain_wave) { stage_a_loads(seg_begin); stage_c_loads(seg_begin); }
    __syncthreads();
    for (int t = seg_begin; t < seg_end + 2; ++t) {
        if (chain_wave) {
            const int sb = t - 1;
            if (sb >= seg_begin && sb < seg_end && tid < 16) {
                __builtin_amdgcn_s_setprio(3);
                const int steps = seg_len(sb) / 8;
                const float4* xa4 = reinterpret_cast<const float4*>(&S.xa[sb & 1][tid][0]);
                float4* ck4 = reinterpret_cast<float4*>(&S.ckpt[sb & 1][tid][0]);
                S.start[sb & 1][tid] = bias;
                if (sb == seg_out) bias_at_out = bias;
                float conv = con;
                asm volatile("" : "+v"(conv));
                auto quad = [&](const float4 v) {
                    bias = bias * conv + v.x;
                    bias = bias * conv + v.y;
                    bias = bias * conv + v.z;
                    bias = bias * conv + v.w;
                    return bias;
                };
                const int q4 = steps / 4, q16 = q4 / 4;
                if (q16 > 0) {      // the NEXT group's sixteen inputs are requested before this group's dependent chain starts (113 -> 108 us per block; the
                                    // same form in frontend_fast_kernel made that kernel 15 % slower and is not used there)
                    float4 n0 = xa4[0], n1 = xa4[1], n2 = xa4[2], n3 = xa4[3];
                    for (int g = 0; g + 1 < q16; ++g) {
                        const float4 v0 = n0, v1 = n1, v2 = n2, v3 = n3;
                        n0 = xa4[4 * g + 4]; n1 = xa4[4 * g + 5]; n2 = xa4[4 * g + 6]; n3 = xa4[4 * g + 7];
                        float4 o;
                        o.x = quad(v0); o.y = quad(v1); o.z = quad(v2); o.w = quad(v3);
                        ck4[g] = o;
                    }
                    float4 o;
                    o.x = quad(n0); o.y = quad(n1); o.z = quad(n2); o.w = quad(n3);
                    ck4[q16 - 1] = o;
                }
                for (int q = q16 * 4; q < q4; ++q) S.ckpt[sb & 1][tid][q] = quad(xa4[q]);
                for (int k = q4 * 4; k < steps; ++k) bias = bias * conv + S.xa[sb & 1][tid][k];
                __builtin_amdgcn_s_setprio(0);
            }
        } else {
            float ar[UA], ai[UA];
#pragma unroll
            for (int u = 0; u < UA; ++u) { ar[u] = nar[u]; ai[u] = nai[u]; }
            const int sc = t - 2;
            const bool doA = t < seg_end, doC = sc >= seg_out && sc < seg_end;
            const size_t segC = size_t(doC ? sc : 0) * FF_SEG;
            const int LA = doA ? seg_len(t) : 0;
            const int LC = doC ? seg_len(sc) : 0;
            stage_a_loads(t + 1);
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const int i = hl + u * HT;
                if (i < LA) {
                    S.xa[t & 1][i & 7][i >> 3] = ar[u] * alpha;
                    S.xa[t & 1][8 + (i & 7)][i >> 3] = ai[u] * alpha;
                }
            }
            float xr[4], xi[4], ph[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { xr[k] = nxr[k]; xi[k] = nxi[k]; ph[k] = nph[k]; }
            stage_c_loads(t + 1);
            if (doC) {
                const int stepsC = LC / 8;
                const int j = hl & 7, q = hl >> 3;
                if (4 * q < stepsC) {
                    float bre = q ? S.ckpt[sc & 1][j][q - 1] : S.start[sc & 1][j];
                    float bim = q ? S.ckpt[sc & 1][8 + j][q - 1] : S.start[sc & 1][8 + j];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int c = 4 * q + k, i = 8 * c + j;
                        if (c < stepsC) {
                            bre = bre * con + xr[k] * alpha;
                            bim = bim * con + xi[k] * alpha;
                            const float re = xr[k] - bre, im = xi[k] - bim;
                            const uint32_t idx = as_usize_mod_lut(ph[k] * st.tab_scale);
                            const float lc = S.lre[idx], ls = S.lim[idx];
                            float2 o;
                            o.x = re * lc + im * ls;
                            o.y = re * ls - im * lc;
                            reinterpret_cast<float2*>(st.out)[(st.out_start + segC + i) & st.out_mask] = o;
                        }
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}

// run s of the block: its first / one-past-last pipeline segment (empty: first >= last)
__device__ __forceinline__ void fs_run_bounds(int nseg, int K, int s, int& first, int& last) {
    const int per = (nseg + K - 1) / K;
    first = s * per < nseg ? s * per : nseg;
    last = first + per < nseg ? first + per : nseg;
}

template <int FMT>
__global__ __launch_bounds__(FF_T) void frontend_spec_kernel(FrontendArgs a) {
    __shared__ FeSharedMem S;
    const int tid = threadIdx.x, s = blockIdx.x, K = a.spec_k;
    const FrontendArgs::Stream st = a.one;
    for (int i = tid; i < LUT; i += FF_T) { S.lre[i] = a.lut[i]; S.lim[i] = a.lut[LUT + i]; }
    const size_t n8 = st.n_samples & ~size_t(7);
    const int nseg = int((n8 + FF_SEG - 1) / FF_SEG);
    int first, last;
    fs_run_bounds(nseg, K, s, first, last);
    float* rec = a.spec_buf + size_t(s) * 32;
    if (first >= last) return;                                   // (an empty run: the walk skips it by the same arithmetic)
    const int warm = a.spec_warm > 0 ? a.spec_warm : FS_WARM_SEGS;
    const int seg_begin = first > warm ? first - warm : 0;
    float b_init = 0.0f;
    if (tid < 16) b_init = tid < 8 ? st.state->bias_re[tid] : st.state->bias_im[tid - 8];
    float bias = b_init;
    if (seg_begin > 0) {
        bias = fs_guess<FMT>(a, st, S, size_t(seg_begin) * FF_SEG, b_init);
        if (a.spec_poison && tid < 16) bias = bias * 1.25f + 0.5f;     // diagnostic: a guess that cannot converge (the walk must repair every run)
    }
    float at_out = bias;
    fs_pipeline<FMT>(a, st, S, seg_begin, first, last, n8, bias, at_out);
    if (tid < 16) { rec[tid] = at_out; rec[16 + tid] = bias; }
}

// the walk over the runs (one workgroup): verify, repair where a guess did not converge, then the block's final state and its tail
template <int FMT>
__global__ __launch_bounds__(FF_T) void frontend_spec_walk_kernel(FrontendArgs a) {
    __shared__ FeSharedMem S;
    const int tid = threadIdx.x, K = a.spec_k;
    const FrontendArgs::Stream st = a.one;
    const size_t n8 = st.n_samples & ~size_t(7);
    const int nseg = int((n8 + FF_SEG - 1) / FF_SEG);
    bool lut_loaded = false;
    // the common case first, all runs at once: lane (s, j) compares what run s entered with against what run s - 1 left; if every
    // run stands, the block's final state is what the last run left (one barrier instead of a dependent global read per run)
    int last_run = 0;
    for (int s = 1; s < K; ++s) { int f, l; fs_run_bounds(nseg, K, s, f, l); if (f < l) last_run = s; }
    int first_bad = K;
    {
        const int s = tid >> 4, j = tid & 15;
        int bad = 0;
        if (s >= 1 && s <= last_run && tid < FF_T)
            bad = __float_as_uint(a.spec_buf[size_t(s) * 32 + j]) != __float_as_uint(a.spec_buf[size_t(s - 1) * 32 + 16 + j]) ? 1 : 0;
        if (!__syncthreads_or(bad)) first_bad = K;
        else {                                                    // the first run that does not stand (rare): found by a second sweep
            S.flag = K;
            __syncthreads();
            if (bad) atomicMin(&S.flag, s);
            __syncthreads();
            first_bad = S.flag;
        }
    }
    float cur = 0.0f;                                            // lanes < 16: the TRUE state after the runs walked so far
    const int from = first_bad < K ? first_bad : last_run + 1;   // runs before `from` stand as they are
    if (tid < 16) cur = a.spec_buf[size_t(from - 1) * 32 + 16 + tid];
    for (int s = from; s <= last_run; ++s) {
        int first, last;
        fs_run_bounds(nseg, K, s, first, last);
        const float* rec = a.spec_buf + size_t(s) * 32;
        const int bad = (tid < 16 && __float_as_uint(rec[tid]) != __float_as_uint(cur)) ? 1 : 0;
        if (__syncthreads_or(bad)) {                             // the run entered with another state than its predecessor left: again, from the right one
            if (!lut_loaded) {
                for (int i = tid; i < LUT; i += FF_T) { S.lre[i] = a.lut[i]; S.lim[i] = a.lut[LUT + i]; }
                lut_loaded = true;
            }
            float bias = cur, at_out = cur;
            fs_pipeline<FMT>(a, st, S, first, first, last, n8, bias, at_out);
            cur = bias;
            if (tid == 0 && a.spec_repairs) atomicAdd(a.spec_repairs, 1u);   // repairs so far (diagnostic)
        } else if (tid < 16) {
            cur = rec[16 + tid];
        }
    }
    for (size_t i = n8 + tid; i < st.n_samples; i += FF_T) {     // the tail that chunks_exact leaves untouched: converted, not processed
        float2 o;
        load_sample<FMT>(st.in, i, o.x, o.y);
        reinterpret_cast<float2*>(st.out)[(st.out_start + i) & st.out_mask] = o;
    }
    if (tid == 0) st.state->phase_accumulator = st.ph_table[st.tab_pos_end] * st.tab_scale;
    if (tid < 16) { if (tid < 8) st.state->bias_re[tid] = cur; else st.state->bias_im[tid - 8] = cur; }
}

}  // namespace

void launch_frontend_fast(hipStream_t s, const FrontendArgs& a, int n_streams, int fmt) {
    if (fmt == GM_FMT_C32) frontend_fast_kernel<GM_FMT_C32><<<n_streams, FF_T, 0, s>>>(a);
    else frontend_fast_kernel<GM_FMT_I8_IQ><<<n_streams, FF_T, 0, s>>>(a);
}

// one stream, one block on a.spec_k workgroups + the walk (see frontend_spec_kernel)
void launch_frontend_spec(hipStream_t s, const FrontendArgs& a, int fmt) {
    if (fmt == GM_FMT_C32) {
        frontend_spec_kernel<GM_FMT_C32><<<a.spec_k, FF_T, 0, s>>>(a);
        frontend_spec_walk_kernel<GM_FMT_C32><<<1, FF_T, 0, s>>>(a);
    } else {
        frontend_spec_kernel<GM_FMT_I8_IQ><<<a.spec_k, FF_T, 0, s>>>(a);
        frontend_spec_walk_kernel<GM_FMT_I8_IQ><<<1, FF_T, 0, s>>>(a);
    }
}

void launch_frontend(hipStream_t s, const FrontendArgs& a, int n_streams, int fmt) {
    if (fmt == GM_FMT_C32) frontend_kernel<GM_FMT_C32><<<n_streams, FE_T, 0, s>>>(a);
    else frontend_kernel<GM_FMT_I8_IQ><<<n_streams, FE_T, 0, s>>>(a);
}

}  // namespace gm
