// acq_composite.hip — acquisition at transform sizes that do not fit one LDS buffer (N > 16384), e.g. a Galileo E1
// code period at 8 Msps (N = 32000, BASELINE configs[3]): N = Q * Nb with Nb one of the in-LDS plans and Q in 2..8.
//
// Same algorithm as acq_kernels.hip (AcquisitionWorker::search_satellite, do_acquisition.rs:158-226: mix -> FFT ->
// x conj(code FFT) -> IFFT -> |.|^2 accumulated over the integrations -> first strict argmax / max / plane sum), with
// the length-N transforms taken apart (n = n1*Nb + n2, k = Q*k2 + k1):
//
//   forward   X[Q k2 + k1] = sum_n2 W_Nb^{n2 k2} * ( W_N^{n2 k1} * sum_n1 x[n1 Nb + n2] W_Q^{n1 k1} )
//             comp_pre_kernel (the Q-point DFTs across the Q blocks + twiddle, the carrier mix fused in) ->
//             the batched in-LDS transform of size Nb (fft_batch_kernel) -> spectrum in "decimated" order [k1][k2]
//   product   elementwise in that order (the code spectra are produced by the same two steps), fused into the loads of
//   inverse   y[n1 Nb + n2] = sum_k1 W_Q^{-n1 k1} * ( W_N^{-n2 k1} * IFFT_Nb(Y[Q k2 + k1])[n2] )
//             the batched inverse transforms (comp_corr_fft_kernel in acq_kernels.hip), then comp_post_kernel: twiddle + Q-point inverse DFTs, |y|^2 accumulated over
//             the integrations in registers, reduced to {max, first argmax, sum} per (worker, bin): no plane is stored.
//
// Intermediates travel through HBM / L2 (about four passes over P*D*M*N*8 bytes): a first, correct version of the
// large-N case; the fused single-LDS-buffer kernels remain the path for N <= 16384.
#include "gm_internal.h"

namespace gm {

namespace {
constexpr int CT = 256;

__device__ __forceinline__ cf comp_load(const void* samples, int fmt, size_t idx) {
    if (fmt == GM_FMT_C32) return reinterpret_cast<const cf*>(samples)[idx];
    if (fmt == GM_FMT_I8_IQ) {
        const char2 v = reinterpret_cast<const char2*>(samples)[idx];
        return cf_make(float(v.x), float(v.y));
    }
    return cf_make(float(reinterpret_cast<const int8_t*>(samples)[idx]), 0.0f);
}

// e^{-+ 2 pi i t / n} for integer t in [0, n): the argument is formed from the exact integer phase
__device__ __forceinline__ cf unit_root(uint32_t t, uint32_t n, bool inverse) {
    float sn, cs;
    sincospif(2.0f * (float(t) / float(n)), &sn, &cs);
    return cf_make(cs, inverse ? sn : -sn);
}
__device__ __forceinline__ cf cmulf(cf a, cf b) {
    return cf_make(__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x));
}

// forward pre-pass.  grid (ceil(Nb / CT), n_items); item = (d, m) for the signal (tables != null: carrier mix
// apply_doppler_shift fused, doppler_shift.rs:43-58, same products) or a code index for the replicas (int8 chips).
// out[item][k1][n2]
template <uint32_t Q>
__global__ __launch_bounds__(CT) void comp_pre_kernel(const void* __restrict__ in, int fmt, const cf* __restrict__ tables,
                                                      cf* __restrict__ out, uint32_t Nb, uint32_t n_int,
                                                      const int8_t* __restrict__ code_samples) {
    const uint32_t n2 = blockIdx.x * CT + threadIdx.x;
    if (n2 >= Nb) return;
    const uint32_t item = blockIdx.y, N = Q * Nb;
    cf x[Q], wq[Q];                                 // compile-time Q: both arrays stay in registers
#pragma unroll
    for (uint32_t j = 0; j < Q; ++j) wq[j] = unit_root(j, Q, false);
#pragma unroll
    for (uint32_t n1 = 0; n1 < Q; ++n1) {
        const uint32_t n = n1 * Nb + n2;
        if (code_samples) {
            x[n1] = cf_make(float(code_samples[size_t(item) * N + n]), 0.0f);        // :134 (i8 -> f32, im = 0)
        } else {
            const uint32_t d = item / n_int, m = item % n_int;
            const cf s = comp_load(in, fmt, size_t(m) * N + n);
            const cf t = tables[size_t(d) * N + n];
            x[n1] = cf_make(s.x * t.x - s.y * t.y, s.x * t.y + s.y * t.x);           // multiply_simd_block
        }
    }
#pragma unroll
    for (uint32_t k1 = 0; k1 < Q; ++k1) {
        cf acc = x[0];
#pragma unroll
        for (uint32_t n1 = 1; n1 < Q; ++n1) acc = cf_add(acc, cmulf(x[n1], wq[(n1 * k1) % Q]));
        const cf w = unit_root(uint32_t((uint64_t(n2) * k1) % N), N, false);
        out[(size_t(item) * Q + k1) * Nb + n2] = cmulf(acc, w);
    }
}

// inverse post-pass + power accumulation + reduction.  One workgroup per (w, d); z[w][d][m][k1][n2].
template <uint32_t Q>
__global__ __launch_bounds__(CT) void comp_post_kernel(const cf* __restrict__ z, uint32_t Nb, uint32_t n_int,
                                                       uint32_t n_bins, const uint32_t* __restrict__ worker_list,
                                                       float* __restrict__ mmax, uint32_t* __restrict__ margmax,
                                                       float* __restrict__ msum) {
    __shared__ float s_p[CT / 64], s_s[CT / 64];
    __shared__ uint32_t s_k[CT / 64];
    const uint32_t d = blockIdx.x, w = blockIdx.y, N = Q * Nb, tid = threadIdx.x;
    const cf* base = z + (size_t(w) * n_bins + d) * n_int * N;
    float best = 0.0f, sum = 0.0f;                 // running max starts from 0.0 like the reference (:195-202)
    uint32_t bestn = 0;
    bool any = false;
    cf wq[Q];
#pragma unroll
    for (uint32_t j = 0; j < Q; ++j) wq[j] = unit_root(j, Q, true);
    for (uint32_t n2 = tid; n2 < Nb; n2 += CT) {
        float acc[Q];
        cf wn[Q];                                  // W_N^{-n2 k1}, the same for every integration
#pragma unroll
        for (uint32_t k1 = 0; k1 < Q; ++k1) { acc[k1] = 0.0f; wn[k1] = unit_root(uint32_t((uint64_t(n2) * k1) % N), N, true); }
        for (uint32_t m = 0; m < n_int; ++m) {
            cf u[Q];
#pragma unroll
            for (uint32_t k1 = 0; k1 < Q; ++k1) u[k1] = cmulf(base[(size_t(m) * Q + k1) * Nb + n2], wn[k1]);
#pragma unroll
            for (uint32_t n1 = 0; n1 < Q; ++n1) {
                cf y = u[0];
#pragma unroll
                for (uint32_t k1 = 1; k1 < Q; ++k1) y = cf_add(y, cmulf(u[k1], wq[(n1 * k1) % Q]));
                acc[n1] += y.x * y.x + y.y * y.y;  // norm_sqr accumulated per integration (:190-192)
            }
        }
#pragma unroll
        for (uint32_t n1 = 0; n1 < Q; ++n1) {
            const uint32_t n = n1 * Nb + n2;
            sum += acc[n1];
            if (acc[n1] > best || (acc[n1] == best && any && n < bestn && acc[n1] > 0.0f)) { best = acc[n1]; bestn = n; any = true; }
        }
    }
    // first strict maximum over n: larger value wins, equal values -> smaller index
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off), os = __shfl_xor(sum, off);
        const uint32_t on = __shfl_xor(bestn, off);
        sum += os;
        if (ob > best || (ob == best && on < bestn)) { best = ob; bestn = on; }
    }
    if ((tid & 63) == 0) { s_p[tid >> 6] = best; s_k[tid >> 6] = bestn; s_s[tid >> 6] = sum; }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < CT / 64; ++i) {
            sum += s_s[i];
            if (s_p[i] > best || (s_p[i] == best && s_k[i] < bestn)) { best = s_p[i]; bestn = s_k[i]; }
        }
        const size_t o = size_t(worker_list[w]) * n_bins + d;
        mmax[o] = best; margmax[o] = best > 0.0f ? bestn : 0u; msum[o] = sum;
    }
}
}  // namespace

#define GM_COMP_Q_SWITCH(Q, CALL)                       \
    switch (Q) {                                        \
        case 2: { constexpr uint32_t QQ = 2; CALL; } break; \
        case 3: { constexpr uint32_t QQ = 3; CALL; } break; \
        case 4: { constexpr uint32_t QQ = 4; CALL; } break; \
        case 5: { constexpr uint32_t QQ = 5; CALL; } break; \
        case 6: { constexpr uint32_t QQ = 6; CALL; } break; \
        default: { constexpr uint32_t QQ = 8; CALL; } break; \
    }

bool comp_q_supported(uint32_t Q) { return Q == 2 || Q == 3 || Q == 4 || Q == 5 || Q == 6 || Q == 8; }

void launch_comp_pre(hipStream_t st, const void* in, int fmt, const cf* tables, cf* out, uint32_t Q, uint32_t Nb,
                     uint32_t n_int, uint32_t n_items, const int8_t* code_samples) {
    GM_COMP_Q_SWITCH(Q, hipLaunchKernelGGL(comp_pre_kernel<QQ>, dim3((Nb + CT - 1) / CT, n_items), dim3(CT), 0, st, in, fmt,
                                           tables, out, Nb, n_int, code_samples))
}
void launch_comp_post(hipStream_t st, const cf* z, uint32_t Q, uint32_t Nb, uint32_t n_int, uint32_t n_bins,
                      const uint32_t* worker_list, uint32_t n_workers, float* mmax, uint32_t* margmax, float* msum) {
    GM_COMP_Q_SWITCH(Q, hipLaunchKernelGGL(comp_post_kernel<QQ>, dim3(n_bins, n_workers), dim3(CT), 0, st, z, Nb, n_int,
                                           n_bins, worker_list, mmax, margmax, msum))
}

}  // namespace gm
