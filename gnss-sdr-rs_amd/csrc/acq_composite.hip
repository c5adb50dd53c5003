// acq_composite.hip — acquisition at transform sizes that do not fit one LDS buffer (N > 16384), e.g. a Galileo E1
// code period at 8 Msps (N = 32000, BASELINE configs[3]) or GPS C/A at 25 Msps (N = 25000): N = Q * Nb with Nb one of six
// in-LDS plans (16000, 8000, 8192, 6000, 5000, 4000) and Q in {2, 3, 4, 5, 6, 8}; the largest Nb that divides N is taken
// (every spectrum element is re-read Q times: measured at N = 32000, 2 x 16000 takes 0.48 ms per dwell, 4 x 8000 0.62 ms).
//
// Same algorithm as acq_kernels.hip (AcquisitionWorker::search_satellite, do_acquisition.rs:158-226: mix -> FFT ->
// x conj(code FFT) -> IFFT -> |.|^2 accumulated over the integrations -> first strict argmax / max / plane sum), with the
// length-N transforms decimated in time, n = Q*n2 + n1 and k = k1*Nb + k2:
//
//   forward   X[k1 Nb + k2] = sum_n1 W_Q^{n1 k1} * W_N^{n1 k2} * ( sum_n2 x[Q n2 + n1] W_Nb^{n2 k2} )
//             comp_fwd_sub_kernel: Q in-LDS transforms over the decimated inputs (carrier mix fused into the loads), then
//             comp_fwd_post_kernel: twiddle + Q-point DFTs across n1 -> the spectrum in NATURAL block order, each block of
//             Nb stored in the paired layout acq_corr reads (PairLayout).  The code spectra are made by the same two steps.
//   inverse   y[Q n2 + n1] = sum_k2 W_Nb^{-n2 k2} * ( W_N^{-n1 k2} * sum_k1 Y[k1 Nb + k2] W_Q^{-n1 k1} ),   Y = X conj(C)
//             comp_corr_kernel: ONE workgroup per (worker, bin) runs, for n1 = 0 .. Q-1 and every integration, one in-LDS
//             transform whose pass-0 inputs are formed on the fly (Q products X conj(C), the Q-point DFT row n1, the
//             twiddle) and whose outputs ARE final correlation values y[Q n2 + n1]: |y|^2 is accumulated over the
//             integrations in registers and reduced to {max, first argmax, sum} — nothing goes back to HBM.
//
// Round 1 decimated in frequency instead: its inverse sub-transforms produced intermediate planes z that a second kernel had
// to combine, 721 MB written and read back per dwell at configs[3] geometry (2.2 GB of fabric traffic for 1.5 GB of
// algorithmic bytes).  Here every spectrum element is re-read Q times from L2 (once per n1), which costs pass-0 load
// slots but no HBM traffic: the working set is the spectra (D*M*N*8 B) and the code spectra (P*N*8 B).
#include "acq_device.h"
#include "acq_comp_ws.h"
#include <vector>

namespace gm {

namespace {
constexpr int CT = 256;

// e^{-+ 2 pi i t / n} for integer t in [0, n): the argument is formed from the exact integer phase
__device__ __forceinline__ cf unit_root(uint32_t t, uint32_t n, bool inverse) {
    float sn, cs;
    sincospif(2.0f * (float(t) / float(n)), &sn, &cs);
    return cf_make(cs, inverse ? sn : -sn);
}

// ------------------------------------------------------------------------------------ forward, step 1
// grid n_items * Q: item = (d, m) for the signal (tables != null: apply_doppler_shift fused, doppler_shift.rs:43-58, same
// products) or a code index for the replicas (int8 chips, :134); n1 = blockIdx % Q.  A[item][n1][k2], natural order.
template <class PLX>
__global__ __launch_bounds__(MixPlanOf<PLX>::type::T) void comp_fwd_sub_kernel(const void* __restrict__ samples, int fmt,
                                                             const cf* __restrict__ tables,
                                                             const int8_t* __restrict__ code_samples,
                                                             const cf* __restrict__ tw_fwd, cf* __restrict__ A,
                                                             uint32_t Q, uint32_t n_int, const uint16_t* __restrict__ order) {
    // order != null (base sizes whose correlation plan reads a permuted spectrum): the sub-transform leaves in STORAGE order —
    // A[item][n1][p] holds element order[p] — staged through the LDS buffer like stage F (acq_kernels.hip), so that
    // comp_fwd_post_kernel reads and writes position by position, coalesced.
    // PLX: the base size's registered plan; PL: the plan the forward sub-transform runs on (MixPlanOf: like stage F this is one
    // round of latency-bound workgroups) — `tw_fwd` is PL's table (PlanOps::fill_tw_mix)
    using PL = typename MixPlanOf<PLX>::type;
    constexpr int STAGE = CorrMode<typename CompPlanOf<PLX>::type>::PERMUTED ? PL::N + PL::N / 32 + 1 : 0;
    constexpr int LDS_N = PL::LDS_ELEMS + PL::TW_TOTAL > STAGE ? PL::LDS_ELEMS + PL::TW_TOTAL : STAGE;
    __shared__ cf lds[LDS_N];
    cf* tw = lds + PL::LDS_ELEMS;
    const int tid = threadIdx.x;
    load_twiddles<PL>(tw, tw_fwd, tid);
    const uint32_t item = blockIdx.x / Q, n1 = blockIdx.x % Q;
    const size_t N = size_t(Q) * PL::N;
    const uint32_t d = item / n_int, m = item % n_int;
    cf* dst = A + size_t(blockIdx.x) * PL::N;
    constexpr int NB0 = PL::NB(0), NBL = PL::NB(PL::NP - 1);
    auto in = [&](int it, int r) {
        const size_t n = size_t(Q) * uint32_t((tid + it * PL::T) + r * NB0) + n1;
        if (code_samples) return cf_make(float(code_samples[size_t(item) * N + n]), 0.0f);
        const cf s = load_sample(samples, fmt, size_t(m) * N + n);
        const cf t = tables[size_t(d) * N + n];
        return cf_make(s.x * t.x - s.y * t.y, s.x * t.y + s.y * t.x);           // multiply_simd_block
    };
    if (!STAGE || !order) {
        lds_transform<PL, false>(in, [&](int it, int r, cf val) { dst[(tid + it * PL::T) + r * NBL] = val; }, lds, tw, tid);
    } else {
        {
            cf v0[PL::IT0][PL::R0];
            Fft<PL, false>::pass0_stage1(v0, in, tid);
            __syncthreads();
            Fft<PL, false>::pass0_stage2(v0, lds, tid);
        }
        __syncthreads();
        MiddlePasses<PL, false, 1>::run(lds, tw, tid);
        cf vl[PL::ITL][PL::RL];
        Fft<PL, false>::last_stage1(vl, lds, tw, tid);
        __syncthreads();
        Fft<PL, false>::last_stage2(vl, [&](int it, int r, cf val) {
            const int k = (tid + it * PL::T) + r * NBL;
            lds[k + (k >> 5)] = val; }, tid);
        __syncthreads();
        for (int p = tid; p < PL::N; p += PL::T) {
            const int k = order[p];
            dst[p] = lds[k + (k >> 5)];
        }
    }
}

// ------------------------------------------------------------------------------------ forward, step 2
// grid (ceil(Nb / CT), n_items): X[item][k1][place] = sum_n1 W_Q^{n1 k1} W_N^{n1 k2} A[item][n1][..] for element k2 of the block;
// place: the element's position in the stored block — CorrLayout<CP>::slot(k2) (paired != 0: pairs, and the permutation of a
// prime-factor / hybrid correlation plan) or k2 (paired == 0: natural order, the API's view of the code spectra).
// order != null: A is already in storage order (comp_fwd_sub_kernel), thread = position, k2 = order[position].
template <class PL, uint32_t Q>
__global__ __launch_bounds__(CT) void comp_fwd_post_kernel(const cf* __restrict__ A, cf* __restrict__ X, int paired,
                                                           const uint16_t* __restrict__ order) {
    using CL = CorrLayout<typename CompPlanOf<PL>::type>;
    const uint32_t t = blockIdx.x * CT + threadIdx.x;
    if (t >= uint32_t(PL::N)) return;
    const uint32_t item = blockIdx.y, Nb = PL::N, N = Q * Nb;
    const uint32_t k2 = order ? uint32_t(order[t]) : t;
    const uint32_t src = order ? t : k2;
    cf a[Q], wq[Q];
#pragma unroll
    for (uint32_t j = 0; j < Q; ++j) wq[j] = unit_root(j, Q, false);
#pragma unroll
    for (uint32_t n1 = 0; n1 < Q; ++n1)
        a[n1] = cf_mul(A[(size_t(item) * Q + n1) * Nb + src], unit_root(uint32_t((uint64_t(n1) * k2) % N), N, false));
    const uint32_t place = order ? t : (paired ? uint32_t(CL::slot(int(k2))) : k2);
#pragma unroll
    for (uint32_t k1 = 0; k1 < Q; ++k1) {
        cf acc = a[0];
#pragma unroll
        for (uint32_t n1 = 1; n1 < Q; ++n1) acc = cf_add(acc, cf_mul(a[n1], wq[(n1 * k1) % Q]));
        X[(size_t(item) * Q + k1) * Nb + place] = acc;
    }
}

// The code-side factor of sub-transform n1, once per handle (codes are static): comb[p][n1][k1][pos] =
// conj(C[p][k1][pos]) * W_Q^{-n1 k1} * W_N^{-n1 k2(pos)} — elementwise on the paired layout, so positions carry over.
template <class PL, uint32_t Q>
__global__ __launch_bounds__(256) void comp_code_comb_kernel(const cf* __restrict__ code_paired, const cf* __restrict__ twn,
                                                             cf* __restrict__ comb) {
    constexpr uint32_t Nb = PL::N;
    const uint32_t pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= Nb) return;
    const uint32_t p = blockIdx.y / Q, n1 = blockIdx.y % Q;
    const cf t = twn[size_t(n1) * Nb + pos];
#pragma unroll
    for (uint32_t k1 = 0; k1 < Q; ++k1) {
        const cf c = code_paired[(size_t(p) * Q + k1) * Nb + pos];
        const cf g = cf_mul(cf_mul(cf_make(c.x, -c.y), unit_root((n1 * k1) % Q, Q, true)), t);
        comb[((size_t(p) * Q + n1) * Q + k1) * Nb + pos] = g;
    }
}

// Forward step 2 folded into the code-side table (base plans whose sub-transforms leave in storage order, `order` != null): an input of
// sub-transform n1 is sum_k1 X[k1][pos] comb[n1][k1][pos] with X[k1][pos] = sum_n1' W_Q^{n1' k1} W_N^{n1' k2} A[n1'][pos], i.e.
// sum_n1' A[n1'][pos] * comb2[n1][n1'][pos],  comb2[n1][n1'][pos] = W_N^{n1' k2(pos)} * sum_k1 W_Q^{n1' k1} comb[n1][k1][pos] — the same
// Q loads and products per element, read from the forward sub-transforms as they stand: no comp_fwd_post_kernel per dwell (12 us of
// the 360 at the configs[3] Galileo geometry, and a kernel boundary).
template <class PL, uint32_t Q>
__global__ __launch_bounds__(256) void comp_fold_post_kernel(const cf* __restrict__ comb, const uint16_t* __restrict__ order, cf* __restrict__ comb2) {
    constexpr uint32_t Nb = PL::N, N = Q * Nb;
    const uint32_t pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= Nb) return;
    const uint32_t k2 = order[pos];
    const size_t base = size_t(blockIdx.y) * Q * Nb + pos;          // blockIdx.y = code * Q + n1
    cf c[Q];
#pragma unroll
    for (uint32_t k1 = 0; k1 < Q; ++k1) c[k1] = comb[base + size_t(k1) * Nb];
#pragma unroll
    for (uint32_t n1p = 0; n1p < Q; ++n1p) {
        cf acc = c[0];
#pragma unroll
        for (uint32_t k1 = 1; k1 < Q; ++k1) acc = cf_add(acc, cf_mul(c[k1], unit_root((n1p * k1) % Q, Q, false)));
        comb2[base + size_t(n1p) * Nb] = cf_mul(acc, unit_root(uint32_t((uint64_t(n1p) * k2) % N), N, false));
    }
}

// ------------------------------------------------------------------------------------ inverse, fused
// One workgroup per (worker, bin).  spectra [d][m][k1][paired k2], code [p][k1][paired k2], twn [n1][paired k2] =
// W_N^{-n1 k2} (e^{+...}: inverse).
// PLANES (gm_acq_cfg.strict_sum_order): every accumulated power value also goes to planes[(p * n_bins + d) * N + its natural index]
// (an instantiation of its own: the shipped kernel carries nothing of it)
template <class PLX, uint32_t Q, bool PLANES = false>
__global__ __launch_bounds__(CompPlanOf<PLX>::type::T, CompPlanOf<PLX>::type::WAVES_PER_EU) void comp_corr_kernel(
    const cf* __restrict__ spectra, const cf* __restrict__ code_fft, const cf* __restrict__ twn, const cf* __restrict__ tw_inv,
    float* __restrict__ mmax, uint32_t* __restrict__ margmax, float* __restrict__ msum,
    const uint32_t* __restrict__ worker_list, int n_workers, int n_bins, int n_int, int cb, int rows_max, float* __restrict__ planes = nullptr) {
    using PL = typename CompPlanOf<PLX>::type;      // the base size's plan for this path (acq_device.h): plain or hybrid
    constexpr bool HYB = CorrMode<PL>::HYBRID;
    static_assert(!CorrMode<PL>::PFA, "composite bases: plain or hybrid correlation plans");
    static_assert(PL::IT0 == 1, "composite base plans: one pass-0 butterfly per lane");
    constexpr bool PAIRED = PairLayout<PL>::PAIRED;      // first radix <= 25: 16-byte pair loads; else (16368, 16384) plain 8-byte loads
    // Equal contiguous share of the bin-major item list per XCD (blocks b and b + 8 share an XCD: speed only).  Inside an XCD
    // the share — a strip of rows_max Doppler bins, ragged at both ends — is walked in blocks of `cb` workers: the ~32
    // workgroups resident on the XCD at a time then cover cb workers x all of the strip's bins, so every combined code table
    // (Q*N*8 bytes per worker) is shared by ~5 workgroups and every bin's spectra by cb of them, instead of 32 workers x 1 bin
    // (each table read by one workgroup only: 1.59 GB of fabric traffic per dwell at the configs[3] Galileo geometry).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int items = n_bins * n_workers, share = (items + 7) >> 3;
    const int it_lo = xcd * share, it_hi = it_lo + share < items ? it_lo + share : items;
    int item;
    if (cb <= 0) {
        item = it_lo + slot;
        if (slot >= share || item >= it_hi) return;
    } else {
        const int d_lo = it_lo / n_workers, per_blk = rows_max * cb;
        const int blk = slot / per_blk, rem = slot - blk * per_blk, dr = rem / cb, w = blk * cb + (rem - dr * cb);
        item = (d_lo + dr) * n_workers + w;
        if (w >= n_workers || item < it_lo || item >= it_hi) return;
    }
    const int d = item / n_workers, p = int(worker_list[item - d * n_workers]);

    __shared__ cf lds[PL::LDS_ELEMS + PL::TW_TOTAL];
    cf* tw = lds + PL::LDS_ELEMS;
    const int tid = threadIdx.x;
    if constexpr (!HYB) load_twiddles<PL>(tw, tw_inv, tid);
    constexpr int Nb = PL::N, NB0 = PL::NB(0), NBL = PL::NB(PL::NP - 1), NPAIR = PL::R0 / 2;
    constexpr bool ODD0 = (PL::R0 & 1) != 0;
    constexpr uint32_t N = Q * uint32_t(Nb);
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(spectra + size_t(d) * n_int * N, unsigned(n_int) * N * 8u);
    // code_fft: the combined tables [code][n1][k1][pos] of comp_code_comb_kernel
    const int oob = 0x7ffffff0;                              // lanes without a pass-0 butterfly: dropped by the range check
    const int v16 = tid < NB0 ? tid * 16 : oob, v8 = tid < NB0 ? tid * 8 : oob;
    auto lo = [](u32x4 v) { return cf_make(__uint_as_float(v.x), __uint_as_float(v.y)); };
    auto hi = [](u32x4 v) { return cf_make(__uint_as_float(v.z), __uint_as_float(v.w)); };

    float bv = 0.0f, sum = 0.0f;
    uint32_t bi = 0xffffffffu;
    for (uint32_t n1 = 0; n1 < Q; ++n1) {
        const __amdgpu_buffer_rsrc_t crs = make_rsrc(code_fft + (size_t(p) * Q + n1) * N, N * 8u);
        float acc[PL::ITL][PL::RL];
#pragma unroll
        for (int it = 0; it < PL::ITL; ++it)
#pragma unroll
            for (int r = 0; r < PL::RL; ++r) acc[it][r] = 0.0f;
        for (int m = 0; m < n_int; ++m) {
            // vals[r] = sum over k1 of X[m][k1][k] * comb[n1][k1][k]: the Q-point inverse DFT row, the inverse twiddle and
            // conj(code) (:184-186) are all inside the table
            cf vals[PL::R0];
            if constexpr (!PAIRED) {
#pragma unroll
                for (int r = 0; r < PL::R0; ++r) {
                    cf s0 = cf_make(0.f, 0.f);
#pragma unroll
                    for (uint32_t k1 = 0; k1 < Q; ++k1) {
                        const cf x1 = buf_load_cf(xrs, v8, ((m * int(Q) + int(k1)) * Nb + r * NB0) * 8);
                        const cf c1 = buf_load_cf(crs, v8, (int(k1) * Nb + r * NB0) * 8);
                        const cf p0 = cf_mul(x1, c1);
                        s0 = k1 == 0 ? p0 : cf_add(s0, p0);
                    }
                    vals[r] = s0;
                }
            }
#pragma unroll
            for (int rp = 0; rp < (PAIRED ? NPAIR : 0); ++rp) {
                cf s0 = cf_make(0.f, 0.f), s1 = s0;
#pragma unroll
                for (uint32_t k1 = 0; k1 < Q; ++k1) {
                    const u32x4 x4 = __builtin_amdgcn_raw_buffer_load_b128(xrs, v16, ((m * int(Q) + int(k1)) * Nb + rp * 2 * NB0) * 8, 0);
                    const u32x4 c4 = __builtin_amdgcn_raw_buffer_load_b128(crs, v16, (int(k1) * Nb + rp * 2 * NB0) * 8, 0);
                    const cf p0 = cf_mul(lo(x4), lo(c4)), p1 = cf_mul(hi(x4), hi(c4));
                    if (k1 == 0) { s0 = p0; s1 = p1; }
                    else { s0 = cf_add(s0, p0); s1 = cf_add(s1, p1); }
                }
                vals[2 * rp] = s0;
                vals[2 * rp + 1] = s1;
            }
            if constexpr (ODD0 && PAIRED) {
                cf s0 = cf_make(0.f, 0.f);
#pragma unroll
                for (uint32_t k1 = 0; k1 < Q; ++k1) {
                    const cf x1 = buf_load_cf(xrs, v8, ((m * int(Q) + int(k1)) * Nb + 2 * NPAIR * NB0) * 8);
                    const cf c1 = buf_load_cf(crs, v8, (int(k1) * Nb + 2 * NPAIR * NB0) * 8);
                    const cf p0 = cf_mul(x1, c1);
                    s0 = k1 == 0 ? p0 : cf_add(s0, p0);
                }
                vals[PL::R0 - 1] = s0;
            }
            lds_transform<PL, true>([&](int, int r) { return vals[r]; },
                                    [&](int it, int r, cf v) { acc[it][r] = acc[it][r] + (v.x * v.x + v.y * v.y); },   // += norm_sqr() (:190-192)
                                    lds, tw, tid);
        }
        // this sub-transform's outputs are y[Q n2 + n1]: fold them into the lane's running first strict maximum / sum
#pragma unroll
        for (int it = 0; it < PL::ITL; ++it) {
            bool mine;
            if constexpr (HYB) mine = PL::last_active(tid); else mine = tid + it * PL::T < NBL;
            if (mine) {
#pragma unroll
                for (int r = 0; r < PL::RL; ++r) {
                    uint32_t e;                                  // this slot's element of the sub-transform's output
                    if constexpr (HYB) e = uint32_t(PL::out_index(tid, r)); else e = uint32_t((tid + it * PL::T) + r * NBL);
                    take_better(bv, bi, acc[it][r], Q * e + n1);
                    sum += acc[it][r];
                    if constexpr (PLANES) planes[(size_t(p) * n_bins + d) * N + (Q * e + n1)] = acc[it][r];
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const uint32_t oi = uint32_t(__shfl_xor(int(bi), off, 64));
        const float os = __shfl_xor(sum, off, 64);
        take_better(bv, bi, ov, oi);
        sum += os;
    }
    __syncthreads();   // everyone is done with the LDS transform buffer: reuse it as scratch
    float* sv = reinterpret_cast<float*>(lds);
    uint32_t* si = reinterpret_cast<uint32_t*>(lds) + 64;
    float* ss = reinterpret_cast<float*>(lds) + 128;
    const int wave = tid >> 6, lane = tid & 63;
    constexpr int NW = PL::T / 64;
    if (lane == 0) { sv[wave] = bv; si[wave] = bi; ss[wave] = sum; }
    __syncthreads();
    if (tid == 0) {
        float fv = sv[0], fs = ss[0];
        uint32_t fi = si[0];
        for (int w = 1; w < NW; ++w) { take_better(fv, fi, sv[w], si[w]); fs += ss[w]; }
        if (fi == 0xffffffffu) fi = 0;   // all-NaN / all-zero plane: the reference keeps (0.0, 0)
        const size_t o = size_t(p) * n_bins + d;
        mmax[o] = fv; margmax[o] = fi; msum[o] = fs;
    }
}

template <class PL, uint32_t Q> struct CompLaunch {
    using CP = typename CompPlanOf<PL>::type;
    static int fill_order(uint16_t* order) { return fill_order_table<CP>(order); }
    static void relayout(hipStream_t st, const cf* nat, cf* stored, int n_blocks) {
        hipLaunchKernelGGL(relayout_kernel<CP>, dim3(n_blocks * 4 < 1024 ? n_blocks * 4 : 1024), dim3(256), 0, st, nat, stored, n_blocks);
    }
    static void fwd_sub(hipStream_t st, const void* samples, int fmt, const cf* tables, const int8_t* code_samples,
                        const cf* tw_fwd, cf* A, uint32_t n_items, uint32_t n_int, const uint16_t* order) {
        hipLaunchKernelGGL(comp_fwd_sub_kernel<PL>, dim3(n_items * Q), dim3(MixPlanOf<PL>::type::T), 0, st, samples, fmt, tables, code_samples,
                           tw_fwd, A, Q, n_int, order);
    }
    static void fwd_post(hipStream_t st, const cf* A, cf* X, uint32_t n_items, int paired, const uint16_t* order) {
        hipLaunchKernelGGL((comp_fwd_post_kernel<PL, Q>), dim3((PL::N + CT - 1) / CT, n_items), dim3(CT), 0, st, A, X, paired, order);
    }
    static void corr(hipStream_t st, const cf* spectra, const cf* code_paired, const cf* twn, const cf* tw_inv, float* mmax,
                     uint32_t* margmax, float* msum, const uint32_t* worker_list, int n_workers, int n_bins, int n_int, float* planes) {
        if (n_workers <= 0) return;
        static const int cb_env = diag_int("GM_COMP_CB", -1);   // diagnostic: workers per block (0: plain order)
        const int items = n_workers * n_bins, share = (items + 7) / 8;
        // measured at 36 codes x 41 bins x N = 2 x 16000: plain order 0.351 ms, blocks of 4 / 8 / 12 workers 0.313 / 0.314 / 0.314,
        // blocks of 2 or 6 no gain (6 x ~5 bins is exactly the resident set: every workgroup of a round then starts a new table)
        const int cb = cb_env >= 0 ? cb_env : (n_workers > 4 ? 4 : 0);
        // bins a strip can touch: a share of `share` items starting anywhere in a row
        const int rows_max = (share + n_workers - 2) / n_workers + 1;
        const int slots = cb > 0 ? ((n_workers + cb - 1) / cb) * rows_max * cb : share;
        if (planes) {                       // strict_sum_order: the variants that also store the power planes (launch_plane_strict_sum follows)
            if constexpr (CompWs<CP>::USE)
                hipLaunchKernelGGL((comp_corr_ws_kernel<CP, Q, false, true>), dim3(8 * slots), dim3(1024), 0, st, spectra, code_paired, mmax, margmax, msum,
                                   worker_list, n_workers, n_bins, n_int, cb, rows_max, planes);
            else
                hipLaunchKernelGGL((comp_corr_kernel<PL, Q, true>), dim3(8 * slots), dim3(PL::T), 0, st, spectra, code_paired, twn, tw_inv,
                                   mmax, margmax, msum, worker_list, n_workers, n_bins, n_int, cb, rows_max, planes);
            return;
        }
        if constexpr (CompWs<CP>::USE)      // base 16000: the wave-specialised kernel (acq_comp_ws.h)
            hipLaunchKernelGGL((comp_corr_ws_kernel<CP, Q>), dim3(8 * slots), dim3(1024), 0, st, spectra, code_paired, mmax, margmax, msum,
                               worker_list, n_workers, n_bins, n_int, cb, rows_max, static_cast<float*>(nullptr));
        else
            hipLaunchKernelGGL((comp_corr_kernel<PL, Q>), dim3(8 * slots), dim3(PL::T), 0, st, spectra, code_paired, twn, tw_inv,
                               mmax, margmax, msum, worker_list, n_workers, n_bins, n_int, cb, rows_max, static_cast<float*>(nullptr));
    }
    // W_N^{-n1 k2} (inverse sign) for n1 < Q, in the paired position of k2: built in double on the host
    static void fill_twn(cf* out) {
        const uint32_t Nb = PL::N, N = Q * Nb;
        for (uint32_t n1 = 0; n1 < Q; ++n1)
            for (uint32_t k2 = 0; k2 < Nb; ++k2) {
                const double a = 2.0 * ct::kPi * double((uint64_t(n1) * k2) % N) / double(N);
                out[size_t(n1) * Nb + CorrLayout<CP>::slot(int(k2))] = cf_make(float(::cos(a)), float(::sin(a)));
            }
    }
    static void comb(hipStream_t st, const cf* code_paired, const cf* twn, cf* out, uint32_t n_codes) {
        hipLaunchKernelGGL((comp_code_comb_kernel<PL, Q>), dim3((PL::N + 255) / 256, n_codes * Q), dim3(256), 0, st, code_paired, twn, out);
    }
    static void fold_post(hipStream_t st, const cf* comb_in, const uint16_t* order, cf* comb2, uint32_t n_codes) {
        hipLaunchKernelGGL((comp_fold_post_kernel<PL, Q>), dim3((PL::N + 255) / 256, n_codes * Q), dim3(256), 0, st, comb_in, order, comb2);
    }
    static constexpr CompOps ops() { return CompOps{PL::N, int(Q), &fwd_sub, &fwd_post, &corr, &fill_twn, &comb, &fill_order, &relayout, &fold_post}; }
};
}  // namespace

}  // namespace gm
// 16368's generic plan runs here with its twiddles (AsPlain: the prime-factor form is the fused kernel's)
namespace gm { template <> struct CompPlanOf<Plan16368> { using type = AsPlain<Plan16368>; }; }
namespace gm { template <> struct CompPlanOf<Plan8184> { using type = AsPlain<Plan8184>; }; }
namespace gm { template <> struct CompPlanOf<Plan8192> { using type = Plan8192; }; }     // the registered four-pass plan, not the fused kernel's [16, 32, 16]
#ifdef GM_COMP_PLAIN_16000        // (A/B switch: the generic kernel on the plain [25, 20, 32] plan, what rounds 2 - 3 shipped)
namespace gm { template <> struct CompPlanOf<Plan16000> { using type = Plan16000; }; }
#endif
namespace gm {
// base plans of the composite sizes: one pass-0 butterfly per lane; first radix <= 25 -> paired 16-byte loads, else 8-byte loads
#define GM_COMP_ENTRY(PL)                                                                            \
    CompLaunch<PL, 2>::ops(), CompLaunch<PL, 3>::ops(), CompLaunch<PL, 4>::ops(), CompLaunch<PL, 5>::ops(), \
        CompLaunch<PL, 6>::ops(), CompLaunch<PL, 8>::ops(),
#define GM_COMP_ONE(PL, Q) CompLaunch<PL, Q>::ops(),
// find_comp takes the FIRST entry whose Q x base is the size, and a size with an in-LDS plan never comes here: of the 54 (base, Q)
// pairs only 31 can be chosen — 2 x 8000 is 16000, 4 x 8000 is 2 x 16000, 6 x 6000 ... — and only those are instantiated (round 6: the
// kernels nobody can launch were a fifth of this file's code objects, the most register-hungry ones among them: 8 x 6000, 8 x 8000).
// GM_COMP_ALL_Q (a diagnostic build) brings every pair back for GM_COMP_BASE's A/B of two decompositions of one size.
#ifdef GM_COMP_ALL_Q
static const CompOps g_comp[] = {GM_COMP_ENTRY(Plan16384) GM_COMP_ENTRY(Plan16368) GM_COMP_ENTRY(Plan16000) GM_COMP_ENTRY(Plan8000) GM_COMP_ENTRY(Plan8192) GM_COMP_ENTRY(Plan8184)
                                     GM_COMP_ENTRY(Plan6000) GM_COMP_ENTRY(Plan5000) GM_COMP_ENTRY(Plan4000)};
#else
static const CompOps g_comp[] = {GM_COMP_ENTRY(Plan16384) GM_COMP_ENTRY(Plan16368) GM_COMP_ENTRY(Plan16000)
                                 GM_COMP_ONE(Plan8000, 3) GM_COMP_ONE(Plan8000, 5) GM_COMP_ONE(Plan8192, 3) GM_COMP_ONE(Plan8192, 5)
                                 GM_COMP_ONE(Plan8184, 3) GM_COMP_ONE(Plan8184, 5)
                                 GM_COMP_ONE(Plan8184, 2)     /* reachable through GM_COMP_BASE only: the diagnostics-switch test runs 16368 as 2 x 8184 */
                                 GM_COMP_ONE(Plan6000, 3) GM_COMP_ONE(Plan6000, 5) GM_COMP_ONE(Plan6000, 6)
                                 GM_COMP_ONE(Plan5000, 4) GM_COMP_ONE(Plan5000, 5)};
#endif

// strict_sum_order on the composite path.  is_good_satellite's plane sum in the reference's own order (do_acquisition.rs:229-235): eight
// running f32 sums over chunks_exact(8) — lane l adds power[8c + l] for c = 0, 1, ... — then reduce_sum, an ordered add of the eight
// lanes starting from -0.0.  One workgroup per plane; the plane travels through LDS in chunks of 8192 values (coalesced loads), eight
// lanes of wave 0 walk each chunk sequentially and carry their sums from chunk to chunk.
__global__ __launch_bounds__(256) void plane_strict_sum_kernel(const float* __restrict__ planes, float* __restrict__ msum,
                                                                const uint32_t* __restrict__ worker_list, int n_bins, uint32_t N) {
    constexpr uint32_t CH = 8192;
    __shared__ float pl[CH];
    const int tid = threadIdx.x;
    const size_t o = size_t(worker_list[blockIdx.y]) * n_bins + blockIdx.x;
    const float* src = planes + o * N;
    const uint32_t n8 = N & ~7u;                                    // chunks_exact(8) drops a tail
    float ls = 0.0f;                                                // f32x8::splat(0.0): lane tid < 8 holds lane tid of it
    for (uint32_t c0 = 0; c0 < n8; c0 += CH) {
        const uint32_t len = n8 - c0 < CH ? n8 - c0 : CH;
        for (uint32_t i = tid; i < len; i += 256) pl[i] = src[c0 + i];
        __syncthreads();
        if (tid < 8) {
#pragma unroll 8
            for (uint32_t c = 0; c < len / 8; ++c) ls = ls + pl[c * 8 + tid];
        }
        __syncthreads();
    }
    if (tid < 64) {
        float t = -0.0f;                                            // simd_reduce_add_ordered(v, -0.0)
#pragma unroll
        for (int l = 0; l < 8; ++l) t = t + __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ls), l));
        if (tid == 0) msum[o] = t;
    }
}
void launch_plane_strict_sum(hipStream_t st, const float* planes, float* msum, const uint32_t* worker_list, int n_workers, int n_bins, uint32_t N) {
    if (n_workers <= 0 || n_bins <= 0) return;
    hipLaunchKernelGGL(plane_strict_sum_kernel, dim3(n_bins, n_workers), dim3(256), 0, st, planes, msum, worker_list, n_bins, N);
}

// N = Q * Nb: the largest base plan first (fewest sub-transform passes over the spectra)
const CompOps* find_comp(uint32_t n) {
    static const int base_env = diag_int("GM_COMP_BASE", 0);   // diagnostic: force the base size (A/B of 2 x 16000 against 4 x 8000)
    if (base_env > 0)
        for (const CompOps& c : g_comp)
            if (c.nb == base_env && uint32_t(c.nb) * uint32_t(c.q) == n) return &c;
    for (const CompOps& c : g_comp)
        if (uint32_t(c.nb) * uint32_t(c.q) == n) return &c;
    return nullptr;
}

}  // namespace gm
