// acq_corr_wsh.h — stage C for a hybrid plan with two workgroups per CU (N = 8000 = 125 * 64, the bench workload) as a WAVE-SPECIALISED
// kernel: acq_comp_ws.h's division of a transform between two groups of waves, inside each of the two 512-lane workgroups a CU holds.
// Same arguments, same item map, same results contract as acq_corr_kernel.
//
//   * waves 0 - 3 run the radix-25 middle pass (five wave-slots: wave 0 takes two), a barrier of their own (an LDS word), and the
//     radix-16 last pass (eight wave-slots: two per wave, one after the other) — they own the power sums (2 x 16 per lane);
//   * waves 4 - 7 do pass 0 and nothing else: the loads of the spectrum and the code spectrum, x conj(code), the radix-20
//     Good-Thomas butterfly's first half — 400 butterflies on 256 lanes, two per lane on lanes 0 - 143 (part of the second one's
//     values parked in LDS) — registers only until the image is free.  From B2 of transform m they go straight to the loads of m + 1.
// Two workgroup barriers per transform: B1 (the image is free) and B2 (pass-0 image complete).
#pragma once
#include "acq_device.h"
#include "acq_comp_ws.h"

namespace gm {

// one group of the radix-20 butterfly's first half: A = 4 inputs = two stored row pairs of both arrays (x and the code spectrum)
template <class PL, int A, bool REF_MUL> struct WshGroup {
    u32x4 x[A / 2], c[A / 2];
    __device__ __forceinline__ void request(__amdgpu_buffer_rsrc_t xrs, __amdgpu_buffer_rsrc_t crs, int v16, int xoff, int g) {
#pragma unroll
        for (int h = 0; h < A / 2; ++h) {
            const int st = g * A + 2 * h;                     // stored rows st, st + 1 = inputs 2h, 2h + 1 of group g
            x[h] = __builtin_amdgcn_raw_buffer_load_b128(xrs, v16, (xoff + st * PL::NB(0)) * 8, 0);
            c[h] = __builtin_amdgcn_raw_buffer_load_b128(crs, v16, (st * PL::NB(0)) * 8, 0);
        }
    }
    // result_buf[i] *= conj(code[i])  (:184-186); REF_MUL: num-complex's own unfused form (gm_acq_cfg.reference_products)
    static __device__ __forceinline__ cf prod(float ax, float ay, float gx, float gy) {
        const float cx = gx, cy = -gy;
        if constexpr (REF_MUL) return cf_make(ax * cx - ay * cy, ax * cy + ay * cx);
        else return cf_make(__builtin_fmaf(ax, cx, -(ay * cy)), __builtin_fmaf(ax, cy, ay * cx));
    }
    __device__ __forceinline__ void inputs(cf (&t)[A]) const {
#pragma unroll
        for (int h = 0; h < A / 2; ++h) {
            t[2 * h] = prod(__uint_as_float(x[h].x), __uint_as_float(x[h].y), __uint_as_float(c[h].x), __uint_as_float(c[h].y));
            t[2 * h + 1] = prod(__uint_as_float(x[h].z), __uint_as_float(x[h].w), __uint_as_float(c[h].z), __uint_as_float(c[h].w));
        }
    }
};
template <class PL, bool REF_MUL, int DEPTH> struct WshStream {
    using B0 = Bfly<PL::R0, true>;
    static_assert(B0::KIND == 2 && B0::A % 2 == 0 && B0::B >= DEPTH, "first radix: Good-Thomas, groups of an even number of rows");
    WshGroup<PL, B0::A, REF_MUL> buf[DEPTH];
    __device__ __forceinline__ void start(__amdgpu_buffer_rsrc_t xrs, __amdgpu_buffer_rsrc_t crs, int v16, int xoff) {
#pragma unroll
        for (int g = 0; g < DEPTH; ++g) buf[g].request(xrs, crs, v16, xoff, g);
    }
    template <class Emit>
    __device__ __forceinline__ void run(Emit&& emit, __amdgpu_buffer_rsrc_t xrs, __amdgpu_buffer_rsrc_t crs, int v16, int xoff) {
#pragma unroll
        for (int g = 0; g < B0::B; ++g) {
            __builtin_amdgcn_sched_barrier(0);
            cf t[B0::A];
            buf[g % DEPTH].inputs(t);
            if (g + DEPTH < B0::B) buf[g % DEPTH].request(xrs, crs, v16, xoff, g + DEPTH);
            Dft<B0::A, true>::run(t);
#pragma unroll
            for (int k1 = 0; k1 < B0::A; ++k1) {
                asm volatile("" : "+v"(t[k1].x), "+v"(t[k1].y));      // (pinned: see acq_comp_ws.h)
                emit(g * B0::A + k1, t[k1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

template <class PLX, bool REF_MUL>
__device__ __forceinline__ void acq_corr_wsh_body(
    cf* lds, const cf* __restrict__ spectra, const cf* __restrict__ code_fft,
    float* __restrict__ mmax, uint32_t* __restrict__ margmax, float* __restrict__ msum,
    const uint32_t* __restrict__ worker_list, int n_workers, int n_bins, int n_int, int map_mode,
    int split_from, int split_k, int split_items, float* __restrict__ split_scratch, uint32_t* __restrict__ split_counter,
    int strict_sum) {
    using PL = typename CorrPlanOf<PLX>::type;
    using F = Fft<PL, true>;
    using PR = PairRows<PL>;
    constexpr int T = PL::T, R0 = PL::R0, RL = PL::RL, NB0 = PL::NB(0);
    constexpr int WB = 4, WA = T / 64 - WB, NA = 64 * WA, NA2 = NB0 - NA, MW = PL::A1 * PL::GW1, LW = PL::B1 * PL::GW2;
    constexpr int AIT = LW / WB, ARL = RL, TM = 64 * WB;        // power slots per lane: AIT last-pass butterflies x RL outputs, on TM lanes
    static_assert(PL::HYBRID && T == 512 && PairLayout<PL>::PAIRED && (R0 & 1) == 0, "a hybrid plan on 512 lanes, rows in pairs");
    static_assert(LW == AIT * WB && MW >= WB && MW <= 2 * WB && NA2 >= 0 && NA2 <= NA, "last pass: AIT wave-slots per wave on waves 0 .. 3");
    static_assert(PR::nat(0) == 0 && PR::nat(1) == Bfly<R0, true>::B && PR::nat(Bfly<R0, true>::A) == Bfly<R0, true>::A % R0, "rows stored in the butterfly's consumption order");
    // ---- the item of this workgroup: acq_corr_kernel's map (acq_kernels.hip), the same modes
    const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3;
    int slot = wslot, part = 0, parts = 1;
    if (wslot >= split_from) {
        const int h = wslot - split_from;
        slot = split_from + h / split_k;
        part = h % split_k;
        parts = split_k;
    }
    int d, p;
    if (map_mode == 0) {
        const int items = n_bins * n_workers, share = (items + 7) >> 3;
        const int item = xcd * share + slot;
        if (slot >= share || item >= items) return;
        d = item / n_workers;
        p = int(worker_list[item - d * n_workers]);
    } else if (map_mode >= 16) {
        const int cb = map_mode >> 4, rows_max = map_mode & 15;
        const int items = n_bins * n_workers, share = (items + 7) >> 3;
        const int it_lo = xcd * share, it_hi = it_lo + share < items ? it_lo + share : items;
        const int d_lo = it_lo / n_workers, per_blk = rows_max * cb;
        const int blk = slot / per_blk, rem = slot - blk * per_blk, dr = rem / cb, w = blk * cb + (rem - dr * cb);
        const int item = (d_lo + dr) * n_workers + w;
        if (w >= n_workers || item < it_lo || item >= it_hi) return;
        d = d_lo + dr;
        p = int(worker_list[w]);
    } else if (map_mode == 1) {
        d = xcd + 8 * (slot / n_workers);
        if (d >= n_bins) return;
        p = int(worker_list[slot % n_workers]);
    } else {
        const int q = n_bins >> 3, whole = q * n_workers;
        if (slot < whole) {
            d = xcd + 8 * (slot / n_workers);
            p = int(worker_list[slot % n_workers]);
        } else {
            const int left = (n_bins - 8 * q) * n_workers, each = (left + 7) >> 3;
            const int j = slot - whole, item = xcd * each + j;
            if (j >= each || item >= left) return;
            d = 8 * q + item / n_workers;
            p = int(worker_list[item % n_workers]);
        }
    }


    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
    const int tid = wave * 64 + (int(threadIdx.x) & 63);
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(spectra + size_t(d) * n_int * PL::N, unsigned(n_int) * PL::N * 8u);
    const __amdgpu_buffer_rsrc_t crs = make_rsrc(code_fft + size_t(p) * PL::N, PL::N * 8u);
    const int m_per = n_int / parts, m_begin = part * m_per, m_end = m_begin + m_per;
    __shared__ unsigned group_word;                           // comp_ws_wave_group_barrier of waves 0 .. WB - 1
    constexpr int NST = 12;                                   // values of a lane's second butterfly parked in LDS until the image is free
    __shared__ cf stage[NST * (NA2 > 0 ? NA2 : 1)];
    if (tid == 0) group_word = 0;
    __syncthreads();

    float acc[AIT][ARL];                                      // waves 0 .. 3: |y|^2 summed over the integrations (zero and untouched on the others)
#pragma unroll
    for (int it = 0; it < AIT; ++it)
#pragma unroll
        for (int r = 0; r < ARL; ++r) acc[it][r] = 0.0f;

    if (wave >= WB) {
        // ---------------------------------------------------------------- pass 0 only
        const int a0 = tid - 64 * WB;
        WshStream<PL, REF_MUL, 2> st;
        st.start(xrs, crs, a0 * 16, m_begin * PL::N);
        for (int m = m_begin; m < m_end; ++m) {
            int a = a0;
            asm volatile("" : "+v"(a));                       // (addresses derived from the lane number are recomputed, not kept: acq_comp_ws.h)
            const bool two = a < NA2;
            cf va[1][R0], vb[1][R0];
            st.run([&](int i, cf val) { va[0][i] = val; }, xrs, crs, a * 16, m * PL::N);
            if (two) {
                st.start(xrs, crs, (NA + a) * 16, m * PL::N);
                st.run([&](int i, cf val) {
                    if (i < NST) stage[i * NA2 + a] = val;
                    else vb[0][i] = val;
                }, xrs, crs, (NA + a) * 16, m * PL::N);
            }
            if (m + 1 < m_end) st.start(xrs, crs, a * 16, (m + 1) * PL::N);      // in flight while this wave waits for B1 and scatters
            __syncthreads();                                  // B1: the image is free
            asm volatile("" : "+v"(a));
            F::pass0_stage2(va, lds, a);
            if (two) {
#pragma unroll
                for (int i = 0; i < NST; ++i) vb[0][i] = stage[i * NA2 + a];
                F::pass0_stage2(vb, lds, NA + a);
            }
            __syncthreads();                                  // B2: pass-0 image complete
        }
    } else {
        // ---------------------------------------------------------------- middle pass + last pass; owns the power sums
        unsigned gen = 0;
        for (int m = m_begin; m < m_end; ++m) {
            __syncthreads();                                  // B1
            __syncthreads();                                  // B2
            comp_ws_middle_pass<PL>(lds, tid);
            if (wave < MW - WB) comp_ws_middle_pass<PL>(lds, tid + 64 * WB);
            comp_ws_wave_group_barrier(&group_word, ++gen * WB);
#pragma unroll
            for (int it = 0; it < AIT; ++it) {
                // acc += norm_sqr() (:190-192): `acc + p` with p complete, so that the planes of a cut item (each 0 + p) merge to the same words
                comp_ws_last_pass<PL>(lds, tid + 64 * WB * it, [&](int, int r, cf v) {
                    if constexpr (REF_MUL) acc[it][r] = acc[it][r] + (v.x * v.x + v.y * v.y);
                    else acc[it][r] = acc[it][r] + __builtin_fmaf(v.y, v.y, v.x * v.x);
                    asm volatile("" : "+v"(acc[it][r]));
                });
            }
        }
    }

    auto last_active = [&](int it) { return wave < WB && PL::last_active(tid + 64 * WB * it); };
    auto out_index = [&](int it, int r) { return PL::out_index(tid + 64 * WB * it, r); };
    auto slot_ok = [&](int) { return true; };
    constexpr int RL4 = ARL / 4;                                     // 16-byte groups of a lane's power values
    static_assert(ARL % 4 == 0, "power slots in 16-byte groups");
    const int mt = tid;                                             // lane number among the waves that own power sums (waves 0 .. WB - 1)

    if (parts > 1) {      // a part of a cut item (one integration): its plane goes out (see acq_corr_kernel for the protocol and the store-data guard)
        constexpr int SLAB = AIT * RL4 * 4 * TM;                    // floats per power plane, register order of the matrix lanes
        const size_t item_plane0 = (size_t(xcd) * split_items + (slot - split_from)) * size_t(n_int);
        const __amdgpu_buffer_rsrc_t srs = make_rsrc(split_scratch + item_plane0 * SLAB, unsigned(n_int) * SLAB * 4u);
        if (wave < WB) {
            const int voff = mt * 16 + part * SLAB * 4;
#pragma unroll
            for (int it = 0; it < AIT; ++it)
#pragma unroll
                for (int r4 = 0; r4 < RL4; ++r4) {
                    u32x4 v;
                    v.x = __float_as_uint(acc[it][4 * r4 + 0]); v.y = __float_as_uint(acc[it][4 * r4 + 1]);
                    v.z = __float_as_uint(acc[it][4 * r4 + 2]); v.w = __float_as_uint(acc[it][4 * r4 + 3]);
                    __builtin_amdgcn_raw_buffer_store_b128(v, srs, voff + (it * RL4 + r4) * TM * 16, 0, 16);   // sc1: write-through
                    asm volatile("s_nop 1" ::: "memory");
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        __shared__ int s_last;
        if (tid == 0) {
            uint32_t* cnt = split_counter + size_t(xcd) * split_items + (slot - split_from);
            const uint32_t old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (old == uint32_t(parts - 1)) ? 1 : 0;
        }
        __syncthreads();
        if (!s_last) return;
        if (wave < WB) {       // the last arriver: the n_int planes added in integration order (sc1 loads)
#pragma unroll
            for (int it = 0; it < AIT; ++it)
#pragma unroll
                for (int r = 0; r < ARL; ++r) acc[it][r] = 0.0f;
            for (int q = 0; q < n_int; ++q) {
#pragma unroll
                for (int it = 0; it < AIT; ++it)
#pragma unroll
                    for (int r4 = 0; r4 < RL4; ++r4) {
                        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srs, mt * 16, (q * SLAB + (it * RL4 + r4) * TM * 4) * 4, 16);
                        acc[it][4 * r4 + 0] += __uint_as_float(v.x); acc[it][4 * r4 + 1] += __uint_as_float(v.y);
                        acc[it][4 * r4 + 2] += __uint_as_float(v.z); acc[it][4 * r4 + 3] += __uint_as_float(v.w);
                    }
            }
        }
    }

    // strict_sum_order: is_good_satellite's plane sum in the reference's own order (:229-235) — see acq_corr_kernel
    float strict_total = 0.0f;
    if (strict_sum) {
        __syncthreads();
        float* pl = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            if (last_active(it)) {
#pragma unroll
                for (int r = 0; r < ARL; ++r)
                    if (slot_ok(r)) pl[out_index(it, r)] = acc[it][r];
            }
        }
        __syncthreads();
        if (tid < 64) {
            float ls = 0.0f;
            if (tid < 8) {
                constexpr int CHUNKS = PL::N / 8;
#pragma unroll 8
                for (int c = 0; c < CHUNKS; ++c) ls = ls + pl[c * 8 + tid];
            }
            float t = -0.0f;
#pragma unroll
            for (int l = 0; l < 8; ++l) t = t + __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ls), l));
            strict_total = t;
        }
    }

    // per-lane first strict maximum + partial sum: the scan runs on values, the element index is worked out for the winner alone
    // (every slot that holds the maximum when several do: then the lowest index wins — the reference's first strict maximum, :195-202)
    float bv = 0.0f, sum = 0.0f;
    uint32_t bi = 0xffffffffu;
    {
        int bs = -1, ties = 0;
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            if (last_active(it)) {
#pragma unroll
                for (int r = 0; r < ARL; ++r) {
                    const float v = acc[it][r];                     // (the slot that does not exist holds 0 and never wins)
                    if (v > bv) { bv = v; bs = it * ARL + r; }
                    sum += v;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            if (last_active(it)) {
#pragma unroll
                for (int r = 0; r < ARL; ++r) ties += (slot_ok(r) && acc[it][r] == bv) ? 1 : 0;
            }
        }
        if (ties == 1 && bs >= 0) {
            bi = uint32_t(out_index(bs / ARL, bs % ARL));
        } else if (ties >= 1) {
#pragma unroll
            for (int it = 0; it < AIT; ++it) {
                if (last_active(it)) {
#pragma unroll
                    for (int r = 0; r < ARL; ++r)
                        if (slot_ok(r) && acc[it][r] == bv) {
                            const uint32_t i = uint32_t(out_index(it, r));
                            bi = i < bi ? i : bi;
                        }
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
    const int lane = tid & 63;
    constexpr int NW = WB;
    if (lane == 0) { sv[wave] = bv; si[wave] = bi; ss[wave] = sum; }
    __syncthreads();
    if (tid == 0) {
        float fv = sv[0], fs = ss[0];
        uint32_t fi = si[0];
        for (int w = 1; w < NW; ++w) { take_better(fv, fi, sv[w], si[w]); fs += ss[w]; }
        if (fi == 0xffffffffu) fi = 0;   // all-NaN plane: the reference keeps (0.0, 0)
        const size_t o = size_t(p) * n_bins + d;
        mmax[o] = fv; margmax[o] = fi; msum[o] = strict_sum ? strict_total : fs;
    }
}


template <class PLX, bool REF_MUL>
__global__ __launch_bounds__(512, 4) void acq_corr_wsh_kernel(
    const cf* __restrict__ spectra, const cf* __restrict__ code_fft,
    float* __restrict__ mmax, uint32_t* __restrict__ margmax, float* __restrict__ msum,
    const uint32_t* __restrict__ worker_list, int n_workers, int n_bins, int n_int, int map_mode,
    int split_from, int split_k, int split_items, float* __restrict__ split_scratch, uint32_t* __restrict__ split_counter,
    int strict_sum) {
    __shared__ cf image[CorrPlanOf<PLX>::type::LDS_ELEMS];
    acq_corr_wsh_body<PLX, REF_MUL>(image, spectra, code_fft, mmax, margmax, msum, worker_list, n_workers, n_bins, n_int, map_mode, split_from, split_k,
                                    split_items, split_scratch, split_counter, strict_sum);
}

// which registered plans take this kernel, and the floats of one split power plane on it
template <class CP> struct Wsh { static constexpr bool USE = false; };
#ifdef GM_WSH_8000
template <> struct Wsh<HybridPlan<8000, 512, 5, 25, 4, 16>> { static constexpr bool USE = true; };
#endif
template <class CP> constexpr int wsh_split_slab() { return (CP::B1 * CP::GW2 / 4) * (CP::RL / 4) * 4 * 256; }

}  // namespace gm
