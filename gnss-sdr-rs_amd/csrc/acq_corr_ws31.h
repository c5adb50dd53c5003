// acq_corr_ws31.h — stage C for transform sizes whose prime-factor plan ends in a radix-31 pass (N = 16368 = 33 * 16 * 31, the
// reference's own geometry: fs = 16.3676 MHz, do_acquisition.rs:405-436), as a WAVE-SPECIALISED kernel.  Included by
// acq_kernels.hip; same arguments, same item map, same results contract as acq_corr_kernel (which stays the kernel of every
// other size, and of this one under gm_acq_debug_stamps).
//
// Why a kernel of its own.  The image of one transform is 131 KB of a CU's 160 KB of LDS: one workgroup per CU, so nothing hides a
// barrier or a load of the generic kernel (12 waves in lockstep: 428 us per 32 x 29 x 10 launch, vector issue 42 %, 43 % parked).
// A second transform in flight per CU does not fit — not in LDS (a re / im half-image exchange needs two 512-lane workgroups, i.e.
// 128 registers per lane, where 32 complex values + 32 power sums per lane leave nothing for a radix-31 or radix-33 butterfly),
// not as 2 x 8184 through the composite path (0.82 ms measured) — DESIGN.md §4.2 has the table.  What does fit is a second
// transform in flight INSIDE the workgroup, on different waves and different execution units:
//
//   * the radix-31 last pass — 528 dense 31 x 31 DFTs, the one dense contraction on this path — runs on the MATRIX pipe.  In the
//     symmetric form of a prime-length DFT (fft_core.h DftPrime) it is four real 16 x 16 matrix products per butterfly,
//         ca.re = C a.re, ca.im = C a.im, sb.re = S b.re, sb.im = S b.im;   y[q] = ca[q] + j sb[q],  y[31 - q] = ca[q] - j sb[q]
//     with a[0] = u[0], a[k] = u[k] + u[31 - k], b[0] = 0, b[k] = u[k] - u[31 - k], C[q][k] = cos(2 pi q k / 31), S[q][k] =
//     sin(2 pi q k / 31), q, k = 0 .. 15 (row 0 of C is all ones: y[0]; column 0 is the u[0] term: the matrices are exactly
//     16 x 16, nothing is padded).  v_mfma_f32_16x16x4_f32 — f32 in, f32 accumulate, bit for bit an fmaf chain in k order, at the
//     f32 VECTOR rate (MI355X_MICROARCH.md): no arithmetic is saved, the pass leaves the vector issue port (7 600 of 19 200
//     wave-instructions per transform) and the register file (live state ~155 -> ~50 registers) — takes a batch of 16 butterflies
//     per wave: lane l = (kg = l >> 4, n = l & 15) supplies the pair sums / differences k = 4 s + kg of butterfly n in k-step s
//     and receives outputs q = 4 kg + r, r = 0 .. 3 of the same butterfly.  16 matrix instructions per batch, 33 batches;
//   * waves 8 - 15 (two per SIMD) do nothing else: they own the power sums.  Waves 0 - 7 (two per SIMD) own pass 0 — the loads of
//     the spectrum and the code spectrum, x conj(code), the radix-33 butterfly — and hold no power sums.  While waves 8 - 15 run
//     the radix-31 pass of transform m on the matrix pipe, waves 0 - 7 load and start transform m + 1 on the vector and
//     texture-address units.  All 16 waves share the radix-16 middle pass (1023 butterflies on 1024 lanes).
//
// Barriers per transform (both roles execute the same four; `image` = the LDS buffer):
//     [pass-0 waves: loads, products, radix-33 first half (registers only)]      [matrix waves: gather + radix-31 of transform m - 1]
//   B1  the image is free (every radix-31 gather of transform m - 1 has been read)
//     [pass-0 waves: radix-33 second half, scatter]
//   B2  pass-0 image complete
//     [all waves: gather 16, B3, radix 16, scatter]
//   B4  the image holds the radix-31 inputs of transform m
// Also measured and left out: the middle pass on the matrix waves alone with their own LDS-word barrier (two butterflies per lane; the
// pass-0 waves then go from B2 straight to the next loads, as in acq_comp_ws.h) — correct, no scratch, and no faster (303 against
// 299-304 us in the library, 311-321 against 309 in the lab): the matrix waves become the whole critical path.
// Earlier: the middle pass on the matrix waves alone with a private LDS-counter barrier, or on the
// pass-0 waves alone (two butterflies per lane either way: 64 more registers, spills inside the loops: 470 - 545 us against 336), the
// left-over batch rotated over the matrix waves or run in two low-register halves (spills: 386 - 410 us), wave priorities (no change).
// What bounds the kernel now: during the radix-31 phase the matrix pipe is ~77 % busy (33 x 16 instructions x 32 cycles / 4 SIMDs =
// 4 224 cycles per transform against a phase of ~6 000), and the single image serialises that phase with pass 0's scatter and the
// radix-16 pass.
// The matrix-pipe role is NOT what north_star foresaw ("MFMA is not used: no dense contraction here"): for this size there is
// one, and it is half of the kernel's arithmetic.  Measured: DESIGN.md §4.2 / §5.
#pragma once
#include "acq_device.h"
#include "ws31_core.h"

namespace gm {

// STAMPS (diagnostic, instantiated by tools/corr_lab only): lane 0 of waves 0 and W0 of workgroup 0 writes the shader clock at the
// phase boundaries of every transform into g_ws31_stamps[m][role][8]
__device__ long long* g_ws31_stamps = nullptr;
template <bool STAMPS> __device__ __forceinline__ void ws31_stamp(long long* base, int m, int role, int phase) {
    if constexpr (STAMPS) {
        if (base) {
            unsigned long long t;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            base[(size_t(m) * 3 + role) * 8 + phase] = (long long)t;
        }
    }
}

template <class PLX, bool REF_MUL, bool STAMPS = false>
__global__ __launch_bounds__(1024, 1) void acq_corr_ws31_kernel(
    const cf* __restrict__ spectra, const cf* __restrict__ code_fft,
    float* __restrict__ mmax, uint32_t* __restrict__ margmax, float* __restrict__ msum,
    const uint32_t* __restrict__ worker_list, int n_workers, int n_bins, int n_int, int map_mode,
    int split_from, int split_k, int split_items, float* __restrict__ split_scratch, uint32_t* __restrict__ split_counter,
    int strict_sum) {
    using PL = typename Ws31PlanOf<PLX::N>::type;
    static_assert(PLX::NP == 3 && PLX::R[0] == 33 && PLX::R[1] == 16 && PLX::R[2] == 31, "the stored order is the [33, 16, 31] prime-factor plan's");
    constexpr int T = PL::T, NB0 = PL::NB(0), W0 = 8, NWM = 8;
    using MF = Mfma31<PL, W0, NWM>;
    static_assert(PL::IT0 == 1 && PL::IT(1) == 1 && NB0 <= 64 * W0, "pass 0 on waves 0 .. W0 - 1, one butterfly per lane and pass");
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

    __shared__ cf lds[PL::LDS_ELEMS + PL::NB(PL::NP - 1)];       // one row behind the image: Mfma31's discarded partner read of row 0
    const int tid = threadIdx.x, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(spectra + size_t(d) * n_int * PL::N, unsigned(n_int) * PL::N * 8u);
    const __amdgpu_buffer_rsrc_t crs = make_rsrc(code_fft + size_t(p) * PL::N, PL::N * 8u);
    const int m_per = n_int / parts, m_begin = part * m_per, m_end = m_begin + m_per;

    __shared__ float lacc[MF::RL * 64];                            // the left-over batch's eight power sums per lane of wave W0's map, [slot][lane]
    if (wave == W0) {
#pragma unroll
        for (int r = 0; r < MF::RL; ++r) lacc[r * 64 + (tid & 63)] = 0.0f;
    }
    constexpr int AIT = MF::ITL, ARL = MF::RL, ITF = MF::ITF;       // the matrix waves' power slots (zero and untouched on the pass-0 waves)
    float acc[AIT][ARL];
#pragma unroll
    for (int it = 0; it < AIT; ++it)
#pragma unroll
        for (int r = 0; r < ARL; ++r) acc[it][r] = 0.0f;

    long long* stb = nullptr;
    if constexpr (STAMPS) stb = (blockIdx.x == 0 && (tid & 63) == 0 && (wave == 0 || wave == W0)) ? g_ws31_stamps : nullptr;
    const typename MF::Consts mconst = MF::consts(tid);
    const typename MF::Bases mb = MF::bases(lds, tid);
    // acc += norm_sqr() (:190-192): `acc + p` with p complete, so that the planes of a cut item (each 0 + p) merge to the same words
    auto out = [&](int it, int r, cf v) {
        if constexpr (REF_MUL) acc[it][r] = acc[it][r] + (v.x * v.x + v.y * v.y);
        else acc[it][r] = acc[it][r] + __builtin_fmaf(v.y, v.y, v.x * v.x);
        // (pinned here: left alone, hipcc moves the sums of ALL batches behind the last batch's products — the four products' 16
        // result registers per batch stay live until then, and the power sums go to scratch memory)
        asm volatile("" : "+v"(acc[it][r]));
    };
    // the same for the left-over batch (run by wave WX: a lane reads and writes only its own words, in program order)
    auto out_lds = [&](int, int r, cf v) {
        float* a = &lacc[r * 64 + (tid & 63)];
        if constexpr (REF_MUL) *a = *a + (v.x * v.x + v.y * v.y);
        else *a = *a + __builtin_fmaf(v.y, v.y, v.x * v.x);
    };
    constexpr int WX = 0;                                         // the pass-0 wave that runs the left-over batch
    __shared__ float xcs[8 * 64];                                 // ... and its lanes' matrix entries (Mfma31::Consts of wave W0's map)
    const int tid_x = 64 * W0 + (tid & 63);                      // ... as lane (tid & 63) of wave W0's map
    if (wave < W0) {
        // ---------------------------------------------------------------- pass-0 role (vector + texture-address units)
        // its lane constants wait in LDS and its gather bases are recomputed per use (lane number through an opaque move): kept in
        // registers across the loop they were ten more live values on every pass-0 wave, and hipcc reloaded two of them from scratch
        // memory behind B1
        if (wave == WX) {
            const typename MF::Consts xc0 = MF::consts(tid_x);
#pragma unroll
            for (int i = 0; i < 4; ++i) { xcs[i * 64 + (tid & 63)] = xc0.c[i]; xcs[(4 + i) * 64 + (tid & 63)] = xc0.s[i]; }
        }
        auto left_over = [&]() {
            int tx = tid_x;
            asm volatile("" : "+v"(tx));
            typename MF::Consts xconst;
#pragma unroll
            for (int i = 0; i < 4; ++i) { xconst.c[i] = xcs[i * 64 + (tx & 63)]; xconst.s[i] = xcs[(4 + i) * 64 + (tx & 63)]; }
            MF::left_over_batch(MF::bases(lds, tx), tx, xconst, out_lds);
        };
        // Both arrays are stored in row PAIRS, in the order the radix-33 butterfly consumes its inputs (PairRows, acq_corr_plans.h):
        // input r is stored row s = row(r), half s & 1 of the 16-byte pair s >> 1 of this lane; the two inputs of a pair are asked for
        // one after the other, and the second request of the same pair is the same load expression (one load instruction).
        using PR = PairRows<PLX>;
        static_assert(PR::FORCE && PairLayout<PLX>::PAIRED && PairLayout<PLX>::NB0 == NB0, "the [33, 16, 31] plan's paired, consumption-ordered rows");
        const int v16 = tid < NB0 ? tid * 16 : 0x7ffffff0, v8 = tid < NB0 ? tid * 8 : 0x7ffffff0;   // lanes 496 .. 511: out of range, no request
        auto pick = [](u32x4 q, int half) { return half ? cf_make(__uint_as_float(q.z), __uint_as_float(q.w)) : cf_make(__uint_as_float(q.x), __uint_as_float(q.y)); };
        for (int m = m_begin; m < m_end; ++m) {
            u32x4 qx, qc;                                             // the pair in hand (its second half is the next input asked for)
            auto in = [&](int, int r) {
                const int s_ = PR::row(r);                            // (a compile-time constant after unrolling: 0, 1, 2, ... in call order)
                cf a, g;
                if (s_ < 32) {
                    if ((s_ & 1) == 0) {                              // (loaded as ONE value used twice: two expressions would be narrowed to 8-byte loads)
                        qx = __builtin_amdgcn_raw_buffer_load_b128(xrs, v16, (m * PL::N + (s_ >> 1) * 2 * NB0) * 8, 0);
                        qc = __builtin_amdgcn_raw_buffer_load_b128(crs, v16, ((s_ >> 1) * 2 * NB0) * 8, 0);
                    }
                    a = pick(qx, s_ & 1);
                    g = pick(qc, s_ & 1);
                } else {
                    a = buf_load_cf(xrs, v8, (m * PL::N + 32 * NB0) * 8);
                    g = buf_load_cf(crs, v8, 32 * NB0 * 8);
                }
                const cf c = cf_make(g.x, -g.y);
                // result_buf[i] *= conj(code[i])  (:184-186); REF_MUL: num-complex's own unfused form (gm_acq_cfg.reference_products)
                if constexpr (REF_MUL) return cf_make(a.x * c.x - a.y * c.y, a.x * c.y + a.y * c.x);
                else return cf_make(__builtin_fmaf(a.x, c.x, -(a.y * c.y)), __builtin_fmaf(a.x, c.y, a.y * c.x));
            };
            ws31_stamp<STAMPS>(stb, m, 0, 0);
            {
                cf v0[PL::IT0][PL::R0];
                Fft<PL, true, true>::pass0_stage1(v0, in, tid);
                // the left-over batch of transform m - 1 (its radix-31 inputs are in the image until B1), in this wave's wait for B1
                if (wave == WX && m > m_begin) left_over();
                ws31_stamp<STAMPS>(stb, m, 0, 1);
                __syncthreads();                                    // B1: the image is free
                ws31_stamp<STAMPS>(stb, m, 0, 2);
                // second half of the radix-33 Good-Thomas butterfly (fft_core.h Bfly<33>: A = 3, B = 11, KIND 2): three 11-point
                // DFTs over v[n2 * 3 + k1], output (k1, k2) is element (k1 EA + k2 EB) mod 33 of the butterfly -> lds[b * 33 + ..]
                if (tid < NB0) {
                    constexpr int EA = Bfly<33, true>::EA, EB = Bfly<33, true>::EB;
                    cf* dst = lds + tid * 33;
#pragma unroll
                    for (int k1 = 0; k1 < 3; ++k1) {
                        cf u[11];
#pragma unroll
                        for (int n2 = 0; n2 < 11; ++n2) u[n2] = v0[0][n2 * 3 + k1];
                        dft11_inv_stream(u, [&](int k2, cf val) { dst[(k1 * EA + k2 * EB) % 33] = val; });
                    }
                }
            }
            ws31_stamp<STAMPS>(stb, m, 0, 3);
            __syncthreads();                                        // B2: pass-0 image complete
            ws31_stamp<STAMPS>(stb, m, 0, 4);
            MiddlePasses<PL, true, 1, true>::run(lds, nullptr, tid);   // B3, B4 inside
            ws31_stamp<STAMPS>(stb, m, 0, 5);
        }
        if (wave == WX) left_over();                                // the last transform's
        __syncthreads();                                            // the left-over batch's sums are complete
    } else {
        // ---------------------------------------------------------------- matrix-pipe role (owns the power sums)
        for (int m = m_begin; m < m_end; ++m) {
            ws31_stamp<STAMPS>(stb, m, 1, 0);
            if (m > m_begin) MF::pass(mb, tid, mconst, out, [&](int it) { ws31_stamp<STAMPS>(stb, m, 2, it); });      // the radix-31 pass of transform m - 1, beside pass 0 of transform m
            ws31_stamp<STAMPS>(stb, m, 1, 1);
            __syncthreads();                                        // B1: every gather of transform m - 1 has been read
            ws31_stamp<STAMPS>(stb, m, 1, 2);
            __syncthreads();                                        // B2
            ws31_stamp<STAMPS>(stb, m, 1, 4);
            MiddlePasses<PL, true, 1, true>::run(lds, nullptr, tid);   // B3, B4 inside
            ws31_stamp<STAMPS>(stb, m, 1, 5);
        }
        MF::pass(mb, tid, mconst, out);                             // the last transform's
        __syncthreads();                                            // the left-over batch's sums are complete (wave WX)
        if constexpr (MF::EXTRA == 1) {
            if (wave == W0) {                                       // the left-over batch's sums join the register slots for the epilogue
#pragma unroll
                for (int r = 0; r < ARL; ++r) acc[ITF][r] = lacc[r * 64 + (tid & 63)];
            }
        }
    }

    auto last_active = [&](int it) { return MF::batch_active(tid, it); };
    auto out_index = [&](int it, int r) { return MF::index(tid, it, r); };
    auto slot_ok = [&](int r) { return MF::slot_ok(tid, r); };
    constexpr int WA = W0, TM = T - 64 * WA;                         // the waves that own power sums: WA .. 15
    constexpr int RL4 = ARL / 4;                                     // 16-byte groups of a lane's power values
    static_assert(ARL % 4 == 0, "power slots in 16-byte groups");
    const int mt = tid - 64 * WA;                                   // lane number among the waves that own power sums

    if (parts > 1) {      // a part of a cut item (one integration): its plane goes out (see acq_corr_kernel for the protocol and the store-data guard)
        constexpr int SLAB = AIT * RL4 * 4 * TM;                    // floats per power plane, register order of the matrix lanes
        const size_t item_plane0 = (size_t(xcd) * split_items + (slot - split_from)) * size_t(n_int);
        const __amdgpu_buffer_rsrc_t srs = make_rsrc(split_scratch + item_plane0 * SLAB, unsigned(n_int) * SLAB * 4u);
        if (wave >= WA) {
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
        if (wave >= WA) {       // the last arriver: the n_int planes added in integration order (sc1 loads)
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
    constexpr int NW = T / 64;
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

// which registered plans take this kernel, and the floats of one split power plane on it
template <class PL> struct Ws31 {
    static constexpr bool USE = !PL::HYBRID && PL::COPRIME && PL::NP == 3 && PL::R[0] == 33 && PL::R[1] == 16 && PL::R[2] == 31 && PL::N % 16 == 0 &&
                                (PL::N / 31) % 16 == 0;
};
template <class PL> constexpr int ws31_split_slab() {
    using W = typename Ws31PlanOf<PL::N>::type;
    using MF = Mfma31<W, 8, 8>;
    return MF::ITL * (MF::RL / 4) * 4 * 64 * 8;
}

}  // namespace gm
