// acq_comp_ws.h — comp_corr_kernel's work (acq_composite.hip: the fused inverse of the composite sizes N = Q * Nb) for the base
// size Nb = 16000 (N = 32000: a Galileo E1 code period at 8 Msps, BASELINE configs[3]) as a WAVE-SPECIALISED kernel.
// Same arguments, same item walk, same results contract as comp_corr_kernel.
//
// Why.  One sub-transform's image is 128 KB of a CU's 160 KB of LDS: one workgroup per CU, 16 waves in lockstep.  Every input of a
// sub-transform is formed from 2Q loads (Q spectrum blocks x their combined code tables): 512 KB per sub-transform at Q = 2, which a CU's
// load path streams from L2 in no less than ~4 us (tools/ubench/l2_read.hip: 125 GB/s per CU, cache-resident, 16-byte loads) — of the
// 12.8 us the lockstep kernel spends per sub-transform, with nothing else running on the CU meanwhile (66 % of the wave-cycles parked,
// vector issue 0.39).  A second workgroup does not fit, a second image does not fit.  What fits — as in acq_corr_ws31.h — is the next
// sub-transform's pass 0 on OTHER waves, its values held in registers until the image is free:
//
//   * the hybrid plan [20, 25, 32] (fft_core.h HybridPlan<16000, 1024, 5, 25, 4, 32>) runs its last pass on waves 0 - 7 (500 radix-32
//     butterflies, four constant-twiddle groups of two waves).  Those waves own the power sums (32 per lane) and also run the radix-25
//     middle pass — its ten wave-slots on eight waves, waves 0 and 1 take two — separated from the last pass by a barrier of their own
//     (an LDS word, comp_ws_wave_group_barrier: the hardware barrier would stop the other waves too);
//   * waves 8 - 15 do pass 0 and nothing else: the 2Q loads per row pair, the Q products, the radix-20 Good-Thomas butterfly's first
//     half — 800 butterflies on 512 lanes: two per lane on lanes 0 - 287, whose second butterfly parks 15 of its 20 intermediate values
//     in the 35 KB of LDS the image leaves free; the first loads of s + 1 are requested in front of B1 of s.  From B2 of sub-transform s they go straight to the loads of s + 1, which are in flight
//     during the middle AND the last pass of s.  They hold no power sums and never run a radix-25 or radix-32 butterfly.
//
// Workgroup barriers per sub-transform (every wave executes the same two):
//       [waves 0-7: middle pass of s - 1, their own barrier, last pass of s - 1]      [waves 8-15: loads + first halves of pass 0 of s]
//   B1  the image is free
//       [waves 8-15: second halves of the radix-20 butterflies, scatter]
//   B2  pass-0 image of s complete
// Both arrays are stored in row PAIRS in the order the radix-20 butterfly consumes its inputs (PairRows<CorrPlan16000>): every 16-byte
// load feeds the next two inputs and nothing waits in registers for its partner.
//
// Measured and left out: the 288 second butterflies dealt evenly to the eight pass-0 waves (36 lanes each) instead of to the first
// 4.5 — every wave then runs both code paths and B1 comes no earlier: 266 against 253 us.
// Non-temporal hints (`nt`) on the code-table loads or on the spectrum loads, to keep the other array in L2: 333 / 302 us.
// Measured (configs[3] Galileo geometry, 36 codes x 41 bins x 2 periods; tools/corr_lab/comp_ws_stamps.hip): 307 us for
// the lockstep kernel on the plain [25, 20, 32] plan -> 255 us (rocprofv3 in bench.py: 243).  The pass-0 waves now stream 512 KB per 5.3 us and CU (~100 GB/s of the
// 125 the load path gives): the kernel is bound by its loads.  What it took beyond the roles — every item cost a multiple of its size
// in a first version that ran at 531 us:
//   * no scratch memory inside the loops.  A spilled register is reloaded through the same load path as the pass-0 waves' requests and
//     waits behind them (1 000 - 2 000 cycles per reload site): the roles are two loops (one loop with role branches: 176 spills); the
//     last pass is a function of its own (called inline on the __shared__ array: 220 spills); the wave number is a scalar; lane numbers
//     go through an opaque move per iteration so that derived addresses are recomputed instead of kept; the first-half values are pinned
//     in front of B1 (left alone hipcc requests all 20Q loads at once, sinks every product behind the barrier and spills the loads);
//   * the per-n1 fold of the power sums runs on values, one index per lane, reduced over the wave into scalar state (slot by slot with
//     indices: 36 000 cycles per fold, as much as two sub-transforms).
#pragma once
#include "acq_device.h"

namespace gm {

template <class CP> struct CompWs { static constexpr bool USE = false; };
#ifndef GM_COMP_NO_WS          // (A/B switch: the generic comp_corr_kernel on the hybrid plan)
template <> struct CompWs<HybridPlan<16000, 1024, 5, 25, 4, 32>> { static constexpr bool USE = true; };
#endif

// STAMPS (diagnostic, instantiated by tools/corr_lab only): lane 0 of waves 0, 7 and 8 of four workgroups writes the shader clock at the
// phase boundaries of every sub-transform into g_comp_ws_stamps[workgroup][s][wave slot][8]
__device__ long long* g_comp_ws_stamps = nullptr;
__device__ long long* g_comp_ws_wg = nullptr;          // (diagnostic) [workgroup][2]: clock at entry and exit of every workgroup that has an item
template <bool STAMPS> __device__ __forceinline__ void comp_ws_stamp(long long* base, int s, int slot, int phase) {
    if constexpr (STAMPS) {
        if (base) {
            unsigned long long t;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            base[(size_t(s) * 3 + slot) * 8 + phase] = (long long)t;
        }
    }
}

// The first half of one lane's pass-0 butterfly: inputs = sum over k1 of X[m][k1][k] * comb[n1][k1][k] (the Q-point inverse DFT row, the
// twiddle and conj(code), :184-186, are inside the table), then the radix-20 Good-Thomas butterfly's first half (fft_core.h Bfly<20>::s1:
// five 4-point DFTs over inputs (5 n1 + 4 N2) mod 20), group by group: the four inputs of a group are two stored row pairs = 4Q 16-byte
// loads requested together.  DEPTH groups are in flight: group g + DEPTH - 1 is requested before group g is consumed, and the groups stay
// apart in the schedule (hoisted to the top, the 20Q loads of a butterfly cost 270 spilled registers).  xoff: element offset of spectrum
// block (m, k1 = 0); emit(i, value) receives value i of the butterfly's R0 intermediate values.
template <class PL, uint32_t Q, int ROWS> struct CompWsGroup {       // a load unit: ROWS stored rows (an even number) = ROWS / 2 row pairs x Q blocks of both arrays
    u32x4 x[ROWS / 2][Q], c[ROWS / 2][Q];
    __device__ __forceinline__ void request(__amdgpu_buffer_rsrc_t xrs, __amdgpu_buffer_rsrc_t crs, int v16, int xoff, int u) {
#pragma unroll
        for (int h = 0; h < ROWS / 2; ++h)
#pragma unroll
            for (uint32_t k1 = 0; k1 < Q; ++k1) {
                const int st = u * ROWS + 2 * h;              // stored rows st, st + 1
                x[h][k1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, v16, (xoff + int(k1) * PL::N + st * PL::NB(0)) * 8, 0);
                c[h][k1] = __builtin_amdgcn_raw_buffer_load_b128(crs, v16, (int(k1) * PL::N + st * PL::NB(0)) * 8, 0);
            }
    }
    template <int OFF, int A> __device__ __forceinline__ void inputs(cf (&t)[A]) const {        // t[OFF .. OFF + ROWS - 1]
        auto lo = [](u32x4 v) { return cf_make(__uint_as_float(v.x), __uint_as_float(v.y)); };
        auto hi = [](u32x4 v) { return cf_make(__uint_as_float(v.z), __uint_as_float(v.w)); };
#pragma unroll
        for (int h = 0; h < ROWS / 2; ++h) {
            cf s0 = cf_mul(lo(x[h][0]), lo(c[h][0])), s1 = cf_mul(hi(x[h][0]), hi(c[h][0]));
#pragma unroll
            for (uint32_t k1 = 1; k1 < Q; ++k1) {
                s0 = cf_add(s0, cf_mul(lo(x[h][k1]), lo(c[h][k1])));
                s1 = cf_add(s1, cf_mul(hi(x[h][k1]), hi(c[h][k1])));
            }
            t[OFF + 2 * h] = s0;
            t[OFF + 2 * h + 1] = s1;
        }
    }
};
template <class PL, uint32_t Q> struct CompWsStream {
    using B0 = Bfly<PL::R0, true>;
    static_assert(B0::KIND == 2 && B0::A % 2 == 0 && B0::B >= 2 && (B0::A == 4 || B0::A == 2), "first radix: Good-Thomas, groups of two or four rows");
    // Load units in flight.  A unit of ROWS rows is 2 Q ROWS / 2 16-byte loads = 4 Q ROWS registers: whole groups (ROWS = A = 4), two in
    // flight at Q = 2 (64 registers), one at Q = 3, 4; above that single row pairs (ROWS = 2: 8 Q registers), one in flight.
    static constexpr int ROWS = Q <= 4 ? B0::A : 2, UPG = B0::A / ROWS, NU = B0::B * UPG;      // units per group, units per butterfly
    static constexpr int DEPTH = Q <= 2 ? 2 : 1;
    CompWsGroup<PL, Q, ROWS> buf[DEPTH];
    // request the first DEPTH units of a butterfly (they may then be in flight across a barrier: run() consumes them first)
    __device__ __forceinline__ void start(__amdgpu_buffer_rsrc_t xrs, __amdgpu_buffer_rsrc_t crs, int v16, int xoff) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) buf[u].request(xrs, crs, v16, xoff, u);
    }
    // unit J of group g: its inputs into t[J * ROWS ..], then (behind the last use of the unit in hand) the request of unit + DEPTH
    template <int J> __device__ __forceinline__ void unit(cf (&t)[B0::A], int g, __amdgpu_buffer_rsrc_t xrs, __amdgpu_buffer_rsrc_t crs, int v16, int xoff) {
        const int u = g * UPG + J;
        buf[u % DEPTH].template inputs<J * ROWS>(t);
        if (u + DEPTH < NU) buf[u % DEPTH].request(xrs, crs, v16, xoff, u + DEPTH);
        if (UPG > 1) __builtin_amdgcn_sched_barrier(0);
    }
    template <class Emit>
    __device__ __forceinline__ void run(Emit&& emit, __amdgpu_buffer_rsrc_t xrs, __amdgpu_buffer_rsrc_t crs, int v16, int xoff) {
#pragma unroll
        for (int g = 0; g < B0::B; ++g) {
            __builtin_amdgcn_sched_barrier(0);
            cf t[B0::A];
            unit<0>(t, g, xrs, crs, v16, xoff);
            if constexpr (UPG > 1) unit<1>(t, g, xrs, crs, v16, xoff);
            Dft<B0::A, true>::run(t);
#pragma unroll
            for (int k1 = 0; k1 < B0::A; ++k1) {
                // (the values are pinned HERE: left alone, hipcc requests all 20Q loads at once and moves every product and the whole
                // butterfly behind the barrier that follows, 128 registers of loads in flight across it and 9 spilled 16-byte loads)
                asm volatile("" : "+v"(t[k1].x), "+v"(t[k1].y));
                emit(g * B0::A + k1, t[k1]);                  // (index: a compile-time constant after unrolling)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

// The last pass as a function of its own, the image handed over as a pointer: called on the __shared__ array itself from the kernel body,
// the same two calls cost 220 spilled registers (measured on a cut-down kernel: 0 spills through this function, 223 inline)
template <class PL, class Out> __device__ __forceinline__ void comp_ws_last_pass(cf* lds, int tid, Out&& out) {
    cf vl[PL::ITL][PL::RL];
    Fft<PL, true>::last_stage1(vl, lds, nullptr, tid);
    Fft<PL, true>::last_stage2(vl, out, tid);
}

// the radix-25 middle pass (in place per lane: no barrier inside) without MiddlePasses' closing workgroup barrier
template <class PL> __device__ __forceinline__ void comp_ws_middle_pass(cf* lds, int tid) {
    cf v[PL::IT(1)][PL::R[1]];
    Fft<PL, true>::template mid_stage1<1>(v, lds, nullptr, tid);
    __builtin_amdgcn_sched_barrier(0);
    Fft<PL, true>::template mid_stage2<1>(v, lds, tid);
}

// A barrier among SOME waves of the workgroup (the hardware barrier counts all of them): every participating wave adds one to an LDS
// word and waits until the word has reached `target` = (barriers so far) x (participating waves).  Release / acquire at workgroup scope:
// the wave's LDS writes have landed before its arrival counts, its LDS reads start after the last arrival.  Every participating wave
// passes here the same number of times (the sub-transform count of the item), so the word reaches every target.
__device__ __forceinline__ void comp_ws_wave_group_barrier(unsigned* word, unsigned target) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) {
        __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// (the kernel's body is a function that receives the image as a pointer, for the same reason)
template <class CP, uint32_t Q, bool STAMPS, bool PLANES>
__device__ __forceinline__ void comp_corr_ws_body(
    cf* lds, const cf* __restrict__ spectra, const cf* __restrict__ code_fft,
    float* __restrict__ mmax, uint32_t* __restrict__ margmax, float* __restrict__ msum,
    const uint32_t* __restrict__ worker_list, int n_workers, int n_bins, int n_int, int cb, int rows_max, float* __restrict__ planes) {
    using PL = CP;
    using F = Fft<PL, true>;
    using PR = PairRows<PL>;
    constexpr int T = 1024, Nb = PL::N, R0 = PL::R0, RL = PL::RL, NB0 = PL::NB(0);
    // roles by wave: 0 .. WB - 1 middle pass + last pass + power sums; the others (WA waves) pass 0 only: two butterflies per lane on the
    // first NA2 lanes, one on the others.  The middle pass's MW wave-slots (k1 groups x GW1) go to waves 0 .. WB - 1, the first MW - WB
    // of them take two.
    constexpr int WB = 8, WA = T / 64 - WB, NA = 64 * WA, NA2 = NB0 - NA, MW = PL::A1 * PL::GW1;
    static_assert(PL::HYBRID && PL::T == T && PairLayout<PL>::PAIRED && (R0 & 1) == 0, "a hybrid plan on 1024 lanes, rows in pairs");
    static_assert(PL::B1 * PL::GW2 <= WB && MW >= WB && MW <= 2 * WB && NA2 >= 0 && NA2 <= NA, "last and middle pass on waves 0 .. 7, pass 0 on the others");
    static_assert(PR::nat(0) == 0 && PR::nat(1) == Bfly<R0, true>::B && PR::nat(Bfly<R0, true>::A) == Bfly<R0, true>::A % R0, "rows stored in the butterfly's consumption order");
    constexpr uint32_t N = Q * uint32_t(Nb);

    // ---- the item of this workgroup: comp_corr_kernel's walk (acq_composite.hip)
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

    // the wave number as a scalar: role, k1 / j1 groups of the hybrid passes and every other per-wave decision then branch on scalar
    // registers (as a vector value it was the one hipcc spilled, reloaded from scratch memory at the head of every phase)
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
    const int tid = wave * 64 + (int(threadIdx.x) & 63);
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(spectra + size_t(d) * n_int * N, unsigned(n_int) * N * 8u);
    __shared__ unsigned group_word;                           // comp_ws_wave_group_barrier of waves 0 .. WB - 1
    if (tid == 0) group_word = 0;
    const int S = int(Q) * n_int;                             // sub-transforms of this item: s = n1 * n_int + m
    long long* stb = nullptr;                                 // (diagnostic) waves 0, 7, WB of workgroups 0, 504, 1008, 1512: [4][S][3][8]
    const int wslot = wave == 0 ? 0 : (wave == WB ? 2 : 1);
    if constexpr (STAMPS)
        stb = (blockIdx.x % 504 == 0 && blockIdx.x < 2016 && (tid & 63) == 0 && (wave == 0 || wave == WB - 1 || wave == WB) && g_comp_ws_stamps)
                  ? g_comp_ws_stamps + size_t(blockIdx.x / 504) * S * 24 : nullptr;
    // sub-transform s reads the spectrum blocks of integration m = s mod n_int and the combined tables of n1 = s / n_int
    auto crs_of = [&](int s) { return make_rsrc(code_fft + (size_t(p) * Q + s / n_int) * N, N * 8u); };
    auto xoff_of = [&](int s) { return (s % n_int) * int(Q) * Nb; };
    // A lane with two butterflies keeps the first NST intermediate values of its second one in LDS until the image is free (the 35 KB
    // the image leaves: a lane reads and writes only its own words): with all 40 values of both in registers beside the loads in
    // flight, hipcc spilled 11 registers and the scatter phase began with their reloads from scratch memory
    constexpr int NST = 15;
    __shared__ cf stage[NST * (NA2 > 0 ? NA2 : 1)];
    __syncthreads();                                          // (group_word is zero)

    // the WAVE's running first strict maximum / sum, wave-uniform (scalar registers)
    float bv = 0.0f, sum = 0.0f;
    uint32_t bi = 0xffffffffu;

    // Two loops, one per role; every wave executes the same two workgroup barriers per sub-transform:
    //   B1(s)  the image is free (every last-pass read of s - 1 is done)      -> the pass-0 waves write their butterflies' outputs
    //   B2(s)  pass-0 image of s complete                                    -> waves 0 .. 7 run the middle pass, meet at their wave-group
    //                                                                           barrier, then run the last pass of s
    // A pass-0 wave goes from B2(s) straight to the loads of s + 1: they are in flight during the middle AND the last pass of s.
    if (wave >= WB) {
        // ---------------------------------------------------------------- pass 0 only (vector memory + a radix-20 butterfly or two)
        const int a0 = tid - 64 * WB;
        CompWsStream<PL, Q> st;
        st.start(xrs, crs_of(0), a0 * 16, xoff_of(0));
        for (int s = 0; s < S; ++s) {
            // (the lane number goes through an opaque move in every iteration: the addresses derived from it — load offsets, the staging
            // slot, the scatter bases — are then worked out where they are used, a few instructions each; hoisted out of the loop they were
            // four more registers live across it, which hipcc spilled and reloaded from scratch memory behind each barrier)
            int a = a0;
            asm volatile("" : "+v"(a));
            const bool two = a < NA2;
            cf va[1][R0], vb[1][R0];
            comp_ws_stamp<STAMPS>(stb, s, wslot, 0);
            const __amdgpu_buffer_rsrc_t crs = crs_of(s);
            const int xoff = xoff_of(s);
            st.run([&](int i, cf val) { va[0][i] = val; }, xrs, crs, a * 16, xoff);
            if (two) {
                st.start(xrs, crs, (NA + a) * 16, xoff);
                st.run([&](int i, cf val) {
                    if (i < NST) stage[i * NA2 + a] = val;
                    else vb[0][i] = val;
                }, xrs, crs, (NA + a) * 16, xoff);
            }
            // the first groups of the NEXT sub-transform are requested here: in flight while this wave waits for B1 and scatters
            if (s + 1 < S) st.start(xrs, crs_of(s + 1), a * 16, xoff_of(s + 1));
            comp_ws_stamp<STAMPS>(stb, s, wslot, 1);
            __syncthreads();                                  // B1
            comp_ws_stamp<STAMPS>(stb, s, wslot, 2);
            asm volatile("" : "+v"(a));
            F::pass0_stage2(va, lds, a);
            if (two) {
#pragma unroll
                for (int i = 0; i < NST; ++i) vb[0][i] = stage[i * NA2 + a];
                F::pass0_stage2(vb, lds, NA + a);
            }
            comp_ws_stamp<STAMPS>(stb, s, wslot, 3);
            __syncthreads();                                  // B2
        }
    } else {
        // ---------------------------------------------------------------- middle pass + last pass; owns the power sums
        float acc[RL];                                        // |y|^2 summed over the integrations of the current n1
#pragma unroll
        for (int r = 0; r < RL; ++r) acc[r] = 0.0f;
        int s = 0;
        for (uint32_t n1 = 0; n1 < Q; ++n1) {
            for (int m = 0; m < n_int; ++m, ++s) {
                comp_ws_stamp<STAMPS>(stb, s, wslot, 0);
                __syncthreads();                              // B1
                comp_ws_stamp<STAMPS>(stb, s, wslot, 2);
                __syncthreads();                              // B2
                comp_ws_stamp<STAMPS>(stb, s, wslot, 4);
                comp_ws_middle_pass<PL>(lds, tid);
                if (wave < MW - WB) comp_ws_middle_pass<PL>(lds, tid + 64 * WB);      // (wave-slots WB .. MW - 1)
                comp_ws_stamp<STAMPS>(stb, s, wslot, 5);
                comp_ws_wave_group_barrier(&group_word, unsigned(s + 1) * WB);
                comp_ws_stamp<STAMPS>(stb, s, wslot, 6);
                // += norm_sqr() (:190-192)
                comp_ws_last_pass<PL>(lds, tid, [&](int, int r, cf v) { acc[r] = acc[r] + (v.x * v.x + v.y * v.y); });
                comp_ws_stamp<STAMPS>(stb, s, wslot, 7);
            }
            // Sub-transform n1's outputs are y[Q n2 + n1]: fold the finished sums into the wave's running first strict maximum / sum.  The
            // scan runs on VALUES.  A lane's slots r = 0 .. RL - 1 are the elements (e0 + STEP r) mod Nb, STEP = Nb / RL (HybridPlan::
            // out_index: one wrap over the 32 slots), so of several slots that hold the lane's maximum the lowest element is the first
            // slot at or behind the wrap if there is one, else the first slot of all: one index per lane, no search.  (Slot by slot with
            // indices, the 32 index computations and the registers hipcc spilled around them took 36 000 cycles per n1.)
            {
                constexpr int STEP = Nb / RL;
                static_assert(STEP == PL::A * PL::B1 && STEP * RL == Nb, "slot q2 of a lane is element e0 + A B1 q2 (mod Nb): HybridPlan::out_index");
                const int e0 = PL::out_index(tid, 0), rw = (Nb - e0 + STEP - 1) / STEP;      // slots rw .. RL - 1 have wrapped
                if constexpr (PLANES) {                       // gm_acq_cfg.strict_sum_order: the finished sums at their natural indices
                    if (PL::last_active(tid)) {
                        float* dst = planes + (size_t(p) * n_bins + d) * N + n1;
#pragma unroll
                        for (int r = 0; r < RL; ++r) {
                            int e = e0 + STEP * r;
                            e = e >= Nb ? e - Nb : e;
                            dst[size_t(Q) * e] = acc[r];
                        }
                    }
                }
                float m1 = 0.0f, ps = 0.0f;                  // (from 0.0 like the reference's scan, :195-202: a NaN never becomes the maximum)
#pragma unroll
                for (int r = 0; r < RL; ++r) m1 = acc[r] > m1 ? acc[r] : m1;
                int r_any = -1, r_wrapped = -1;
#pragma unroll
                for (int r = RL - 1; r >= 0; --r) {
                    const bool holds = acc[r] == m1;
                    r_any = holds ? r : r_any;
                    r_wrapped = (holds && r >= rw) ? r : r_wrapped;
                }
#pragma unroll
                for (int r = 0; r < RL; ++r) ps += acc[r];
                // this lane's proposal, reduced over the wave at once and merged into the wave's state (the same rule at every level:
                // larger value, equal values -> lower index)
                float pv = 0.0f;
                uint32_t pi = 0xffffffffu;
                if (PL::last_active(tid) && r_any >= 0) {
                    const int rb = r_wrapped >= 0 ? r_wrapped : r_any;
                    int e = e0 + STEP * rb;
                    e = e >= Nb ? e - Nb : e;
                    pv = m1;
                    pi = Q * uint32_t(e) + n1;
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const float ov = __shfl_xor(pv, off, 64);
                    const uint32_t oi = uint32_t(__shfl_xor(int(pi), off, 64));
                    ps += __shfl_xor(ps, off, 64);
                    take_better(pv, pi, ov, oi);
                }
                take_better(bv, bi, __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pv))), uint32_t(__builtin_amdgcn_readfirstlane(int(pi))));
                sum += __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ps)));
            }
#pragma unroll
            for (int r = 0; r < RL; ++r) acc[r] = 0.0f;
        }
    }

    __syncthreads();   // everyone is done with the image: reuse it as scratch
    float* sv = reinterpret_cast<float*>(lds);
    uint32_t* si = reinterpret_cast<uint32_t*>(lds) + 64;
    float* ss = reinterpret_cast<float*>(lds) + 128;
    if ((tid & 63) == 0) { sv[wave] = bv; si[wave] = bi; ss[wave] = sum; }
    __syncthreads();
    if (tid == 0) {
        float fv = sv[0], fs = ss[0];
        uint32_t fi = si[0];
        for (int w = 1; w < WB; ++w) { take_better(fv, fi, sv[w], si[w]); fs += ss[w]; }
        if (fi == 0xffffffffu) fi = 0;   // all-NaN / all-zero plane: the reference keeps (0.0, 0)
        const size_t o = size_t(p) * n_bins + d;
        mmax[o] = fv; margmax[o] = fi; msum[o] = fs;
    }
}

// PLANES (gm_acq_cfg.strict_sum_order): an instantiation of its own that also stores the accumulated power planes (acq_composite.hip)
template <class CP, uint32_t Q, bool STAMPS = false, bool PLANES = false>
__global__ __launch_bounds__(1024, 1) void comp_corr_ws_kernel(
    const cf* __restrict__ spectra, const cf* __restrict__ code_fft,
    float* __restrict__ mmax, uint32_t* __restrict__ margmax, float* __restrict__ msum,
    const uint32_t* __restrict__ worker_list, int n_workers, int n_bins, int n_int, int cb, int rows_max, float* __restrict__ planes) {
    __shared__ cf image[CP::LDS_ELEMS];
    if constexpr (STAMPS) {
        if (g_comp_ws_wg && threadIdx.x == 0) { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); g_comp_ws_wg[2 * blockIdx.x] = (long long)t; }
    }
    comp_corr_ws_body<CP, Q, STAMPS, PLANES>(image, spectra, code_fft, mmax, margmax, msum, worker_list, n_workers, n_bins, n_int, cb, rows_max, planes);
    if constexpr (STAMPS) {
        if (g_comp_ws_wg && threadIdx.x == 0) { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); g_comp_ws_wg[2 * blockIdx.x + 1] = (long long)t; }
    }
}

}  // namespace gm
