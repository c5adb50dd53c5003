// acq_device.h — device-side building blocks shared by the acquisition kernels (acq_kernels.hip: fused in-LDS sizes,
// acq_composite.hip: sizes above one LDS buffer): the workgroup-wide in-LDS transform, buffer-descriptor loads, the paired
// spectrum layout and the (value, index) reduction rule of the reference's argmax scan (do_acquisition.rs:195-202).
#pragma once
#include "gm_internal.h"
#include "fft_plans.h"

namespace gm {

struct NoStamp { __device__ __forceinline__ void operator()(int) const {} };

// hybrid plans run their middle pass in place per wave group (fft_core.h): a one-wave group orders its own LDS reads before its
// writes by program order alone
template <class PL> constexpr bool mid_pass_needs_barrier() {
    if constexpr (PL::HYBRID) return PL::P1_NEEDS_BARRIER;
    else return true;
}
template <class PL, bool INV, int S, bool PFA = false> struct MiddlePasses {
    // st(k): optional diagnostic stamp hook, called only next to barriers (k = 4.. in program order)
    template <class St = NoStamp>
    static __device__ __forceinline__ void run(cf* lds, const cf* tw, int tid, St st = St()) {
        if constexpr (S <= PL::NP - 2) {
            cf v[PL::IT(S)][PL::R[S]];
            Fft<PL, INV, PFA>::template mid_stage1<S>(v, lds, tw, tid);
            st(4);
            if constexpr (mid_pass_needs_barrier<PL>()) __syncthreads();   // every lane has read its inputs: the image may be overwritten
            else __builtin_amdgcn_sched_barrier(0);   // no hardware barrier, but the two halves stay apart in the schedule (the
                                                      // butterfly's second half pulled over the first cost 65 spilled registers)
            st(5);
            Fft<PL, INV, PFA>::template mid_stage2<S>(v, lds, tid);
            st(6);
            __syncthreads();
            st(7);
            MiddlePasses<PL, INV, S + 1, PFA>::run(lds, tw, tid);
        }
    }
};

// One length-N transform by the whole workgroup: in(it, r) feeds pass 0, out(it, r, value) receives
// the natural-order outputs.  Safe to call back to back (the first barrier orders the scatter after
// the previous transform's last LDS reads and after the twiddle-table load).
// PFA (PL::COPRIME plans only): the prime-factor form across the passes — inputs in Pfa<PL>::in_slot order, outputs at
// Pfa<PL>::out_index, `tw` unused.
// `hook(k)` is called at fixed points of the transform (k = 0: behind pass 0's LDS writes, 1: behind the middle passes, 2: between
// the halves of the last butterfly) — acq_corr_kernel requests a part of the next transform's inputs from one of them.
template <class PL, bool INV, bool PFA = false, class In, class Out, class Hook = NoStamp>
__device__ __forceinline__ void lds_transform(In&& in, Out&& out, cf* lds, const cf* tw, int tid, Hook&& hook = Hook()) {
    {
        cf v0[PL::IT0][PL::R0];
        Fft<PL, INV, PFA>::pass0_stage1(v0, in, tid);
        __syncthreads();
        Fft<PL, INV, PFA>::pass0_stage2(v0, lds, tid);
    }
    hook(0);
    __syncthreads();
    MiddlePasses<PL, INV, 1, PFA>::run(lds, tw, tid);
    hook(1);
    cf vl[PL::ITL][PL::RL];
    Fft<PL, INV, PFA>::last_stage1(vl, lds, tw, tid);
    hook(2);
    Fft<PL, INV, PFA>::last_stage2(vl, out, tid);
}

template <class PL> __device__ __forceinline__ void load_twiddles(cf* tw_lds, const cf* tw_g, int tid) {
    for (int i = tid; i < PL::TW_TOTAL; i += PL::T) tw_lds[i] = tw_g[i];
}

// Buffer-descriptor loads: the 128-bit resource sits in SGPRs, the per-lane byte offset is ONE VGPR
// shared by every element of the butterfly, the per-element stride goes into the scalar offset.
// (Flat addressing made hipcc keep one 64-bit VGPR address per element live across the m loop.)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, int(bytes), 0x00020000);
}
__device__ __forceinline__ cf buf_load_cf(__amdgpu_buffer_rsrc_t rsrc, int voff_bytes, int soff_bytes) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff_bytes, soff_bytes, 0);
    return cf_make(__uint_as_float(v.x), __uint_as_float(v.y));
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Paired layout of the stage F -> stage C spectra and of the code spectra the correlation kernel reads.
// Lane b of pass 0 of the inverse transform consumes elements b + r*NB0, r = 0..R0-1.  With 8-byte elements stored in
// natural order that is R0 8-byte loads per lane, and pass 0 is bound by the texture-address path (an 8-byte-per-lane
// load moves half the bytes of a 16-byte one per issue slot: configs[1] 213 -> 190 us per launch with pairs).  So the
// elements of one lane are stored two by two: position of natural index k = b + r*NB0 is
//     r < 2*(R0/2):  ((r/2)*NB0 + b)*2 + (r & 1)        [R0/2][NB0][2]  -> one 16-byte load per pair, coalesced over b
//     r = R0-1, R0 odd:  2*(R0/2)*NB0 + b                 [NB0]           -> one 8-byte load
// Only acq_mix_fft_kernel (writer), pair_codes_kernel (writer) and acq_corr_kernel (reader) know this layout.
// Plans whose first radix is too large for both arrays' pass-0 elements to sit in registers at once (R0 = 31, 32, 33: the
// hoisted loads cost Plan16368 20 % at 167 VGPRs) keep the natural order and the element-by-element loads.
template <class PL> struct PairLayout {
    static constexpr int R0 = PL::R0, NB0 = PL::NB(0), NPAIR = R0 / 2;
    static constexpr bool PAIRED = R0 <= 25 || PairRows<PL>::FORCE;
    // (rows may be stored in another order than the natural one: PairRows, acq_corr_plans.h — the identity except for N = 16368)
    static __host__ __device__ __forceinline__ int pos(int k) {
        if constexpr (!PAIRED) return k;
        const int rn = k / NB0, b = k - rn * NB0, r = PairRows<PL>::row(rn);
        return r < 2 * NPAIR ? ((r >> 1) * NB0 + b) * 2 + (r & 1) : 2 * NPAIR * NB0 + b;
    }
    static __host__ __device__ __forceinline__ int unpos(int p) {      // the inverse of pos
        if constexpr (!PAIRED) return p;
        if (p >= 2 * NPAIR * NB0) return (p - 2 * NPAIR * NB0) + PairRows<PL>::nat(R0 - 1) * NB0;
        const int pair = p >> 1, rp = pair / NB0, b = pair - rp * NB0;
        return b + PairRows<PL>::nat(2 * rp + (p & 1)) * NB0;
    }
};

// The plan the correlation kernel runs its inverse transform on.  Default: the plan of the size itself; fft_plans.h names a
// HybridPlan for sizes whose factorisation allows one (N = 8000 = 125 * 64).  The forward transforms (stage F, the code
// spectra) always run the plain plan on natural-order input and produce natural-order output; what differs is the ORDER in
// which that output is stored (CorrLayout), chosen by the reader.
template <class PL> struct CorrPlanOf { using type = PL; };

// keyed on the CORRELATION plan CP:
template <class CP> struct CorrMode {
    static constexpr bool HYBRID = CP::HYBRID;                       // HybridPlan: prime-factor across A x B, constant twiddles inside
    static constexpr bool PFA = CP::COPRIME && !CP::HYBRID;          // generic plan with pairwise coprime radices: no twiddles at all
    static constexpr bool PERMUTED = HYBRID || PFA;                  // stored spectra are not in natural element order
};
// CODE_PAIRED: the kernel takes BOTH arrays as 16-byte pairs (PairLoad); RELAYOUT: the stored spectra / code spectra are not in
// natural order (paired, permuted or both), i.e. the code spectra go through pair_codes_kernel once per handle
template <class CP> struct CorrLayout {
    static constexpr bool CODE_PAIRED = PairLayout<CP>::PAIRED;
    static constexpr bool RELAYOUT = PairLayout<CP>::PAIRED || CorrMode<CP>::PERMUTED;
    // spectrum element index k -> storage slot e of the inverse transform's input, and back
    static __host__ __device__ __forceinline__ int perm(int k) {
        if constexpr (CorrMode<CP>::HYBRID) return CP::in_slot(k);
        else if constexpr (CorrMode<CP>::PFA) return Pfa<CP>::in_slot(k);
        else return k;
    }
    static __host__ __device__ __forceinline__ int unperm(int e) {
        if constexpr (CorrMode<CP>::HYBRID) return CP::slot_to_index(e);
        else if constexpr (CorrMode<CP>::PFA) return Pfa<CP>::slot_to_index(e);
        else return e;
    }
    // element index -> position in the stored array (pairs applied on top of the permutation), and back
    static __host__ __device__ __forceinline__ int slot(int k) { return PairLayout<CP>::pos(perm(k)); }
    static __host__ __device__ __forceinline__ int index_at(int p) { return unperm(PairLayout<CP>::unpos(p)); }
};

// a generic plan with pairwise coprime radices, run WITH its twiddles and natural element order (not as a prime-factor transform)
template <class PL> struct AsPlain : PL { static constexpr bool COPRIME = false; };

// The composite path (acq_composite.hip) picks its base size's correlation plan separately: measured at the configs[3] Galileo
// geometry (N = 2 x 16000, 36 codes) the hybrid 16000 plan runs comp_corr_kernel at 0.396 ms where the plain [25, 20, 32] plan
// takes 0.353 (the pass-0 inputs are formed from 2Q loads per element there and the 128-register cap of 1024 lanes bites
// first), while the fused kernel at N = 16000 gains 37 % from it and N = 5 x 8000 gains 5 %.
template <class PL> struct CompPlanOf { using type = typename CorrPlanOf<PL>::type; };

// natural-order spectra [n][N] -> the stored order of correlation plan CP (pairs and / or permutation), once per handle
template <class CP>
__global__ __launch_bounds__(256) void relayout_kernel(const cf* __restrict__ nat, cf* __restrict__ stored, int n) {
    const size_t total = size_t(n) * CP::N;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < total; i += size_t(gridDim.x) * 256) {
        const size_t c = i / CP::N;
        const int k = int(i - c * CP::N);
        stored[c * CP::N + CorrLayout<CP>::slot(k)] = nat[i];
    }
}
// host: order[p] = the spectrum element stored at position p (PERMUTED layouts; returns 0 and writes nothing otherwise)
template <class CP> inline int fill_order_table(uint16_t* order) {
    if constexpr (!CorrMode<CP>::PERMUTED) return 0;
    else {
        static_assert(CP::N <= 65536, "16-bit element indices");
        if (order) for (int p = 0; p < CP::N; ++p) order[p] = uint16_t(CorrLayout<CP>::index_at(p));
        return CP::N;
    }
}

// one lane's pass-0 elements of one paired array, in registers: R0/2 16-byte loads (+ one 8-byte load when R0 is odd)
template <class PL> struct PairLoad {
    static constexpr int NPAIR = PL::R0 / 2, NB0 = PL::NB(0);
    static constexpr bool ODD = (PL::R0 & 1) != 0;
    u32x4 q[NPAIR > 0 ? NPAIR : 1];
    cf last;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int b, int base_elems) { load_range<0, NPAIR, true>(rs, b, base_elems); }
    // pairs [P0, P1) (and the odd element when TAIL): acq_corr_kernel requests the first pairs of the NEXT transform ahead
    template <int P0, int P1, bool TAIL>
    __device__ __forceinline__ void load_range(__amdgpu_buffer_rsrc_t rs, int b, int base_elems) {
        const int oob = 0x7ffffff0;                         // b >= NB0: out of the descriptor's range -> no request, zeros
        const int v16 = b < NB0 ? b * 16 : oob, v8 = b < NB0 ? b * 8 : oob;
#pragma unroll
        for (int rp = P0; rp < P1; ++rp) q[rp] = __builtin_amdgcn_raw_buffer_load_b128(rs, v16, (base_elems + rp * 2 * NB0) * 8, 0);
        if constexpr (ODD && TAIL) last = buf_load_cf(rs, v8, (base_elems + 2 * NPAIR * NB0) * 8);
    }
    __device__ __forceinline__ cf get(int rn) const {   // rn (natural row) is a compile-time constant after unrolling
        const int r = PairRows<PL>::row(rn);             // its stored row
        if (ODD && r == PL::R0 - 1) return last;
        const u32x4 v = q[r >> 1];
        return (r & 1) ? cf_make(__uint_as_float(v.z), __uint_as_float(v.w)) : cf_make(__uint_as_float(v.x), __uint_as_float(v.y));
    }
};

__device__ __forceinline__ cf load_sample(const void* samples, int fmt, size_t idx) {
    if (fmt == GM_FMT_C32) return reinterpret_cast<const cf*>(samples)[idx];
    if (fmt == GM_FMT_I8_IQ) {
        const char2 v = reinterpret_cast<const char2*>(samples)[idx];
        return cf_make(float(v.x), float(v.y));
    }
    return cf_make(float(reinterpret_cast<const int8_t*>(samples)[idx]), 0.0f);
}

// (value, index) reduction: larger value wins, equal values -> lower index (first strict maximum)
__device__ __forceinline__ void take_better(float& bv, uint32_t& bi, float v, uint32_t i) {
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
}

}  // namespace gm
