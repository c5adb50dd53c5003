// hyb16_core.h — the radix-16 LAST pass of a hybrid plan (fft_core.h HybridPlan, N = 8000 = 125 * 64 as [20, 25, 16]) on the MATRIX
// pipe: lane / slot / element maps and constants (host/device portable: tests/cpu/test_fft_core.cpp emulates the pass lane by lane
// with g++), and the device-only pass() built on v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: an fmaf chain in k order,
// MI355X_MICROARCH.md "f32-input MFMA").
//
// The pass.  Wave group j1 (GW2 waves) runs the A = A1*A2 butterflies whose twiddles are W_B^{j1 r}: butterfly beta reads its 16
// inputs x[r] from 16 consecutive image cells and produces y[q2] = sum_r x[r] W_B^{-+ r (j1 + B1 q2)} — the twiddle and the
// 16-point DFT are ONE 16 x 16 complex matrix M[q2][r] = exp(+-j 2 pi r (j1 + B1 q2) / B), constant per wave group.  As real
// products:  y.re = Mr x.re - Mi x.im,  y.im = Mi x.re + Mr x.im  — four 16 x 16 x 16 products per batch of 16 butterflies, 16
// matrix instructions (k-steps of 4).  Lane l = (g = l >> 4, n = l & 15) supplies inputs r = 4 s + g of butterfly n in k-step s
// and receives outputs q2 = 4 g + i, i = 0 .. 3, of the same butterfly.  A wave takes 64 consecutive beta = 4 batches.
//
// Replaces (where enabled, HybMfma16) Fft<HybridPlan>::last_stage1 / last_stage2 of the inverse transform at
// src/acquisition/do_acquisition.rs:188; the power accumulation (:190-192) and the argmax scan (:195-202) read the outputs from the
// slots described here.
#pragma once
#include "fft_core.h"

namespace gm {

template <class PL, bool INV> struct Mfma16 {
    static_assert(PL::HYBRID && PL::R[2] == 16, "hybrid plan ending in radix 16");
    static constexpr int A = PL::A, A2 = PL::A2, B = PL::B, B1 = PL::B1, GW2 = PL::GW2;
    static constexpr int NBATCH = 4;                          // batches per wave: 64 consecutive beta
    static constexpr int NSLOT = 4 * NBATCH;                  // power slots per lane: slot = 4 t + i
    struct Tab { float c[B], s[B]; };
    static constexpr Tab make() {
        Tab t{};
        for (int e = 0; e < B; ++e) { const ct::cs v = ct::cossin2pi(e, B); t.c[e] = float(v.c); t.s[e] = float(INV ? v.s : -v.s); }
        return t;
    }
    static constexpr Tab tab = make();
    // the lane's matrix entries: row q2 = lane & 15, column r = 4 s + (lane >> 4) of its wave group's matrix
    struct Consts { float mr[4], mi[4], nmi[4]; };
    static GM_HD Consts consts(int tid) {
        Consts m;
        const int q2 = tid & 15, g = (tid >> 4) & 3, j1 = PL::last_j1(tid);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int e = ((4 * s + g) * (j1 + B1 * q2)) % B;
            m.mr[s] = tab.c[e];
            m.mi[s] = tab.s[e];
            m.nmi[s] = -tab.s[e];
        }
        return m;
    }
    static GM_HD bool wave_active(int tid) { return PL::last_j1(tid) < B1; }
    static GM_HD int beta(int tid, int t) { return ((tid >> 6) % GW2) * 64 + 16 * t + (tid & 15); }
    static GM_HD bool slot_active(int tid, int slot) { return wave_active(tid) && beta(tid, slot >> 2) < A; }
    static GM_HD int outq(int tid, int slot) { return 4 * ((tid >> 4) & 3) + (slot & 3); }
    // element index of power slot `slot` of this lane (Good's map on the output side, HybridPlan::out_index)
    static GM_HD int index(int tid, int slot) {
        const int be = beta(tid, slot >> 2), j1 = PL::last_j1(tid);
        const int oa = be / A2 + PL::A1 * (be % A2), ob = j1 + B1 * outq(tid, slot);
        return (B * oa + A * ob) % PL::N;
    }
    // image cell of input r = 4 s + g of batch t's butterfly of this lane (beta >= A: cells of the image that hold other data —
    // the slot is never read back, slot_active)
    static GM_HD int cell(int tid, int t, int s) {
        const int be = beta(tid, t), k1 = be / A2, q1 = be - k1 * A2, g = (tid >> 4) & 3;
        return q1 * PL::STR1 + k1 * PL::GS1 + PL::last_j1(tid) * 16 + 4 * s + g;
    }
#if defined(__HIPCC__)
    struct Gather { cf x[4]; };
    static __device__ __forceinline__ void gather(int t, const cf* lds, int tid, Gather& gth) {
#pragma unroll
        for (int s = 0; s < 4; ++s) gth.x[s] = lds[cell(tid, t, s)];
    }
    // out(slot, y): the complex outputs of batch t
    template <class Out>
    static __device__ __forceinline__ void products(int t, const Gather& gth, const Consts& m, Out&& out) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 dr = {0.f, 0.f, 0.f, 0.f}, di = dr;
        // two accumulation chains, alternating: a chain's next instruction is two issue slots (64 cycles) behind its predecessor,
        // past the 40-cycle dependent latency
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            dr = __builtin_amdgcn_mfma_f32_16x16x4f32(m.mr[s], gth.x[s].x, dr, 0, 0, 0);
            di = __builtin_amdgcn_mfma_f32_16x16x4f32(m.mi[s], gth.x[s].x, di, 0, 0, 0);
            dr = __builtin_amdgcn_mfma_f32_16x16x4f32(m.nmi[s], gth.x[s].y, dr, 0, 0, 0);
            di = __builtin_amdgcn_mfma_f32_16x16x4f32(m.mr[s], gth.x[s].y, di, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) out(4 * t + i, cf_make(dr[i], di[i]));
    }
    // the whole pass of one transform on this lane's wave; the gathers of batch t + 1 are in flight while batch t's products run
    template <class Out>
    static __device__ __forceinline__ void pass(const cf* lds, int tid, const Consts& m, Out&& out) {
        if (!wave_active(tid)) return;                          // wave-uniform
        Gather g[2];
        gather(0, lds, tid, g[0]);
#pragma unroll
        for (int t = 0; t < NBATCH; ++t) {
            if (t + 1 < NBATCH) gather(t + 1, lds, tid, g[(t + 1) & 1]);
            products(t, g[t & 1], m, out);
        }
    }
#endif
};

// which hybrid plans run their last pass on the matrix pipe (acq_corr_kernel)
template <class PL> struct HybMfma16 { static constexpr bool USE = false; };
// what acq_corr_kernel names instead of Mfma16 for every other plan (never used: the uses sit in discarded if-constexpr branches)
struct Mfma16None {
    static constexpr int NSLOT = 0;
    struct Consts {};
};
template <class PL, bool INV, bool USE> struct Mfma16Sel { using type = Mfma16None; };
template <class PL, bool INV> struct Mfma16Sel<PL, INV, true> { using type = Mfma16<PL, INV>; };

}  // namespace gm
