// ws31_core.h — the host/device-portable part of the wave-specialised N = 16368 kernel (acq_corr_ws31.h): the plan, the matrix-pipe
// radix-31 pass's lane / slot / element maps and constants, the streaming 11-point DFT of pass 0's second half.  tests/cpu/
// test_fft_core.cpp emulates the whole transform from these (g++, no GPU): stored order -> pass 0 -> radix 16 -> the matrix products
// lane by lane -> element indices, against a float64 DFT.  The device-only batch() (MFMA builtins) is compiled by hipcc alone.
#pragma once
#include "fft_core.h"
#include "fft_plans.h"

namespace gm {

template <int N_> struct Ws31PlanOf { using type = Plan<N_, 1024, 33, 16, 31>; };

// the matrix-pipe radix-31 pass: NWM waves starting at wave W0 share the NB / 16 batches — wave w takes batches (w - W0) + NWM it,
// it < ITF = NBATCH / NWM, power sums in registers — and the one batch left over (33 = 8 x 4 + 1) is wave W0's batch it = ITF, whose
// eight power sums per lane live in LDS (2 KB of the 25 KB the image leaves free) while the integrations run: a fifth set of
// registers (40 power sums + the pass's ~80) spills under the 128-register cap, and a scratch reload inside the matrix loop queues
// behind the pass-0 waves' 528 loads (measured: 3 300 cycles per batch instead of 1 200); on a pass-0 wave instead, the batch delays
// that wave's loads and the whole workgroup waits for it at B1 (2 400 - 4 000 cycles per transform).
template <class PL, int W0, int NWM> struct Mfma31 {
    static constexpr int NB = PL::NB(PL::NP - 1), NBATCH = NB / 16;
    static constexpr int ITF = NBATCH / NWM, EXTRA = NBATCH - NWM * ITF, ITL = ITF + EXTRA, RL = 8;
    static_assert(PL::R[PL::NP - 1] == 31 && NB % 16 == 0 && PL::COPRIME, "prime-factor plan ending in radix 31");
    static_assert(EXTRA == 0 || EXTRA == 1, "at most one batch left over");
    struct Tab { float c[31], s[31]; };
    static constexpr Tab make() {
        Tab t{};
        for (int e = 0; e < 31; ++e) { const ct::cs v = ct::cossin2pi(e, 31); t.c[e] = float(v.c); t.s[e] = float(v.s); }
        return t;
    }
    static constexpr Tab tab = make();
    struct Consts { float c[4], s[4]; };                       // the lane's matrix entries: row q = lane & 15, column k = 4 s + (lane >> 4)
    static GM_HD Consts consts(int tid) {
        Consts m;
        const int q = tid & 15, kg = (tid >> 4) & 3;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int e = (q * (4 * st + kg)) % 31;
            m.c[st] = tab.c[e];
            m.s[st] = tab.s[e];                                   // inverse transform (e^{+...}): y[q] = ca + j sb with s = +sin (DftPrime)
        }
        return m;
    }
    static GM_HD int batch_of(int tid, int it) { return (tid >> 6) - W0 + NWM * it; }
    static GM_HD bool batch_active(int tid, int it) {
        const int w = (tid >> 6) - W0;
        return w >= 0 && w < NWM && w + NWM * it < NBATCH;
    }
    static GM_HD bool slot_ok(int tid, int r8) { return !(((tid >> 4) & 3) == 0 && r8 == 1); }   // q = 0 has no partner output
    static GM_HD int bfly(int tid, int it) { return 16 * batch_of(tid, it) + (tid & 15); }
    static GM_HD int outq(int tid, int r8) { const int q = 4 * ((tid >> 4) & 3) + (r8 >> 1); return (r8 & 1) ? 31 - q : q; }
    // element index of power slot (it, r8) of this lane (Good's map on the output side of the prime-factor transform, fft_core.h Pfa)
    static GM_HD int index(int tid, int it, int r8) { return Pfa<PL>::out_index(bfly(tid, it), outq(tid, r8)); }
    // batch `it` of this lane's wave: gathers from the LDS image (no twiddles), out(it, r8, y) receives the complex outputs
    // The gathers of ALL batches of a lane hang off TWO base addresses plus compile-time offsets (<= 54 784 bytes: inside a DS
    // instruction's 16-bit offset field): row k = 4 st + kg of batch it is pa[st * 4 NB + it * 16 NWM], its partner row 31 - k
    // is pb[(3 - st) * 4 NB + it * 16 NWM].  (Written with one index expression per read, hipcc kept a precomputed address per
    // (batch, row) alive across the integrations loop, spilt them, and every batch began with scratch reloads queued behind the
    // pass-0 waves' loads: 3 300 cycles per batch instead of ~700.)  For k = 0 the partner read is row 31, one row BEHIND the image:
    // the image is declared one row longer and the value is discarded.
    struct Bases { const cf *pa, *pb; };
    static GM_HD Bases bases(const cf* lds, int tid) {
        const int kg = (tid >> 4) & 3, col = 16 * ((tid >> 6) - W0) + (tid & 15);
        return Bases{lds + col + kg * NB, lds + col + (19 - kg) * NB};
    }
#if defined(__HIPCC__)
    template <class Out>
    static __device__ __forceinline__ void batch(int it, const Bases& bs, int tid, const Consts& m, Out&& out) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const int kg = (tid >> 4) & 3;
        if (batch_active(tid, it)) {                              // wave-uniform
            f32x4 dcr = {0.f, 0.f, 0.f, 0.f}, dci = dcr, dsr = dcr, dsi = dcr;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int k = 4 * st + kg;
                const cf up = bs.pa[st * 4 * NB + it * 16 * NWM];
                const cf um = bs.pb[(3 - st) * 4 * NB + it * 16 * NWM];
                const cf a = k == 0 ? up : cf_add(up, um);
                const cf b = k == 0 ? cf_make(0.f, 0.f) : cf_sub(up, um);
                dcr = __builtin_amdgcn_mfma_f32_16x16x4f32(m.c[st], a.x, dcr, 0, 0, 0);
                dci = __builtin_amdgcn_mfma_f32_16x16x4f32(m.c[st], a.y, dci, 0, 0, 0);
                dsr = __builtin_amdgcn_mfma_f32_16x16x4f32(m.s[st], b.x, dsr, 0, 0, 0);
                dsi = __builtin_amdgcn_mfma_f32_16x16x4f32(m.s[st], b.y, dsi, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float car = dcr[r], cai = dci[r], sbr = dsr[r], sbi = dsi[r];
                out(it, 2 * r, cf_make(car - sbi, cai + sbr));                                      // y[q]      = ca + j sb
                const bool none = (kg == 0 && r == 0);
                out(it, 2 * r + 1, none ? cf_make(0.f, 0.f) : cf_make(car + sbi, cai - sbr));      // y[31 - q] = ca - j sb
            }
        }
    }
#endif
};

// Inverse 11-point DFT in the symmetric real-coefficient form (fft_core.h DftPrime), STREAMING: the pair sums / differences
// replace the inputs, every output pair is handed to `out` as soon as it exists.  Live state: the 11 inputs + 2 complex sums —
// the generic Bfly keeps inputs, a / b and all 11 outputs (68 registers), which does not fit beside the radix-33 butterfly's 33
// intermediate values under this kernel's 128-register cap.
template <int Q, int J> GM_HD void dft11_acc(const cf (&u)[11], cf& ca, cf& sb);
template <int Q, class Out> GM_HD void dft11_q(const cf (&u)[11], Out&& out);
template <class Out> GM_HD void dft11_inv_stream(cf (&u)[11], Out&& out) {
    cf y0 = u[0];
#pragma unroll
    for (int j = 1; j <= 5; ++j) {
        const cf a = cf_add(u[j], u[11 - j]), b = cf_sub(u[j], u[11 - j]);
        u[j] = a; u[11 - j] = b;                                  // a_j in u[j], b_j in u[11 - j]
        y0 = cf_add(y0, a);
    }
    out(0, y0);
    dft11_q<1>(u, out);
}
template <int Q, class Out> GM_HD void dft11_q(const cf (&u)[11], Out&& out) {
    if constexpr (Q <= 5) {
        cf ca = u[0], sb = cf_make(0.f, 0.f);
        dft11_acc<Q, 1>(u, ca, sb);
        const cf jsb = cf_mulj<true>(sb);                         // inverse: y[q] = ca + j sb, y[11 - q] = ca - j sb
        out(Q, cf_add(ca, jsb));
        out(11 - Q, cf_sub(ca, jsb));
        dft11_q<Q + 1>(u, out);
    }
}
template <int Q, int J> GM_HD void dft11_acc(const cf (&u)[11], cf& ca, cf& sb) {
    if constexpr (J <= 5) {
        constexpr ct::cs v = ct::cossin2pi(long(J) * Q, 11);
        constexpr float c = float(v.c), sn = float(v.s);
        ca.x = __builtin_fmaf(c, u[J].x, ca.x); ca.y = __builtin_fmaf(c, u[J].y, ca.y);
        sb.x = __builtin_fmaf(sn, u[11 - J].x, sb.x); sb.y = __builtin_fmaf(sn, u[11 - J].y, sb.y);
        dft11_acc<Q, J + 1>(u, ca, sb);
    }
}

}  // namespace gm
