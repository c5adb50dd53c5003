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
// it < ITF = NBATCH / NWM, power sums in registers — and the one batch left over (33 = 8 x 4 + 1) has wave W0's lane map at it = ITF;
// its eight power sums per lane live in LDS (2 KB of the 25 KB the image leaves free) while the integrations run.  It is run by a
// PASS-0 wave, behind that wave's first half and in front of B1, where the pass-0 waves wait for the matrix waves anyway (~4 500
// cycles per transform): on a matrix wave it made that wave's phase five batches long against the others' four (the whole
// workgroup waits for the longest), and a fifth register set of power sums spills under the 128-register cap.
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
    // One batch in two steps, so that the caller can request batch it + 1's gathers before it runs batch it's products (asked for
    // where they are used, every pair of LDS reads was followed by a full wait: eight exposed LDS latencies per batch, 32 per transform).
    struct Gather { cf up[4], um[4]; };
    static __device__ __forceinline__ void gather(int it, const Bases& bs, int tid, Gather& g) {
        if (batch_active(tid, it)) {                              // wave-uniform
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                g.up[st] = bs.pa[st * 4 * NB + it * 16 * NWM];
                g.um[st] = bs.pb[(3 - st) * 4 * NB + it * 16 * NWM];
            }
        }
    }
    template <class Out>
    static __device__ __forceinline__ void products(int it, const Gather& g, int tid, const Consts& m, Out&& out) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const int kg = (tid >> 4) & 3;
        if (batch_active(tid, it)) {                              // wave-uniform
            f32x4 dcr = {0.f, 0.f, 0.f, 0.f}, dci = dcr, dsr = dcr, dsi = dcr;
            // the 16 operands first, then the 16 matrix instructions back to back: with a vector instruction in front of every matrix
            // instruction (the form hipcc picks to save registers) each of them waits for its turn on the vector pipe, which the
            // pass-0 waves of the same SIMD keep busy, and the matrix pipe idles in between (measured: 42 % busy in this phase)
            cf a[4], b[4];
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int k = 4 * st + kg;
                const cf up = g.up[st], um = g.um[st];
                a[st] = k == 0 ? up : cf_add(up, um);
                b[st] = k == 0 ? cf_make(0.f, 0.f) : cf_sub(up, um);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                dcr = __builtin_amdgcn_mfma_f32_16x16x4f32(m.c[st], a[st].x, dcr, 0, 0, 0);
                dci = __builtin_amdgcn_mfma_f32_16x16x4f32(m.c[st], a[st].y, dci, 0, 0, 0);
                dsr = __builtin_amdgcn_mfma_f32_16x16x4f32(m.s[st], b[st].x, dsr, 0, 0, 0);
                dsi = __builtin_amdgcn_mfma_f32_16x16x4f32(m.s[st], b[st].y, dsi, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float car = dcr[r], cai = dci[r], sbr = dsr[r], sbi = dsi[r];
                out(it, 2 * r, cf_make(car - sbi, cai + sbr));                                      // y[q]      = ca + j sb
                const bool none = (kg == 0 && r == 0);
                out(it, 2 * r + 1, none ? cf_make(0.f, 0.f) : cf_make(car + sbi, cai - sbr));      // y[31 - q] = ca - j sb
            }
        }
    }
    template <class Out>
    static __device__ __forceinline__ void batch(int it, const Bases& bs, int tid, const Consts& m, Out&& out) {
        Gather g;
        gather(it, bs, tid, g);
        products(it, g, tid, m, out);
    }
    // the radix-31 pass of one transform on this lane's wave: batches 0 .. ITF - 1 into out(it, r8, y); the gathers of batch it + 1 are
    // in flight while batch it's products run
    struct NoMark { __device__ __forceinline__ void operator()(int) const {} };
    template <class Out, class Mark = NoMark>
    static __device__ __forceinline__ void pass(const Bases& bs, int tid, const Consts& m, Out&& out, Mark&& mark = Mark()) {
        Gather g[2];
        gather(0, bs, tid, g[0]);
#pragma unroll
        for (int it = 0; it < ITF; ++it) {
            if (it + 1 < ITF) gather(it + 1, bs, tid, g[(it + 1) & 1]);
            products(it, g[it & 1], tid, m, out);
            mark(it);                                           // (diagnostic hook: phase stamps of tools/corr_lab/ws31_stamps.hip)
        }
    }
    // the left-over batch (it = ITF of wave W0's lane map): run by whichever wave has the time, with tid = that map's lane
    // (64 W0 + lane) and the constants / bases of that lane
    template <class Out>
    static __device__ __forceinline__ void left_over_batch(const Bases& bs, int tid_w0, const Consts& m, Out&& out) {
        if constexpr (EXTRA == 1) batch(ITF, bs, tid_w0, m, out);
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
