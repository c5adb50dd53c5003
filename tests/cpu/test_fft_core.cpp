// CPU emulation of the in-LDS FFT (gnss-sdr-rs_amd/csrc/fft_core.h): the T "threads" of a
// workgroup are run phase by phase, a phase boundary standing for a workgroup barrier.
// Checks every shipped plan, forward and inverse, against a float64 O(N^2) DFT.
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft_core.h"
#include "fft_plans.h"
#include "ws31_core.h"

using namespace gm;

template <class PL, bool INV, int S, bool PFA = false> struct Middle {
    static void run(std::vector<cf>& lds, const std::vector<cf>& tw) {
        if constexpr (S <= PL::NP - 2) {
            constexpr int IT = PL::IT(S), R = PL::R[S];
            std::vector<cf> regs(size_t(PL::T) * IT * R);
            for (int tid = 0; tid < PL::T; ++tid) {   // phase: gather + first half of the butterfly
                auto& v = *reinterpret_cast<cf(*)[IT][R]>(&regs[size_t(tid) * IT * R]);
                Fft<PL, INV, PFA>::template mid_stage1<S>(v, lds.data(), tw.data(), tid);
            }
            for (int tid = 0; tid < PL::T; ++tid) {   // barrier; phase: second half + scatter
                auto& v = *reinterpret_cast<cf(*)[IT][R]>(&regs[size_t(tid) * IT * R]);
                Fft<PL, INV, PFA>::template mid_stage2<S>(v, lds.data(), tid);
            }
            Middle<PL, INV, S + 1, PFA>::run(lds, tw);
        }
    }
};

// PFA: the prime-factor form across the passes (plans with pairwise coprime radices): element i of the input sits at storage
// slot Pfa::in_slot(i), output (b, q) of the last pass is element Pfa::out_index(b, q); no twiddles
template <class PL, bool INV, bool PFA = false> static double run_plan(const char* name) {
    constexpr int N = PL::N, T = PL::T;
    std::vector<cf> x(N), xs(N), y(N), lds(PL::LDS_ELEMS), tw(PL::TW_TOTAL + 1);
    fill_twiddles<PL>(tw.data(), INV, [](double a) { return std::cos(a); }, [](double a) { return std::sin(a); });
    unsigned s = 12345u + N;
    for (int i = 0; i < N; ++i) {
        s = s * 1664525u + 1013904223u; float a = float(int(s >> 8) % 2001 - 1000) / 100.f;
        s = s * 1664525u + 1013904223u; float b = float(int(s >> 8) % 2001 - 1000) / 100.f;
        x[i] = cf_make(a, b);
    }
    for (int i = 0; i < N; ++i) xs[PFA ? Pfa<PL>::in_slot(i) : i] = x[i];
    {   // pass 0: stage 1 for every thread, (barrier), stage 2 for every thread
        std::vector<cf> regs(size_t(T) * PL::IT0 * PL::R0);
        for (int tid = 0; tid < T; ++tid) {
            auto& v = *reinterpret_cast<cf(*)[PL::IT0][PL::R0]>(&regs[size_t(tid) * PL::IT0 * PL::R0]);
            Fft<PL, INV, PFA>::pass0_stage1(v, [&](int it, int r) { return xs[(tid + it * T) + r * PL::NB(0)]; }, tid);
        }
        for (int tid = 0; tid < T; ++tid) {
            auto& v = *reinterpret_cast<cf(*)[PL::IT0][PL::R0]>(&regs[size_t(tid) * PL::IT0 * PL::R0]);
            Fft<PL, INV, PFA>::pass0_stage2(v, lds.data(), tid);
        }
    }
    Middle<PL, INV, 1, PFA>::run(lds, tw);
    for (int tid = 0; tid < T; ++tid) {
        cf v[PL::ITL][PL::RL];
        Fft<PL, INV, PFA>::last_stage1(v, lds.data(), tw.data(), tid);
        Fft<PL, INV, PFA>::last_stage2(v, [&](int it, int r, cf val) {
            const int b = tid + it * T;
            y[PFA ? Pfa<PL>::out_index(b, r) : b + r * PL::NB(PL::NP - 1)] = val; }, tid);
    }
    // reference: float64 DFT
    std::vector<std::complex<double>> w(N);
    for (int i = 0; i < N; ++i) w[i] = std::polar(1.0, (INV ? 2.0 : -2.0) * M_PI * i / N);
    double num = 0, den = 0;
    for (int k = 0; k < N; ++k) {
        std::complex<double> acc = 0;
        size_t idx = 0;
        for (int n = 0; n < N; ++n) { acc += std::complex<double>(x[n].x, x[n].y) * w[idx]; idx += k; if (idx >= size_t(N)) idx -= N; }
        std::complex<double> d = acc - std::complex<double>(y[k].x, y[k].y);
        num += std::norm(d); den += std::norm(acc);
    }
    double err = std::sqrt(num / den);
    std::printf("%-28s N=%6d T=%4d %s%s rel_l2_err=%.3e lds_elems=%d tw=%d\n", name, N, T, INV ? "inv" : "fwd", PFA ? " prime-factor" : "", err,
                PL::LDS_ELEMS, PL::TW_TOTAL);
    return err;
}

// hybrid prime-factor / Cooley-Tukey plans (fft_core.h HybridPlan): element i at slot in_slot(i), output (lane, q2) is element
// out_index(lane, q2); slot_to_index must invert in_slot
template <class HP, bool INV> static double run_hybrid(const char* name) {
    constexpr int N = HP::N, T = HP::T;
    std::vector<cf> x(N), xs(N), y(N, cf_make(1e30f, 1e30f)), lds(HP::LDS_ELEMS);
    unsigned s = 777u + N;
    for (int i = 0; i < N; ++i) {
        s = s * 1664525u + 1013904223u; float a = float(int(s >> 8) % 2001 - 1000) / 100.f;
        s = s * 1664525u + 1013904223u; float b = float(int(s >> 8) % 2001 - 1000) / 100.f;
        x[i] = cf_make(a, b);
    }
    std::vector<char> seen(N, 0);
    for (int i = 0; i < N; ++i) {
        const int e = HP::in_slot(i);
        if (e < 0 || e >= N || seen[e] || HP::slot_to_index(e) != i) { std::printf("%s: in_slot / slot_to_index broken at %d\n", name, i); return 1.0; }
        seen[e] = 1;
        xs[e] = x[i];
    }
    using F = Fft<HP, INV>;
    constexpr int IT1 = HP::IT(1);
    {
        std::vector<cf> regs(size_t(T) * HP::R0);
        for (int tid = 0; tid < T; ++tid)
            F::pass0_stage1(*reinterpret_cast<cf(*)[1][HP::R0]>(&regs[size_t(tid) * HP::R0]), [&](int, int r) { return xs[tid + r * HP::NB(0)]; }, tid);
        for (int tid = 0; tid < T; ++tid) F::pass0_stage2(*reinterpret_cast<cf(*)[1][HP::R0]>(&regs[size_t(tid) * HP::R0]), lds.data(), tid);
    }
    {
        // (the middle pass is in place per lane: running every lane's reads before any lane's writes is one legal order)
        std::vector<cf> regs(size_t(T) * IT1 * HP::R[1]);
        for (int tid = 0; tid < T; ++tid) F::template mid_stage1<1>(*reinterpret_cast<cf(*)[IT1][HP::R[1]]>(&regs[size_t(tid) * IT1 * HP::R[1]]), lds.data(), nullptr, tid);
        for (int tid = 0; tid < T; ++tid) F::template mid_stage2<1>(*reinterpret_cast<cf(*)[IT1][HP::R[1]]>(&regs[size_t(tid) * IT1 * HP::R[1]]), lds.data(), tid);
    }
    for (int tid = 0; tid < T; ++tid) {
        cf v[1][HP::RL];
        F::last_stage1(v, lds.data(), nullptr, tid);
        F::last_stage2(v, [&](int, int q, cf val) { y[HP::out_index(tid, q)] = val; }, tid);
    }
    std::vector<std::complex<double>> w(N);
    for (int i = 0; i < N; ++i) w[i] = std::polar(1.0, (INV ? 2.0 : -2.0) * M_PI * i / N);
    double num = 0, den = 0;
    for (int k = 0; k < N; ++k) {
        std::complex<double> acc = 0;
        size_t idx = 0;
        for (int n = 0; n < N; ++n) { acc += std::complex<double>(x[n].x, x[n].y) * w[idx]; idx += k; if (idx >= size_t(N)) idx -= N; }
        std::complex<double> d = acc - std::complex<double>(y[k].x, y[k].y);
        num += std::norm(d); den += std::norm(acc);
    }
    const double err = std::sqrt(num / den);
    std::printf("%-28s N=%6d T=%4d %s hybrid rel_l2_err=%.3e lds_elems=%d\n", name, N, T, INV ? "inv" : "fwd", err, HP::LDS_ELEMS);
    return err;
}

// The wave-specialised N = 16368 transform of csrc/acq_corr_ws31.h, emulated lane by lane from csrc/ws31_core.h: the stored order
// (prime-factor input permutation, then row pairs in the radix-33 butterfly's consumption order: PairRows) -> pass 0 as the
// pass-0 waves run it (inputs asked for by stored row, Bfly<33> first half, streaming 11-point second half) -> the radix-16 pass
// -> the radix-31 pass as the matrix waves run it: every batch's gathers through the two base addresses, the four 16 x 16 x 16
// products formed from the LANES' constants exactly as v_mfma_f32_16x16x4_f32 combines them (A[l & 15][l >> 4], B[l >> 4][l & 15],
// D[4 (l >> 4) + r][l & 15]; an fmaf chain in k order), the left-over batch, the slot -> element map.  Every element must be
// produced exactly once and equal the float64 inverse DFT.
static double run_ws31() {
    using PLX = gm::Plan16368;
    using PL = gm::Ws31PlanOf<PLX::N>::type;
    using MF = gm::Mfma31<PL, 8, 8>;
    using PR = gm::PairRows<PLX>;
    constexpr int N = PL::N, T = PL::T, NB0 = PL::NB(0), NB = MF::NB;
    static_assert(PR::FORCE && T == 1024 && NB0 == 496 && NB == 528 && MF::ITF == 4 && MF::EXTRA == 1, "the shipped shape");
    std::vector<cf> x(N), st(N + 16), lds(PL::LDS_ELEMS + NB, cf_make(1e30f, -1e30f)), y(N, cf_make(1e30f, 1e30f));
    unsigned sd = 4242u;
    for (int i = 0; i < N; ++i) {
        sd = sd * 1664525u + 1013904223u; float a = float(int(sd >> 8) % 2001 - 1000) / 100.f;
        sd = sd * 1664525u + 1013904223u; float b = float(int(sd >> 8) % 2001 - 1000) / 100.f;
        x[i] = cf_make(a, b);
    }
    // stored order: element k -> prime-factor slot e -> (row rn = e / NB0, lane b) -> stored row s = row(rn) -> position
    std::vector<char> used(N, 0);
    for (int r = 0; r < 33; ++r) if (PR::nat(PR::row(r)) != r || PR::row(PR::nat(r)) != r) { std::printf("ws31: PairRows row / nat are not inverse at %d\n", r); return 1.0; }
    for (int k = 0; k < N; ++k) {
        const int e = Pfa<PL>::in_slot(k), rn = e / NB0, b = e - rn * NB0, s_ = PR::row(rn);
        const int pos = s_ < 32 ? ((s_ >> 1) * NB0 + b) * 2 + (s_ & 1) : 32 * NB0 + b;
        if (pos < 0 || pos >= N || used[pos]) { std::printf("ws31: stored order is not a permutation at %d\n", k); return 1.0; }
        used[pos] = 1;
        st[pos] = x[k];
    }
    // pass 0 on lanes 0 .. 495: the s-th input the butterfly asks for must be stored row s (16-byte pair s >> 1, half s & 1)
    {
        std::vector<cf> regs(size_t(NB0) * 33);
        for (int tid = 0; tid < NB0; ++tid) {
            int calls = 0; bool in_order = true;
            auto in = [&](int, int r) {
                const int s_ = PR::row(r);
                in_order = in_order && s_ == calls++;
                return s_ < 32 ? st[((s_ >> 1) * NB0 + tid) * 2 + (s_ & 1)] : st[32 * NB0 + tid];
            };
            Fft<PL, true, true>::pass0_stage1(*reinterpret_cast<cf(*)[1][33]>(&regs[size_t(tid) * 33]), in, tid);
            if (!in_order || calls != 33) { std::printf("ws31: pass 0 does not consume its rows in stored order (lane %d)\n", tid); return 1.0; }
        }
        for (int tid = 0; tid < NB0; ++tid) {
            cf (&v0)[33] = *reinterpret_cast<cf(*)[33]>(&regs[size_t(tid) * 33]);
            constexpr int EA = Bfly<33, true>::EA, EB = Bfly<33, true>::EB;
            cf* dst = lds.data() + tid * 33;
            for (int k1 = 0; k1 < 3; ++k1) {
                cf u[11];
                for (int n2 = 0; n2 < 11; ++n2) u[n2] = v0[n2 * 3 + k1];
                gm::dft11_inv_stream(u, [&](int k2, cf val) { dst[(k1 * EA + k2 * EB) % 33] = val; });
            }
        }
    }
    Middle<PL, true, 1, true>::run(lds, std::vector<cf>(1));
    // the radix-31 pass on waves 8 .. 15
    std::vector<typename MF::Consts> mc(T);
    for (int tid = 0; tid < T; ++tid) mc[tid] = MF::consts(tid);
    int produced = 0;
    for (int wave = 8; wave < 16; ++wave) {
        for (int it = 0; it < MF::ITL; ++it) {
            if (!MF::batch_active(wave * 64, it)) continue;
            float a_re[4][64], a_im[4][64], b_re[4][64], b_im[4][64];     // [k-step][lane]
            for (int l = 0; l < 64; ++l) {
                const int tid = wave * 64 + l, kg = (l >> 4) & 3;
                const typename MF::Bases bs = MF::bases(lds.data(), tid);
                for (int stp = 0; stp < 4; ++stp) {
                    const int k = 4 * stp + kg;
                    const cf up = bs.pa[stp * 4 * NB + it * 16 * 8], um = bs.pb[(3 - stp) * 4 * NB + it * 16 * 8];
                    if (k > 0 && (bs.pa + stp * 4 * NB + it * 16 * 8 != lds.data() + MF::bfly(tid, it) + k * NB ||
                                  bs.pb + (3 - stp) * 4 * NB + it * 16 * 8 != lds.data() + MF::bfly(tid, it) + (31 - k) * NB)) {
                        std::printf("ws31: gather addresses off (wave %d lane %d it %d step %d)\n", wave, l, it, stp); return 1.0;
                    }
                    const cf a = k == 0 ? up : cf_add(up, um), b = k == 0 ? cf_make(0.f, 0.f) : cf_sub(up, um);
                    a_re[stp][l] = a.x; a_im[stp][l] = a.y; b_re[stp][l] = b.x; b_im[stp][l] = b.y;
                }
            }
            for (int l = 0; l < 64; ++l) {          // D[q = 4 (l >> 4) + r][n = l & 15] = sum over k-steps, then over kg, of A[q][k] * B[k][n]
                const int tid = wave * 64 + l, n = l & 15;
                for (int r = 0; r < 4; ++r) {
                    const int q = 4 * (l >> 4) + r;
                    float dcr = 0.f, dci = 0.f, dsr = 0.f, dsi = 0.f;
                    for (int stp = 0; stp < 4; ++stp)
                        for (int kg = 0; kg < 4; ++kg) {
                            const typename MF::Consts& m = mc[wave * 64 + 16 * kg + q];      // the lane that holds A[q][4 stp + kg]
                            const int src = 16 * kg + n;                                     // ... and the one that holds B[4 stp + kg][n]
                            dcr = __builtin_fmaf(m.c[stp], a_re[stp][src], dcr); dci = __builtin_fmaf(m.c[stp], a_im[stp][src], dci);
                            dsr = __builtin_fmaf(m.s[stp], b_re[stp][src], dsr); dsi = __builtin_fmaf(m.s[stp], b_im[stp][src], dsi);
                        }
                    for (int side = 0; side < 2; ++side) {
                        const int r8 = 2 * r + side;
                        if (!MF::slot_ok(tid, r8)) continue;
                        const int idx = MF::index(tid, it, r8);
                        if (idx < 0 || idx >= N || y[idx].x != 1e30f) { std::printf("ws31: slot -> element map is not a bijection at %d\n", idx); return 1.0; }
                        y[idx] = side == 0 ? cf_make(dcr - dsi, dci + dsr) : cf_make(dcr + dsi, dci - dsr);
                        ++produced;
                    }
                }
            }
        }
    }
    if (produced != N) { std::printf("ws31: %d of %d elements produced\n", produced, N); return 1.0; }
    std::vector<std::complex<double>> w(N);
    for (int i = 0; i < N; ++i) w[i] = std::polar(1.0, 2.0 * M_PI * i / N);
    double num = 0, den = 0;
    for (int k = 0; k < N; ++k) {
        std::complex<double> acc = 0;
        size_t idx = 0;
        for (int n = 0; n < N; ++n) { acc += std::complex<double>(x[n].x, x[n].y) * w[idx]; idx += k; if (idx >= size_t(N)) idx -= N; }
        const std::complex<double> d = acc - std::complex<double>(y[k].x, y[k].y);
        num += std::norm(d); den += std::norm(acc);
    }
    const double err = std::sqrt(num / den);
    std::printf("%-28s N=%6d T=%4d inv wave-specialised (pass 0 on 8 waves, radix 31 as 16x16x16 matrix products on 8) rel_l2_err=%.3e\n", "ws31<Plan16368>", N, T, err);
    return err;
}

// What the wave-specialised composite kernel (csrc/acq_comp_ws.h) assumes about the hybrid base-16000 plan, checked for every lane:
//   * the rows of the stored spectra are in the radix-20 butterfly's consumption order: stored row s holds natural row
//     (5 n1 + 4 N2) mod 20 with s = 4 N2 + n1 (Bfly<20>::s1 asks for its inputs in that order), and row() / nat() are inverse;
//   * a last-pass lane's 32 outputs are the elements (e0 + STEP q2) mod N, STEP = N / 32: one wrap — its fold of the power sums
//     picks the lowest element among equal values from the slot numbers alone (first slot at or behind the wrap, else the first).
static int check_comp_ws_assumptions() {
    using HP = HybridPlan<16000, 1024, 5, 25, 4, 32>;
    using PR = PairRows<HP>;
    int bad = 0;
    for (int s = 0; s < 20; ++s) {
        const int n1 = s % 4, n2 = s / 4;
        if (PR::nat(s) != (5 * n1 + 4 * n2) % 20 || PR::row(PR::nat(s)) != s) ++bad;
    }
    constexpr int STEP = HP::N / HP::RL;
    for (int tid = 0; tid < 512; ++tid) {
        if (!HP::last_active(tid)) continue;
        const int e0 = HP::out_index(tid, 0), rw = (HP::N - e0 + STEP - 1) / STEP;
        int prev = -1, wraps = 0;
        for (int r = 0; r < HP::RL; ++r) {
            const int e = HP::out_index(tid, r);
            if (e != (e0 + STEP * r) % HP::N) ++bad;
            if ((r >= rw) != (e0 + STEP * r >= HP::N)) ++bad;
            if (e < prev) ++wraps;
            prev = e;
        }
        if (wraps > 1) ++bad;
        // the rule itself against a search over indices, on every pair of slots holding the maximum
        for (int a = 0; a < HP::RL; ++a)
            for (int b = a + 1; b < HP::RL; ++b) {
                const int lowest = HP::out_index(tid, a) < HP::out_index(tid, b) ? a : b;
                const int r_any = a, r_wrapped = a >= rw ? a : (b >= rw ? b : -1);
                if ((r_wrapped >= 0 ? r_wrapped : r_any) != lowest) ++bad;
            }
    }
    std::printf("comp_ws assumptions on Hybrid16000: %d violations\n", bad);
    return bad;
}

int main() {
    double worst = 0;
    if (check_comp_ws_assumptions()) return 1;
    worst = std::fmax(worst, run_hybrid<HybridPlan<8000, 512, 5, 25, 4, 16>, true>("Hybrid8000 [20,25,16]"));
    worst = std::fmax(worst, run_hybrid<HybridPlan<8000, 512, 5, 25, 4, 16>, false>("Hybrid8000 [20,25,16]"));
    worst = std::fmax(worst, run_hybrid<HybridPlan<16000, 1024, 5, 25, 4, 32>, true>("Hybrid16000 [20,25,32]"));
#define RUN(PL) worst = std::fmax(worst, run_plan<PL, false>(#PL)); worst = std::fmax(worst, run_plan<PL, true>(#PL)); \
    if constexpr (PL::COPRIME) { worst = std::fmax(worst, run_plan<PL, false, true>(#PL)); worst = std::fmax(worst, run_plan<PL, true, true>(#PL)); }
    GM_FOR_EACH_PLAN(RUN)
    {   // stage F's own forward plan at N = 8000 (MixPlanOf, acq_corr_plans.h): four passes on 1024 lanes
        using MixPlan8000 = gm::MixPlanOf<gm::Plan8000>::type;
        worst = std::fmax(worst, run_plan<MixPlan8000, false>("MixPlanOf<Plan8000>"));
        worst = std::fmax(worst, run_plan<MixPlan8000, true>("MixPlanOf<Plan8000>"));
        using MixPlan16000 = gm::MixPlanOf<gm::Plan16000>::type;
        worst = std::fmax(worst, run_plan<MixPlan16000, false>("MixPlanOf<Plan16000>"));
        using MixPlan16368 = gm::MixPlanOf<gm::Plan16368>::type;
        worst = std::fmax(worst, run_plan<MixPlan16368, false>("MixPlanOf<Plan16368>"));
    }
    worst = std::fmax(worst, run_ws31());
    std::printf("worst %.3e\n", worst);
    return worst < 7e-7 ? 0 : 1;   // f32 FFT rounding of the largest plans; parity tolerances downstream are 1e-5
}
