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

int main() {
    double worst = 0;
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
    std::printf("worst %.3e\n", worst);
    return worst < 7e-7 ? 0 : 1;   // f32 FFT rounding of the largest plans; parity tolerances downstream are 1e-5
}
