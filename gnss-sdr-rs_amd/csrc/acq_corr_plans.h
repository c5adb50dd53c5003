// acq_corr_plans.h — which plan acq_corr_kernel's inverse transform runs on, per transform size (CorrPlanOf, acq_device.h).
// Included at the end of fft_plans.h; only meaningful to translation units that include acq_device.h.
#pragma once
namespace gm {
template <class PL> struct CorrPlanOf;
// Pass-0 input ROWS of the stored spectra / code spectra (PairLayout, acq_device.h).  Lane b of pass 0 consumes rows r = 0 .. R0 - 1
// (element b + r * NB0); FORCE pairs the rows two by two (16-byte loads) also for a first radix above 25, and row() / nat() store them
// in another order than the natural one.  N = 16368: the radix-33 Good-Thomas butterfly consumes its inputs in the order
// (11 n1 + 3 N2) mod 33, n1 = 0 .. 2 inside N2 = 0 .. 10 (fft_core.h Bfly<33>::s1); stored in THAT order, every 16-byte load of
// acq_corr_ws31_kernel feeds the next two inputs of the butterfly and nothing waits in registers for its partner.
template <class PL> struct PairRows {
    static constexpr bool FORCE = false;
    static constexpr int row(int r) { return r; }      // natural row -> stored row
    static constexpr int nat(int s) { return s; }      // stored row -> natural row
};
template <> struct PairRows<Plan<16368, 768, 33, 16, 31>> {
    static constexpr bool FORCE = true;
    static constexpr int row(int r) { return 3 * ((4 * (r % 11)) % 11) + (2 * (r % 3)) % 3; }     // r = (11 n1 + 3 N2) mod 33 -> 3 N2 + n1
    static constexpr int nat(int s) { return (11 * (s % 3) + 3 * (s / 3)) % 33; }
};
// Stage F's forward transform may run on a plan of its own (same N): the stage is ONE round of D*M workgroups, i.e. bound by the
// latency of a single transform, where a plan of more, smaller passes on more lanes wins — the opposite of what the
// throughput-bound inverse in acq_corr_kernel wants.
template <class PL> struct MixPlanOf { using type = PL; };
#ifndef GM_NO_MIX_PLAN
// stage F at N = 8000: four passes of radix 5 / 8 / 10 / 20 on 1024 lanes (56 VGPRs, 67.5 KB: two workgroups per CU) — 18.1 us per
// 410-transform launch against 23.2 for [25, 20, 16] on 512 lanes (tools/mix_lab; [8,10,10,10]: 18.7, [16,25,20] on 512: 20.1)
#ifndef GM_MIX_PLAN_8000
#define GM_MIX_PLAN_8000 Plan<8000, 1024, 5, 8, 10, 20>
#endif
template <> struct MixPlanOf<Plan<8000, 512, 25, 20, 16>> { using type = GM_MIX_PLAN_8000; };
// N = 16368 (one workgroup per CU either way: 290 transforms are two rounds): radix 16 first on 1024 lanes — every lane loads in
// pass 0 — 45.7 us against 57.4 for [33, 16, 31] on 768 lanes ([16,3,11,31]: 47.8, [11,3,16,31]: 51.6)
#ifndef GM_MIX_PLAN_16368
#define GM_MIX_PLAN_16368 Plan<16368, 1024, 16, 33, 31>
#endif
template <> struct MixPlanOf<Plan<16368, 768, 33, 16, 31>> { using type = GM_MIX_PLAN_16368; };
// N = 16000 (the base of the configs[3] Galileo geometry's composite transform, comp_fwd_sub_kernel: 164 sub-transforms, one
// round): forward sub + post 42.5 -> 38.4 us ([10,10,10,16]: 40.7, [16,10,10,10]: 38.6, [32,25,20]: 47.1; tools/comp_time.py on
// A/B libraries) — the strided decimated loads and the post kernel are most of it
#ifndef GM_MIX_PLAN_16000
#define GM_MIX_PLAN_16000 Plan<16000, 1024, 8, 10, 10, 20>
#endif
template <> struct MixPlanOf<Plan<16000, 1024, 25, 20, 32>> { using type = GM_MIX_PLAN_16000; };
#endif
template <> struct CorrPlanOf<Plan8192> { using type = Plan<8192, 512, 16, 32, 16>; };     // (fft_plans.h: why the two differ)
#ifndef GM_NO_HYBRID_PLANS
using CorrPlan8000 = HybridPlan<8000, 512, 5, 25, 4, 16>;     // 125 * 64: passes of radix 20 / 25 / 16
template <> struct CorrPlanOf<Plan8000> { using type = CorrPlan8000; };
using CorrPlan16000 = HybridPlan<16000, 1024, 5, 25, 4, 32>;  // 125 * 128: radix 20 / 25 / 32 (the Galileo-E1 geometry's composite base)
template <> struct CorrPlanOf<Plan16000> { using type = CorrPlan16000; };
// its pass-0 rows in the order the radix-20 Good-Thomas butterfly consumes them, (5 n1 + 4 N2) mod 20, n1 = 0 .. 3 inside N2 = 0 .. 4
// (fft_core.h Bfly<20>::s1): comp_corr_ws_kernel (acq_comp_ws.h) asks for its inputs pair by pair in that order
template <> struct PairRows<CorrPlan16000> {
    static constexpr bool FORCE = false;
    static constexpr int row(int r) { return 4 * ((4 * (r % 5)) % 5) + r % 4; }     // r = (5 n1 + 4 N2) mod 20 -> 4 N2 + n1
    static constexpr int nat(int s) { return (5 * (s % 4) + 4 * (s / 4)) % 20; }
};
// (N = 16368 = 33 * 16 * 31 runs the generic plan's prime-factor form, fft_core.h Pfa; a dedicated in-place-per-lane image
//  [16 rows][33][31] with one barrier fewer measured 7 % SLOWER — 455 against 425 us per 32-PRN launch — and was dropped)
#endif
}  // namespace gm
