// acq_corr_plans.h — which plan acq_corr_kernel's inverse transform runs on, per transform size (CorrPlanOf, acq_device.h).
// Included at the end of fft_plans.h; only meaningful to translation units that include acq_device.h.
#pragma once
namespace gm {
template <class PL> struct CorrPlanOf;
#ifndef GM_NO_HYBRID_PLANS
using CorrPlan8000 = HybridPlan<8000, 512, 5, 25, 4, 16>;     // 125 * 64: passes of radix 20 / 25 / 16
template <> struct CorrPlanOf<Plan8000> { using type = CorrPlan8000; };
using CorrPlan16000 = HybridPlan<16000, 1024, 5, 25, 4, 32>;  // 125 * 128: radix 20 / 25 / 32 (the Galileo-E1 geometry's composite base)
template <> struct CorrPlanOf<Plan16000> { using type = CorrPlan16000; };
// (N = 16368 = 33 * 16 * 31 runs the generic plan's prime-factor form, fft_core.h Pfa; a dedicated in-place-per-lane image
//  [16 rows][33][31] with one barrier fewer measured 7 % SLOWER — 455 against 425 us per 32-PRN launch — and was dropped)
#endif
}  // namespace gm
