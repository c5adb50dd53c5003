// fft_plans.h — the transform sizes the library ships kernels for, and their radix plans.
// N = samples per code period = round(fs / 1 kHz) for GPS L1 C/A (do_acquisition.rs:249-251).
//   8000  : 8 Msps            (BASELINE config 2, the bench workload)    2^6 * 5^3
//   16368 : 16.3676 Msps      (BASELINE config 1, the reference's test capture geometry) 2^4*3*11*31
//   4096  : 4.096 Msps        (the reference's synthetic tracking tests, do_tracking.rs:467)
//   2048, 1024 : small parity cases
//   2000, 4000, 5000, 6000, 8192, 10000, 12000, 15000, 16000, 16384 : other common front-end rates
// Plan<N, T, radices...>: first radix odd where possible (conflict-free stride-R scatter),
// T >= N / R for every pass so each thread owns at most one butterfly per pass (except where noted).
#pragma once
#include "fft_core.h"

namespace gm {
using Plan8000 = Plan<8000, 512, 25, 20, 16>;
// 16368 = 33 * 16 * 31, pairwise coprime: the correlation kernel runs its inverse as a prime-factor transform ACROSS the
// passes (fft_core.h Pfa: no twiddles at all).  12 waves = 3 per SIMD (576 lanes = 9 waves loaded the SIMDs 3/2/2/2); the
// radix-16 pass in the middle (two butterflies per thread), the radix-31 pass last: its 31 outputs go straight into the
// power sums without an LDS scatter (order [33,31,16]: 0.52 ms per 32-PRN dwell; [33,16,31]: 0.47 before the prime-factor form)
using Plan16368 = Plan<16368, 768, 33, 16, 31>;
// 8184 = 8 * 3 * 11 * 31 (pairwise coprime): half of 16368 — in LDS (two workgroups per CU: 65.5 KB each) and the base of 3 x / 5 x 8184.
// Round 6's plan search: [11, 31, 8, 3] on 768 lanes — no scratch memory in LDS (352 us at the 32 x 41 x 10 grid against 370 for [24, 11, 31]
// on 384 lanes, which spilled 26 registers inside the integration loop) and one pass-0 butterfly per lane, so the composites run it too
// (3 x 8184 0.996 -> 0.705 ms, 5 x 8184 2.59 -> 1.79 ms at 32 x 41 x 4)
#ifndef GM_PLAN_8184
#define GM_PLAN_8184 Plan<8184, 768, 11, 31, 8, 3>
#endif
using Plan8184 = GM_PLAN_8184;
using Plan4096 = Plan<4096, 256, 16, 16, 16>;
using Plan2048 = Plan<2048, 256, 8, 16, 16>;
using Plan1024 = Plan<1024, 128, 8, 8, 16>;
using Plan4000 = Plan<4000, 256, 25, 16, 10>;
using Plan10000 = Plan<10000, 1024, 10, 10, 10, 10>;   // (plan search, round 6: 350 -> 321 us at the 32 x 41 x 10 grid against [25, 20, 20] on 512 lanes)
using Plan12000 = Plan<12000, 768, 20, 25, 24>;        // (plan search, round 6: 451 -> 389 us against [25, 3, 10, 16] on 512 lanes)
using Plan16000 = Plan<16000, 1024, 25, 20, 32>;
using Plan2000 = Plan<2000, 128, 25, 10, 8>;      // 2 Msps
#ifndef GM_PLAN_5000
#define GM_PLAN_5000 Plan<5000, 256, 25, 20, 10>
#endif
using Plan5000 = GM_PLAN_5000;     // 5 Msps (plan search, round 6: 134 -> 121 us against [25, 25, 8])
#ifndef GM_PLAN_6000
#define GM_PLAN_6000 Plan<6000, 256, 25, 24, 10>
#endif
using Plan6000 = GM_PLAN_6000;     // 6 Msps (round 6 plan search, tools/corr_lab/plan_search.py: 162.6 -> 137.3 us at the 32 x 41 x 10 grid against [25, 15, 16] on 512 lanes)
// 8192: the registered plan is what the forward transforms and the composite path (3 x, 5 x 8192) run — four passes of <= 16 values, which
// leave the composite kernel 2 - 6 scratch instructions and 12 - 17 % less time than [16, 32, 16] (63 - 121 instructions) —; the in-LDS
// inverse keeps [16, 32, 16] (acq_corr_plans.h CorrPlanOf<Plan8192>: 195 us at the 32 x 41 x 10 grid against 204 - 209 for any four-pass plan)
using Plan8192 = Plan<8192, 512, 16, 8, 8, 8>;    // 8.192 Msps
using Plan15000 = Plan<15000, 768, 25, 25, 24>;   // 15 Msps (768 lanes = 3 waves per SIMD, 170 registers: at 1024 lanes the same radices spilled 39 inside the loop, 561 -> 519 us)
using Plan16384 = Plan<16384, 1024, 16, 8, 8, 16>; // 16.384 Msps (four passes of <= 16 values per lane under the 128-register cap: [32, 32, 16] was 830 us against 498 at configs[1]'s grid)
using Plan512 = Plan<512, 64, 8, 8, 8>;           // factors of the long fine-Doppler FFT (2^16 .. 2^19) at low sample rates
using Plan256 = Plan<256, 64, 16, 16>;
#ifndef GM_NO_PLAN_ROT
template <> struct PlanRot<Plan8000> { static constexpr int rot(int s) { return s == 1 ? 64 : 0; } };   // passes run on waves 0-4 / 1-7 / 0-7
#endif
}  // namespace gm

// correlation plans (acq_device.h CorrPlanOf): 8000 = 125 * 64 as the hybrid [5*4, 25, 16] with constant twiddles per wave
#include "acq_corr_plans.h"

#ifndef GM_FOR_EACH_PLAN   // tools/corr_lab restricts the list to one plan for fast experimental builds
#define GM_FOR_EACH_PLAN(X) \
    X(gm::Plan8000) X(gm::Plan16368) X(gm::Plan8184) X(gm::Plan4096) X(gm::Plan2048) X(gm::Plan1024) \
    X(gm::Plan4000) X(gm::Plan10000) X(gm::Plan12000) X(gm::Plan16000) \
    X(gm::Plan2000) X(gm::Plan5000) X(gm::Plan6000) X(gm::Plan8192) X(gm::Plan15000) X(gm::Plan16384) \
    X(gm::Plan512) X(gm::Plan256)
#endif
