// tools/corr_lab/corr_lab.hip — timing laboratory for stage C at BASELINE configs[1] geometry (32 workers x 41 bins x N = 8000 x 10
// integrations; arguments: workers, bins, integrations) on random data: the PRODUCT kernel source is included as is, restricted to one
// plan (-DLAB_PLAN=...), so that a variant (a macro on the command line, an older source through -DLAB_SRC, a lab header such as
// mfma16_8000/ or wsh_8000/) can be timed beside the shipped kernel on one box.  The timing-ablation macros of rounds 2-3 (no loads,
// no barriers, no twiddles) are no longer in the product sources: their results are in profiles/r03_lab_ablations.txt, and they can be
// repeated with -DLAB_SRC pointing at commit 5af55ed's kernels (tools/README.md).  Prints the median / minimum launch time and a
// checksum of the maxima (equal checksums = same results to seven digits).  Not product code, not a test.
#ifndef LAB_PLAN
#define LAB_PLAN gm::Plan8000
#endif
#define GM_FOR_EACH_PLAN(X) X(LAB_PLAN)
#ifdef LAB_WAVES_PER_EU      // the same plan under a register cap (e.g. 4 waves per SIMD = 128 VGPRs: what two 512-lane workgroups per CU would leave each lane)
#include "../../gnss-sdr-rs_amd/csrc/fft_core.h"
struct LabCapped : LAB_BASE_PLAN { static constexpr int WAVES_PER_EU = LAB_WAVES_PER_EU; };
#endif
#ifndef LAB_REF_MUL
#define LAB_REF_MUL 0      // 1: gm_acq_cfg.reference_products (num-complex rounding of the products)
#endif
#ifndef LAB_SRC
#define LAB_SRC "../../gnss-sdr-rs_amd/csrc/acq_kernels.hip"
#endif
#include LAB_SRC
// the laboratory reads its overrides straight from the environment (the product library's gate is gm_api.hip's diag_int)
namespace gm { int diag_int(const char* name, int dflt) { const char* v = getenv(name); return (v && *v) ? atoi(v) : dflt; } }
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>

int main(int argc, char** argv) {
    using namespace gm;
    const int P = argc > 1 ? atoi(argv[1]) : 32, D = argc > 2 ? atoi(argv[2]) : 41, M = argc > 3 ? atoi(argv[3]) : 10;
    const PlanOps* pl = &g_plans[0];
    const int N = pl->n;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<cf> hx(size_t(D) * M * N), hc(size_t(P) * N), htw(pl->tw_total + 1);
    for (auto& v : hx) v = cf_make(nd(rng), nd(rng));
    for (auto& v : hc) v = cf_make(nd(rng), nd(rng));
    pl->fill_tw(htw.data(), true);
    cf *dx, *dc, *dtw; float* met; uint32_t* wl; float* scratch; uint32_t* counter;
    hipMalloc(&dx, hx.size() * 8); hipMalloc(&dc, hc.size() * 8); hipMalloc(&dtw, htw.size() * 8);
    hipMalloc(&met, size_t(3) * P * D * 4); hipMalloc(&wl, P * 4);
    hipMalloc(&scratch, size_t(GM_CORR_SPLIT_MAX_SLABS) * std::max(pl->split_slab, 1) * 4); hipMalloc(&counter, GM_CORR_SPLIT_MAX_ITEMS * 4);
    hipMemset(counter, 0, GM_CORR_SPLIT_MAX_ITEMS * 4);
    hipMemcpy(dx, hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dc, hc.data(), hc.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dtw, htw.data(), htw.size() * 8, hipMemcpyHostToDevice);
    std::vector<uint32_t> hwl(P); for (int i = 0; i < P; ++i) hwl[i] = i;
    hipMemcpy(wl, hwl.data(), P * 4, hipMemcpyHostToDevice);
    auto go = [&]() { pl->corr(0, dx, dc, dtw, met, reinterpret_cast<uint32_t*>(met) + P * D, met + 2 * P * D, wl, P, D, M, scratch, GM_CORR_SPLIT_MAX_SLABS, counter, 0, 0, LAB_REF_MUL); };
    for (int i = 0; i < 3; ++i) go();
    hipDeviceSynchronize();
    std::vector<float> t;
    for (int rep = 0; rep < 15; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0); go(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    std::vector<float> hm(size_t(3) * P * D);
    hipMemcpy(hm.data(), met, hm.size() * 4, hipMemcpyDeviceToHost);
    double chk = 0; for (int i = 0; i < P * D; ++i) chk += hm[i];
    printf("N=%d P=%d D=%d M=%d: median %.1f us, min %.1f us per launch (checksum of maxima %.6e)\n", N, P, D, M, t[t.size() / 2], t[0], chk);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
