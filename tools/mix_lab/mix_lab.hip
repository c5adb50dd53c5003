// tools/mix_lab/mix_lab.hip — timing laboratory for stage F (acq_mix_fft_kernel) at BASELINE configs[1] geometry (41 bins x 10
// integrations of N = 8000, complex int8; arguments: bins, integrations): the product kernel source included as is, one plan
// (-DLAB_PLAN=...), timed beside an EMPTY launch of the same shape.  Timing only.  Not product code, not a test.
#ifndef LAB_PLAN
#define LAB_PLAN gm::Plan8000
#endif
#define GM_FOR_EACH_PLAN(X) X(LAB_PLAN)
#include "../../gnss-sdr-rs_amd/csrc/acq_kernels.hip"
#include <cstdio>
#include <cstdlib>
namespace gm { int diag_int(const char* name, int dflt) { const char* v = getenv(name); return (v && *v) ? atoi(v) : dflt; } }
#include <vector>
#include <random>
#include <algorithm>

__global__ void lab_empty_kernel() {}

int main(int argc, char** argv) {
    using namespace gm;
    const int D = argc > 1 ? atoi(argv[1]) : 41, M = argc > 2 ? atoi(argv[2]) : 10;
    const PlanOps* pl = &g_plans[0];
    const int N = pl->n;
    std::mt19937 rng(1);
    std::vector<int8_t> hs(size_t(M) * N * 2);
    for (auto& v : hs) v = int8_t(int(rng() % 255) - 127);
    std::vector<cf> ht(size_t(D) * N), htw(pl->tw_total_mix + 1);
    for (size_t i = 0; i < ht.size(); ++i) { const float a = 1e-3f * float(i % 6283); ht[i] = cf_make(cosf(a), sinf(a)); }
    pl->fill_tw_mix(htw.data(), false);
    const int no = pl->fill_order(nullptr);
    std::vector<uint16_t> ho(no > 0 ? no : 1);
    if (no > 0) pl->fill_order(ho.data());
    int8_t* ds; cf *dt, *dtw, *dsp; uint16_t* dord;
    hipMalloc(&ds, hs.size()); hipMalloc(&dt, ht.size() * 8); hipMalloc(&dtw, htw.size() * 8);
    hipMalloc(&dsp, size_t(D) * M * N * 8); hipMalloc(&dord, ho.size() * 2);
    hipMemcpy(ds, hs.data(), hs.size(), hipMemcpyHostToDevice);
    hipMemcpy(dt, ht.data(), ht.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dtw, htw.data(), htw.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dord, ho.data(), ho.size() * 2, hipMemcpyHostToDevice);
    auto go = [&]() { pl->mix_fft(0, ds, GM_FMT_I8_IQ, dt, dtw, dsp, D, M, nullptr, no > 0 ? dord : nullptr, nullptr); };
    for (int i = 0; i < 3; ++i) go();
    hipDeviceSynchronize();
    std::vector<float> t;
    for (int rep = 0; rep < 25; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0); go(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    {   // what an event-timed launch of this shape reads with nothing in it
        std::vector<float> te;
        for (int rep = 0; rep < 25; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0); hipLaunchKernelGGL(lab_empty_kernel, dim3(D * M), dim3(pl->threads), 0, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); te.push_back(ms * 1e3f);
        }
        std::sort(te.begin(), te.end());
        printf("empty kernel %d x %d: median %.1f us\n", D * M, pl->threads, te[te.size() / 2]);
    }
    printf("stage F N=%d D=%d M=%d: median %.1f us, min %.1f us per launch\n", N, D, M, t[t.size() / 2], t[0]);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
