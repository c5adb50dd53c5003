// tools/corr_lab/comp_ws_stamps.hip — phase stamps of the wave-specialised composite kernel (acq_comp_ws.h) at the configs[3] Galileo
// geometry (36 codes x 41 bins x N = 2 x 16000, 2 periods), workgroups 0 / 504 / 1008: wave 0 (middle + last pass), wave 8 (middle pass + a
// pass-0 butterfly), wave 10 (pass 0 only); workgroup durations; and the kernel's duration by HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize comp_ws_stamps.hip -o comp_ws_stamps
#include "../../gnss-sdr-rs_amd/csrc/acq_composite.hip"
namespace gm { int diag_int(const char* name, int dflt) { const char* v = getenv(name); return (v && *v) ? atoi(v) : dflt; } }
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
int main() {
    using namespace gm;
    using CP = CorrPlan16000;
    constexpr uint32_t Q = 2;
    const int P = 36, D = 41, M = 2, N = 32000, S = Q * M;
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<cf> hx(size_t(D) * M * N), hc(size_t(P) * Q * N);
    for (auto& v : hx) v = cf_make(nd(rng), nd(rng));
    for (auto& v : hc) v = cf_make(nd(rng), nd(rng));
    cf *dx, *dc; float* met; uint32_t* wl; long long* st;
    hipMalloc(&dx, hx.size() * 8); hipMalloc(&dc, hc.size() * 8); hipMalloc(&met, size_t(3) * P * D * 4); hipMalloc(&wl, P * 4);
    hipMalloc(&st, 4 * S * 3 * 8 * 8); hipMemset(st, 0, 4 * S * 3 * 8 * 8);
    hipMemcpy(dx, hx.data(), hx.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dc, hc.data(), hc.size() * 8, hipMemcpyHostToDevice);
    std::vector<uint32_t> hwl(P); for (int i = 0; i < P; ++i) hwl[i] = i;
    hipMemcpy(wl, hwl.data(), P * 4, hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(g_comp_ws_stamps), &st, sizeof(st));
    long long* wg; const int NWG = 8 * (((P + 3) / 4) * (((P * D + 7) / 8 + P - 2) / P + 1) * 4);
    hipMalloc(&wg, NWG * 16); hipMemset(wg, 0, NWG * 16);
    hipMemcpyToSymbol(HIP_SYMBOL(g_comp_ws_wg), &wg, sizeof(wg));
    const int items = P * D, share = (items + 7) / 8, cb = 4, rows_max = (share + P - 2) / P + 1, slots = ((P + cb - 1) / cb) * rows_max * cb;
    auto launch_st = [&]() { hipLaunchKernelGGL((comp_corr_ws_kernel<CP, Q, true>), dim3(8 * slots), dim3(1024), 0, 0, dx, dc, met, reinterpret_cast<uint32_t*>(met) + P * D, met + 2 * P * D, wl, P, D, M, cb, rows_max, static_cast<float*>(nullptr)); };
    auto launch = [&]() { hipLaunchKernelGGL((comp_corr_ws_kernel<CP, Q, false>), dim3(8 * slots), dim3(1024), 0, 0, dx, dc, met, reinterpret_cast<uint32_t*>(met) + P * D, met + 2 * P * D, wl, P, D, M, cb, rows_max, static_cast<float*>(nullptr)); };
    for (int rep = 0; rep < 3; ++rep) launch_st();
    hipDeviceSynchronize();
    std::vector<long long> hall(4 * S * 3 * 8);
    hipMemcpy(hall.data(), st, hall.size() * 8, hipMemcpyDeviceToHost);
    for (int w = 0; w < 3; ++w)
    for (int s = 0; s < S; ++s) {
        const long long* h = &hall[size_t(w) * S * 24];
        if (s == 0) printf("---- workgroup %d\n", w * 504);
        const long long* b = &h[(s * 3 + 0) * 8]; const long long* c = &h[(s * 3 + 1) * 8]; const long long* a = &h[(s * 3 + 2) * 8];
        const long long* an = s + 1 < S ? &h[((s + 1) * 3 + 2) * 8] : a;
        printf("s=%d pass-0 wave 8: loads + first halves=%lld wait B1=%lld second halves + scatter=%lld wait B2=%lld | period %lld\n", s, a[1] - a[0], a[2] - a[1], a[3] - a[2],
               an[0] - a[3], an[0] - a[0]);
        printf("     wave 0: wait B1=%lld wait B2=%lld middle (two slots)=%lld group barrier=%lld last pass=%lld | total %lld\n", b[2] - b[0], b[4] - b[2], b[5] - b[4], b[6] - b[5], b[7] - b[6], b[7] - b[0]);
        printf("     wave 7: wait B1=%lld wait B2=%lld middle=%lld group barrier=%lld last pass=%lld\n", c[2] - c[0], c[4] - c[2], c[5] - c[4], c[6] - c[5], c[7] - c[6]);
    }
    {   // workgroup spans of the last stamped launch: tick rate against the event time of one stamped launch, and durations by start order
        hipEvent_t a0, a1; hipEventCreate(&a0); hipEventCreate(&a1);
        hipMemset(wg, 0, NWG * 16);
        hipEventRecord(a0); launch_st(); hipEventRecord(a1); hipEventSynchronize(a1);
        float ms1; hipEventElapsedTime(&ms1, a0, a1);
        std::vector<long long> hw(NWG * 2);
        hipMemcpy(hw.data(), wg, NWG * 16, hipMemcpyDeviceToHost);
        std::vector<std::pair<long long, long long>> v;
        for (int i = 0; i < NWG; ++i) if (hw[2 * i + 1]) v.push_back({hw[2 * i], hw[2 * i + 1] - hw[2 * i]});
        std::sort(v.begin(), v.end());
        long long t0 = v.front().first, t1 = 0; for (auto& x : v) t1 = std::max(t1, x.first + x.second);
        printf("stamped launch: %.1f us, %zu workgroups with an item, span %lld ticks -> %.2f ticks per ns\n", ms1 * 1000, v.size(), t1 - t0, double(t1 - t0) / (ms1 * 1e6));
        for (size_t k = 0; k < v.size(); k += v.size() / 12) printf("  workgroup #%zu by start: start +%lld ticks, duration %lld ticks\n", k, v[k].first - t0, v[k].second);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("kernel: %.1f us per launch (%d workgroups)\n", ms * 1000 / 20, 8 * slots);
    return 0;
}
