// tools/corr_lab/ws31_stamps.hip — phase stamps of the wave-specialised N = 16368 kernel (acq_corr_ws31.h), workgroup 0, waves 0 and 8
#define GM_FOR_EACH_PLAN(X) X(gm::Plan16368)
#include "../../gnss-sdr-rs_amd/csrc/acq_kernels.hip"
namespace gm { int diag_int(const char* name, int dflt) { const char* v = getenv(name); return (v && *v) ? atoi(v) : dflt; } }
#include <cstdio>
#include <vector>
#include <random>
int main() {
    using namespace gm;
    const int P = 32, D = 29, M = 10, N = 16368;
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<cf> hx(size_t(D) * M * N), hc(size_t(P) * N);
    for (auto& v : hx) v = cf_make(nd(rng), nd(rng));
    for (auto& v : hc) v = cf_make(nd(rng), nd(rng));
    cf *dx, *dc; float* met; uint32_t* wl; long long* st;
    hipMalloc(&dx, hx.size() * 8); hipMalloc(&dc, hc.size() * 8); hipMalloc(&met, size_t(3) * P * D * 4); hipMalloc(&wl, P * 4);
    hipMalloc(&st, M * 3 * 8 * 8); hipMemset(st, 0, M * 3 * 8 * 8);
    hipMemcpy(dx, hx.data(), hx.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dc, hc.data(), hc.size() * 8, hipMemcpyHostToDevice);
    std::vector<uint32_t> hwl(P); for (int i = 0; i < P; ++i) hwl[i] = i;
    hipMemcpy(wl, hwl.data(), P * 4, hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(g_ws31_stamps), &st, sizeof(st));
    const int share = (P * D + 7) / 8;
    for (int rep = 0; rep < 3; ++rep)
        hipLaunchKernelGGL((acq_corr_ws31_kernel<Plan16368, false, true>), dim3(8 * share), dim3(1024), 0, 0, dx, dc, met, reinterpret_cast<uint32_t*>(met) + P * D,
                           met + 2 * P * D, wl, P, D, M, 0, share, 1, 0, nullptr, nullptr, 0);
    hipDeviceSynchronize();
    std::vector<long long> h(M * 3 * 8);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    const char* n0[] = {"loads+products+r33a", "wait B1", "r33b+scatter", "wait B2", "middle(gather,B3,r16,scatter,B4)"};
    for (int m = 1; m < M; ++m) {
        const long long* a = &h[(m * 3 + 0) * 8]; const long long* b = &h[(m * 3 + 1) * 8]; const long long* c = &h[(m * 3 + 2) * 8];
        printf("m=%d pass-0 wave: %s=%lld %s=%lld %s=%lld %s=%lld %s=%lld | total %lld\n", m, n0[0], a[1] - a[0], n0[1], a[2] - a[1], n0[2], a[3] - a[2], n0[3], a[4] - a[3], n0[4], a[5] - a[4], a[5] - a[0]);
        printf("     matrix wave: radix31(mfma)=%lld wait B1=%lld wait B2=%lld middle=%lld | total %lld\n", b[1] - b[0], b[2] - b[1], b[4] - b[2], b[5] - b[4], b[5] - b[0]);
        printf("     matrix wave batches: %lld %lld %lld %lld\n", c[0] - b[0], c[1] - c[0], c[2] - c[1], c[3] - c[2]);
    }
    return 0;
}
