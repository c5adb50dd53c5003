"""Generates tools/ubench/valu_forms.hip: issue cost of individual gfx950 VALU instruction FORMS, in cycles per
wave64 instruction per SIMD (s_memtime inside the kernel: independent of the clock the chip holds), at 1 / 2 / 4 waves
per SIMD, every CU busy.  Each variant is one inline-asm block of 64 instructions over explicit VGPRs (8 independent
destination chains so that dependent-issue latency is not what is measured), repeated ITER times.

Why: the correlation kernel is VALU-issue bound; round 1 measured "1.85 ns per wave-instruction whatever the occupancy"
on compiler-generated loops whose operand forms were not controlled.  This separates the effects: opcode (fma / mul / add /
mov / packed), encoding (VOP2 / VOP3 / literal), operand kinds (VGPR / SGPR / inline constant) and VGPR bank pattern."""
import os

HERE = os.path.dirname(os.path.abspath(__file__))

# (name, template); {d} = destination chain register index 0..7 mapped to v[8+d] ; sources from v[16..31], s[..]
V = []
def add(name, fn):
    V.append((name, fn))

add("v_fma_f32 d,a,b,d   (3 vgpr, banks spread)", lambda d: f"v_fma_f32 v{8+d}, v{16+((d+1)%8)}, v{24+((d+2)%8)}, v{8+d}")
add("v_fma_f32 d,a,b,d   (3 vgpr, same bank)", lambda d: f"v_fma_f32 v{8+d}, v{16+(d%8)//4*4+ (d%4)}, v{24+(d%4)}, v{8+d}" if False else f"v_fma_f32 v{8+d}, v{16+(d%4)+4*((d//4)%2)}, v{24+(d%4)+4*((d//4)%2)}, v{8+d}")
add("v_fmac_f32 d,a,b     (VOP2)", lambda d: f"v_fmac_f32 v{8+d}, v{16+((d+1)%8)}, v{24+((d+2)%8)}")
add("v_add_f32 d,a,d      (VOP2)", lambda d: f"v_add_f32 v{8+d}, v{16+((d+1)%8)}, v{8+d}")
add("v_sub_f32 d,a,d      (VOP2)", lambda d: f"v_sub_f32 v{8+d}, v{16+((d+1)%8)}, v{8+d}")
add("v_mul_f32 d,a,d      (VOP2)", lambda d: f"v_mul_f32 v{8+d}, v{16+((d+1)%8)}, v{8+d}")
add("v_add_f32 d,a,b      (VOP2, no self dep)", lambda d: f"v_add_f32 v{8+d}, v{16+((d+1)%8)}, v{24+((d+2)%8)}")
add("v_mul_f32 d,a,b      (VOP2, no self dep)", lambda d: f"v_mul_f32 v{8+d}, v{16+((d+1)%8)}, v{24+((d+2)%8)}")
add("v_fma_f32 d,a,s,d    (sgpr operand)", lambda d: f"v_fma_f32 v{8+d}, v{16+((d+1)%8)}, s20, v{8+d}")
add("v_fma_f32 d,a,0.5,d  (inline const)", lambda d: f"v_fma_f32 v{8+d}, v{16+((d+1)%8)}, 0.5, v{8+d}")
add("v_fmaak_f32 d,a,b,K  (literal)", lambda d: f"v_fmaak_f32 v{8+d}, v{16+((d+1)%8)}, v{24+((d+2)%8)}, 0x3f800123")
add("v_mul_f32 d,K,a      (VOP2 literal)", lambda d: f"v_mul_f32 v{8+d}, 0x3f800123, v{16+((d+1)%8)}")
add("v_fma_f32 d,-a,b,d   (neg modifier)", lambda d: f"v_fma_f32 v{8+d}, -v{16+((d+1)%8)}, v{24+((d+2)%8)}, v{8+d}")
add("v_mov_b32 d,a", lambda d: f"v_mov_b32 v{8+d}, v{16+((d+1)%8)}")
add("v_pk_fma_f32 D,A,B,D (3 vgpr pairs)", lambda d: f"v_pk_fma_f32 v[{32+2*d}:{33+2*d}], v[{48+2*((d+1)%8)}:{49+2*((d+1)%8)}], v[{64+2*((d+2)%8)}:{65+2*((d+2)%8)}], v[{32+2*d}:{33+2*d}]")
add("v_pk_mul_f32 D,A,D", lambda d: f"v_pk_mul_f32 v[{32+2*d}:{33+2*d}], v[{48+2*((d+1)%8)}:{49+2*((d+1)%8)}], v[{32+2*d}:{33+2*d}]")
add("v_pk_add_f32 D,A,D", lambda d: f"v_pk_add_f32 v[{32+2*d}:{33+2*d}], v[{48+2*((d+1)%8)}:{49+2*((d+1)%8)}], v[{32+2*d}:{33+2*d}]")
add("mix fma,mul,add,sub  (butterfly-like)", lambda d: [f"v_fma_f32 v{8+d}, v{16+((d+1)%8)}, v{24+((d+2)%8)}, v{8+d}", f"v_mul_f32 v{8+d}, v{16+((d+1)%8)}, v{8+d}", f"v_add_f32 v{8+d}, v{16+((d+1)%8)}, v{8+d}", f"v_sub_f32 v{8+d}, v{24+((d+1)%8)}, v{8+d}"][d % 4])
add("v_add_f32 + v_fma_f32 alternating", lambda d: [f"v_add_f32 v{8+d}, v{16+((d+1)%8)}, v{8+d}", f"v_fma_f32 v{8+d}, v{16+((d+1)%8)}, v{24+((d+2)%8)}, v{8+d}"][d % 2])
add("v_fma_f64 D,A,B,D", lambda d: f"v_fma_f64 v[{32+2*d}:{33+2*d}], v[{48+2*((d+1)%8)}:{49+2*((d+1)%8)}], v[{64+2*((d+2)%8)}:{65+2*((d+2)%8)}], v[{32+2*d}:{33+2*d}]")
add("v_cndmask_b32 d,a,b  (vcc)", lambda d: f"v_cndmask_b32 v{8+d}, v{16+((d+1)%8)}, v{24+((d+2)%8)}, vcc")
add("v_cvt_f32_i32 d,a", lambda d: f"v_cvt_f32_i32 v{8+d}, v{16+((d+1)%8)}")
add("v_floor_f32 d,a", lambda d: f"v_floor_f32 v{8+d}, v{16+((d+1)%8)}")
add("v_mul_f64 D,A,D", lambda d: f"v_mul_f64 v[{32+2*d}:{33+2*d}], v[{48+2*((d+1)%8)}:{49+2*((d+1)%8)}], v[{32+2*d}:{33+2*d}]")
add("v_rndne_f64 D,A", lambda d: f"v_rndne_f64 v[{32+2*d}:{33+2*d}], v[{48+2*((d+1)%8)}:{49+2*((d+1)%8)}]")
add("v_cvt_f64_f32 D,a", lambda d: f"v_cvt_f64_f32 v[{32+2*d}:{33+2*d}], v{16+((d+1)%8)}")

add("v_cndmask_b32 d,a,b,s[22:23] (VOP3 sgpr mask)", lambda d: f"v_cndmask_b32 v{8+d}, v{16+((d+1)%8)}, v{24+((d+2)%8)}, s[22:23]")
add("v_cmp_gt_f32 vcc,a,b (VOPC)", lambda d: f"v_cmp_gt_f32 vcc, v{16+((d+1)%8)}, v{24+((d+2)%8)}")
add("v_cmp_gt_f32 s[24:25],a,b (VOP3)", lambda d: f"v_cmp_gt_f32 s[24:25], v{16+((d+1)%8)}, v{24+((d+2)%8)}")
add("v_cmp + v_cndmask pairs (vcc)", lambda d: [f"v_cmp_gt_f32 vcc, v{16+((d+1)%8)}, v{24+((d+2)%8)}", f"v_cndmask_b32 v{8+d}, v{16+((d+1)%8)}, v{24+((d+2)%8)}, vcc"][d % 2])
add("v_add_u32 d,a,d", lambda d: f"v_add_u32 v{8+d}, v{16+((d+1)%8)}, v{8+d}")
add("v_and_b32 d,a,d", lambda d: f"v_and_b32 v{8+d}, v{16+((d+1)%8)}, v{8+d}")
add("v_lshl_add_u64 D,A,3,D", lambda d: f"v_lshl_add_u64 v[{32+2*d}:{33+2*d}], v[{48+2*((d+1)%8)}:{49+2*((d+1)%8)}], 3, v[{32+2*d}:{33+2*d}]")
add("v_cvt_i32_f32 d,a", lambda d: f"v_cvt_i32_f32 v{8+d}, v{16+((d+1)%8)}")
add("v_cvt_f32_u32 d,a", lambda d: f"v_cvt_f32_u32 v{8+d}, v{16+((d+1)%8)}")
add("v_fract_f32 d,a", lambda d: f"v_fract_f32 v{8+d}, v{16+((d+1)%8)}")
add("v_trunc_f32 d,a", lambda d: f"v_trunc_f32 v{8+d}, v{16+((d+1)%8)}")
add("v_rndne_f32 d,a", lambda d: f"v_rndne_f32 v{8+d}, v{16+((d+1)%8)}")
add("v_max_f32 d,a,d", lambda d: f"v_max_f32 v{8+d}, v{16+((d+1)%8)}, v{8+d}")
add("v_min_i32 d,a,d", lambda d: f"v_min_i32 v{8+d}, v{16+((d+1)%8)}, v{8+d}")
add("v_med3_f32 d,a,b,d", lambda d: f"v_med3_f32 v{8+d}, v{16+((d+1)%8)}, v{24+((d+2)%8)}, v{8+d}")
add("v_bfe_i32 d,a,4,8", lambda d: f"v_bfe_i32 v{8+d}, v{16+((d+1)%8)}, 4, 8")
add("v_mul_lo_u32 d,a,b", lambda d: f"v_mul_lo_u32 v{8+d}, v{16+((d+1)%8)}, v{24+((d+2)%8)}")
add("v_mad_u32_u24 d,a,b,d", lambda d: f"v_mad_u32_u24 v{8+d}, v{16+((d+1)%8)}, v{24+((d+2)%8)}, v{8+d}")
add("v_xor_b32 d,a,d", lambda d: f"v_xor_b32 v{8+d}, v{16+((d+1)%8)}, v{8+d}")
add("v_sin_f32 d,a", lambda d: f"v_sin_f32 v{8+d}, v{16+((d+1)%8)}")
add("v_rcp_f32 d,a", lambda d: f"v_rcp_f32 v{8+d}, v{16+((d+1)%8)}")
add("v_sqrt_f32 d,a", lambda d: f"v_sqrt_f32 v{8+d}, v{16+((d+1)%8)}")
add("v_perm_b32 d,a,b,d", lambda d: f"v_perm_b32 v{8+d}, v{16+((d+1)%8)}, v{24+((d+2)%8)}, v{8+d}")
add("v_readlane_b32 s26,a,5", lambda d: f"v_readlane_b32 s26, v{16+((d+1)%8)}, 5")
add("v_mov_b32_dpp row_shr:1", lambda d: f"v_mov_b32_dpp v{8+d}, v{16+((d+1)%8)} row_shr:1 row_mask:0xf bank_mask:0xf")
add("v_add_f32_dpp row_shr:1", lambda d: f"v_add_f32_dpp v{8+d}, v{16+((d+1)%8)}, v{8+d} row_shr:1 row_mask:0xf bank_mask:0xf")
# dependent chains (one destination register: the latency of back-to-back dependent instructions of ONE wave)
add("DEP v_mul d,a,d ; v_add d,b,d (vgpr)", lambda d: [f"v_mul_f32 v8, v16, v8", f"v_add_f32 v8, v24, v8"][d % 2])
add("DEP v_mul d,s,d ; v_add d,b,d (sgpr con)", lambda d: [f"v_mul_f32 v8, s20, v8", f"v_add_f32 v8, v24, v8"][d % 2])
add("DEP v_fma d,a,b,d", lambda d: f"v_fma_f32 v8, v16, v24, v8")
add("DEP v_add d,a,d", lambda d: f"v_add_f32 v8, v16, v8")
add("DEP 2 chains interleaved mul/add", lambda d: [f"v_mul_f32 v8, v16, v8", f"v_mul_f32 v9, v16, v9", f"v_add_f32 v8, v24, v8", f"v_add_f32 v9, v24, v9"][d % 4])
UNROLL = 64
clob = ", ".join(f'"v{i}"' for i in range(8, 80))

src = ['// GENERATED by tools/ubench/gen_valu_forms.py — do not edit.', '#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <vector>', '#include <algorithm>',
       'template <int KIND> __global__ void k(long long* cyc, int iters, float seed) {',
       '    // registers v8..v79 are owned by the asm blocks (declared as clobbers); values stay finite: every source is 1.0 or 0.0',
       '    asm volatile("s_mov_b32 s20, 1.0\\n\\ts_mov_b64 s[22:23], -1" ::: "s20", "s22", "s23");',
       '    for (int r = 8; r < 80; ++r) {}',
       ]
init = "\\n\\t".join([f"v_mov_b32 v{i}, 0" for i in range(8, 16)] + [f"v_mov_b32 v{i}, 1.0" for i in range(16, 32)] +
                     [f"v_mov_b32 v{i}, 0" for i in range(32, 48)] + [f"v_mov_b32 v{i}, 0" for i in range(48, 80)])
src.append(f'    asm volatile("{init}" ::: {clob});')
src.append('    unsigned long long t0, t1;')
src.append('    asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier\\n\\ts_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");')
src.append('    for (int i = 0; i < iters; ++i) {')
for ki, (name, fn) in enumerate(V):
    body = "\\n\\t".join(fn(j % 8) for j in range(UNROLL))
    src.append(f'        if constexpr (KIND == {ki}) asm volatile("{body}" ::: {clob}, "vcc", "s20", "s22", "s23", "s24", "s25", "s26");')
src.append('    }')
src.append('    asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");')
src.append('    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = (long long)(t1 - t0);')
src.append('}')
src.append('static const char* NAMES[] = {' + ", ".join('"' + n + '"' for n, _ in V) + '};')
src.append(f'constexpr int NK = {len(V)}, UNROLL = {UNROLL};')
src.append('template <int K> void launch(int blocks, int threads, long long* cyc, int iters) { hipLaunchKernelGGL(k<K>, dim3(blocks), dim3(threads), 0, 0, cyc, iters, 1.0f); }')
src.append('typedef void (*LaunchFn)(int, int, long long*, int);')
src.append('static const LaunchFn L[] = {' + ", ".join(f"launch<{i}>" for i in range(len(V))) + '};')
src.append(r'''
int main() {
    long long* cyc; hipMalloc(&cyc, sizeof(long long) * 256 * 16);
    const int iters = 400;
    printf("cycles per wave64 instruction PER SIMD (s_memtime ticks; median over waves); 256 blocks = every CU busy\n");
    printf("%-44s %8s %8s %8s   %s\n", "form", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "ns per instr per SIMD at 4 w (HIP events)");
    for (int kind = 0; kind < NK; ++kind) {
        double res[3]; double ns4 = 0;
        int wi = 0;
        for (int threads : {256, 512, 1024}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            L[kind](256, threads, cyc, iters); hipDeviceSynchronize();
            hipEventRecord(e0, 0); L[kind](256, threads, cyc, iters); hipEventRecord(e1, 0); hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const int nw = 256 * threads / 64;
            std::vector<long long> h(nw); hipMemcpy(h.data(), cyc, sizeof(long long) * nw, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            const double per_wave = double(h[nw / 2]) / (double(iters) * UNROLL);
            res[wi++] = per_wave / (threads / 256.0);
            if (threads == 1024) ns4 = ms * 1e6 / (double(iters) * UNROLL * 4);
        }
        printf("%-44s %8.2f %8.2f %8.2f   %.3f\n", NAMES[kind], res[0], res[1], res[2], ns4);
    }
    return 0;
}
''')
open(os.path.join(HERE, "valu_forms.hip"), "w").write("\n".join(src) + "\n")
print("wrote valu_forms.hip with", len(V), "variants")
