// acq_kernels.hip — acquisition kernels for gfx950 (CDNA4, wave64).
//
// Replaces the per-PRN loop of AcquisitionWorker::search_satellite
// (src/acquisition/do_acquisition.rs:158-226) with two kernels:
//
//   stage F  acq_mix_fft_kernel  : one workgroup per (Doppler bin d, ms block m):
//            out = s * table[d]  (apply_doppler_shift, doppler_shift.rs:25-58, same rounding sequence)
//            fused into pass 0 of the forward FFT (:182); spectrum[d][m][k] written coalesced.
//            The reference recomputes this for every PRN (32x redundant); it does not depend on the PRN.
//   stage C  acq_corr_kernel     : one workgroup per (worker p, Doppler bin d), looping over the M
//            integrations: X[d][m][k] * conj(C[p][k]) (:184-186) fused into pass 0 of the inverse
//            FFT (:188); |.|^2 accumulated in registers across m (:190-192); finally the first strict
//            argmax / max (:195-202) and the plane sum (:229-235) by wavefront shuffles.
//            conj(C[p]) stays in registers across the m loop, the transform lives in one LDS buffer.
//
// HBM/L2 layout: spectra [D][M][N] c32 contiguous in k (8 B per lane, 512 B per wave-instruction);
// code spectra [P][N].  Algorithmic bytes per dwell: D*M*N*(b_in+8) + P*D*M*N*16.
//
// Compiled with -ffp-contract=off: the "faithful" products below must round like rustc's
// (no FMA); the FFT butterflies use explicit __builtin_fmaf.
#include "acq_device.h"

namespace gm {

// ------------------------------------------------------------------------------------ decision replay
// One wavefront per worker: the D metric triples are staged to LDS (dsm: 3 D words) with coalesced loads, then the wave replays the
// reference's sequential scan out of LDS (no dependent global-memory round trips).  WG: the wave is a workgroup of its own
// (decide_kernel) — else it is wave 0 of a stage-F workgroup (acq_mix_fft_kernel's trailing workgroups) and orders its LDS
// accesses by itself.
template <bool WG>
__device__ __forceinline__ void decide_body(const DecideArgs& a, int p, int lane, float* dsm) {
    const int D = a.n_bins;
    float* smax = dsm;
    float* ssum = dsm + D;
    uint32_t* sarg = reinterpret_cast<uint32_t*>(dsm + 2 * D);
    for (int d = lane; d < D; d += 64) {
        const size_t o = size_t(p) * D + d;
        smax[d] = a.mmax[o]; ssum[d] = a.msum[o]; sarg[d] = a.margmax[o];
    }
    if constexpr (WG) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
    // The reference scans the bins in ascending order keeping a running best plane (first strict maximum, :195-202) and
    // stops at the first bin where that running best passes max/avg > threshold (:204-223).  Here lane d holds the running
    // best THROUGH bin d (its own scan of smax[0..d], same comparisons), evaluates the test for it with the same two IEEE
    // divisions, and the lowest passing lane is the reference's exit point: 41 dependent iterations become one.
    const bool searched = (p >= 64) || ((a.mask_lo >> p) & 1ull);
    float gmax = 0.0f, bsum = 0.0f;                        // best plane starts all-zero (:168)
    uint32_t bphase = 0;
    int bbin = -1;
    bool pass = false;
    float carry_v = 0.0f;                                  // (prefix-scan form) the running best at the end of the previous 64 bins
    int carry_b = -1;
    for (int d0 = 0; d0 < D; d0 += 64) {                   // D <= 64 in practice: one trip
        const int d = d0 + lane;
        if (d0 > 0) { gmax = 0.0f; bsum = 0.0f; bphase = 0; bbin = -1; }
#ifdef GM_DECIDE_SERIAL_SCAN      // the scan as it was: every lane walks smax[0..d] out of LDS (41 dependent round trips: ~2 us)
        if (d < D && searched) {
            for (int q = 0; q <= d; ++q) {
                const float lm = smax[q];
                if (lm > gmax) { gmax = lm; bphase = sarg[q]; bsum = ssum[q]; bbin = q; }
            }
        }
#else
        // running best THROUGH bin d as a wave prefix scan of (value, bin) with the reference's comparison — a later bin replaces
        // the running best only if strictly larger (:195-202), the start value 0.0 never loses to a non-positive or NaN maximum —
        // six shuffle steps instead of d + 1 dependent LDS reads
        {
            const float own = (d < D) ? smax[d] : 0.0f;
            gmax = own > 0.0f ? own : 0.0f;
            bbin = own > 0.0f ? d : -1;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float ev = __shfl_up(gmax, off, 64);      // the running best of the lanes before
                const int eb = __shfl_up(bbin, off, 64);
                if (lane >= off && !(gmax > ev)) { gmax = ev; bbin = eb; }
            }
            if (d0 > 0 && !(gmax > carry_v)) { gmax = carry_v; bbin = carry_b; }     // D > 64: the best through bin d0 - 1
            carry_v = __shfl(gmax, 63, 64); carry_b = __shfl(bbin, 63, 64);
            bphase = bbin >= 0 ? sarg[bbin] : 0u;
            bsum = bbin >= 0 ? ssum[bbin] : 0.0f;
        }
#endif
        if (d < D && searched) {
            if (!a.best_bin_mode || d + 1 == D) {          // strongest-bin mode: test once, after the last bin
                const float avg = __fdiv_rn(bsum - gmax, float(a.fft_size - 1));   // (sum - max) / (N-1)  (:236)
                pass = __fdiv_rn(gmax, avg) > a.threshold;                          // max/avg > 7.0       (:237)
            }
        }
        const unsigned long long votes = __ballot(pass);
        if (votes) {
            if (lane == __ffsll((long long)votes) - 1) {       // the first bin that passes: early exit (:211-222)
                gm_acq_result r;
                r.prn = a.prn_ids[p];
                r.code_phase_samples = bphase;
                r.code_phase_chips = __fdiv_rn(float(bphase) * a.code_rate, a.fs);   // (:215-216)
                r.carrier_freq = a.table_freq[bbin]; r.fs = a.fs; r.mag_relative = gmax;
                r.sample_global_index = a.local_tail + bphase; r.doppler_bin = bbin;
                a.results[p] = r;
                a.found[p] = 1;
            }
            return;
        }
    }
    if (lane == 0) {
        gm_acq_result r;
        r.prn = a.prn_ids[p]; r.code_phase_samples = 0; r.code_phase_chips = 0.f; r.carrier_freq = 0.f;
        r.fs = 0.f; r.mag_relative = 0.f; r.sample_global_index = 0; r.doppler_bin = -1;   // AcquisitionResult::new
        a.results[p] = r;
        a.found[p] = 0;
    }
}
__global__ __launch_bounds__(64) void decide_kernel(DecideArgs a) {
    extern __shared__ float dsm[];
    decide_body<true>(a, blockIdx.x, threadIdx.x, dsm);
}

// ------------------------------------------------------------------------------------ stage F
template <class PLX>
__global__ __launch_bounds__(MixPlanOf<PLX>::type::T) void acq_mix_fft_kernel(const void* __restrict__ samples, int fmt,
                                                            const cf* __restrict__ tables,
                                                            const cf* __restrict__ tw_fwd,
                                                            cf* __restrict__ spectra, int n_int,
                                                            uint32_t* __restrict__ clear_tickets,
                                                            const uint16_t* __restrict__ order, int n_items, DecideArgs dec) {
    // PLX: the size's registered plan (its correlation plan CP fixes the stored order); PL: the plan the FORWARD transform runs
    // on (MixPlanOf, acq_device.h) — `tw_fwd` is PL's base-twiddle table (PlanOps::fill_tw_mix)
    // (permuted storage orders stage the outputs through LDS, one padding element per 32: see below)
    using PL = typename MixPlanOf<PLX>::type;
    using CP = typename CorrPlanOf<PLX>::type;
    static_assert(PL::N == PLX::N, "the mix plan keeps the size");
    constexpr bool PERMUTED = CorrMode<CP>::PERMUTED;
    constexpr int STAGE = PERMUTED ? PL::N + PL::N / 32 + 1 : 0;
    constexpr int LDS_N = PL::LDS_ELEMS + PL::TW_TOTAL > STAGE ? PL::LDS_ELEMS + PL::TW_TOTAL : STAGE;
    __shared__ cf lds[LDS_N];
    // Trailing workgroups (blockIdx >= n_items, dec.n_prn of them): the PREVIOUS dwell's decision, one wave per worker — its
    // metrics are complete (this kernel follows that dwell's stage C on the stream) and stay untouched until this dwell's stage C,
    // which follows this kernel.  As a kernel of its own between stage C and the next stage F the decision costs 5 us and a
    // dependent launch per dwell with 255 CUs idle (gm_acq_set_deferred_decision: gm_acq_decide_dev then keeps it for the next
    // search; any other consumer of the results runs decide_kernel first).  n_bins <= 64 (checked by the caller).
    if (int(blockIdx.x) >= n_items) {
        static_assert(LDS_N * 8 >= 3 * 64 * 4, "room for the decision's staging");
        if (threadIdx.x < 64) decide_body<false>(dec, int(blockIdx.x) - n_items, threadIdx.x, reinterpret_cast<float*>(lds));
        return;
    }
    cf* tw = lds + PL::LDS_ELEMS;
    const int tid = threadIdx.x;
    const int d = blockIdx.x / n_int, m = blockIdx.x % n_int;
    // the ticket counters of the correlation kernel's tail split start every dwell at zero (a launch that died half-way must
    // not leave tickets behind): cleared here, one kernel ahead on the same stream, instead of by a memset launch of their own
    if (clear_tickets && blockIdx.x == 0)
        for (int i = tid; i < GM_CORR_SPLIT_MAX_ITEMS; i += PL::T) clear_tickets[i] = 0u;
    load_twiddles<PL>(tw, tw_fwd, tid);
    const size_t sbase = size_t(m) * PL::N;
    const cf* tab = tables + size_t(d) * PL::N;
    cf* dst = spectra + size_t(blockIdx.x) * PL::N;   // [d][m][k]
    constexpr int NB0 = PL::NB(0);
    // the spectrum is stored in the order the correlation kernel's inverse transform reads it (CorrLayout: pairs and / or the
    // prime-factor permutation): a permutation on the writer's side costs nothing on the reader's
    auto in = [&](int it, int r) {
        const int idx = (tid + it * PL::T) + r * NB0;
        const cf s = load_sample(samples, fmt, sbase + idx);
        const cf t = tab[idx];
        // multiply_simd_block (doppler_shift.rs:43-58): a*c + (b*d*(-1)), a*d + (b*c*(+1))
        return cf_make(s.x * t.x - s.y * t.y, s.x * t.y + s.y * t.x);
    };
    if constexpr (!PERMUTED) {
        constexpr int NBL = PL::NB(PL::NP - 1);
        lds_transform<PL, false>(in, [&](int it, int r, cf val) { dst[PairLayout<CP>::pos((tid + it * PL::T) + r * NBL)] = val; }, lds, tw, tid);
    } else {
        // Permuted order (prime-factor / hybrid correlation plans): neighbouring spectrum elements sit far apart in the stored
        // array, so storing them from the registers would be one 64-byte memory transaction per 8-byte element (stage F
        // 0.054 -> 0.073 ms at N = 16368).  The outputs go back into the (now free) LDS buffer in natural order instead — one
        // padding element per 32, so that the strided read-out below spreads over the banks — and leave in storage order,
        // coalesced; `order[p]` = the element index stored at position p (CorrLayout::index_at, tabulated once per handle).
        constexpr int NBL = PL::NB(PL::NP - 1);
        {
            cf v0[PL::IT0][PL::R0];
            Fft<PL, false>::pass0_stage1(v0, in, tid);
            __syncthreads();
            Fft<PL, false>::pass0_stage2(v0, lds, tid);
        }
        __syncthreads();
        MiddlePasses<PL, false, 1>::run(lds, tw, tid);
        cf vl[PL::ITL][PL::RL];
        Fft<PL, false>::last_stage1(vl, lds, tw, tid);
        __syncthreads();                                       // every lane has gathered its inputs: the image may be overwritten
        Fft<PL, false>::last_stage2(vl, [&](int it, int r, cf val) {
            const int k = (tid + it * PL::T) + r * NBL;
            lds[k + (k >> 5)] = val; }, tid);
        __syncthreads();
        // two consecutive positions per lane and round: one 4-byte read of the order table, two LDS reads, ONE 16-byte store —
        // a wave's store covers 1 KB of contiguous spectrum: stage F 25.4 -> 24.2 us against the element-by-element 8-byte
        // form (round 3, tools/mix_lab).  The write phase is a burst of D*M*N*8 bytes (26 MB at configs[1]) from every workgroup at once: without
        // it the kernel takes 17.6 us, which is the latency of ONE transform by a workgroup alone on its CU (the 410 transforms
        // are a single round); eight positions per lane (64 contiguous bytes, four partial-line stores) was slower: 27.5 us.
        static_assert(PL::N % 2 == 0, "N must be even");
        for (int g = tid; g < PL::N / 2; g += PL::T) {
            const uint32_t o = reinterpret_cast<const uint32_t*>(order)[g];
            const int k0 = int(o & 0xffffu), k1 = int(o >> 16);
            const cf v0 = lds[k0 + (k0 >> 5)], v1 = lds[k1 + (k1 >> 5)];
            reinterpret_cast<float4*>(dst)[g] = make_float4(v0.x, v0.y, v1.x, v1.y);
        }
    }
}

// diagnostic phase stamps of the correlation kernel (gm_acq_debug_stamps): s_memtime of lane 0 of every wave of
// workgroup 0 at the phase boundaries of each transform; null in normal operation
__device__ long long* g_corr_stamps = nullptr;
__device__ __forceinline__ void corr_stamp(long long* base, int m, int wave, int phase) {
    if (base) {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
        base[(size_t(m) * 8 + wave) * 8 + phase] = (long long)t;
    }
}
// The stamped instantiation of every correlation plan (hundreds of scratch instructions each: lane 0's stamps keep state the product
// loop does not) exists only in a DIAGNOSTIC build — GM_EXTRA_FLAGS=-DGM_DIAG_STAMPS GM_LIB_SUFFIX=_diag python build.py, loaded
// with GM_LIB_PATH (tools/acq_stamps.py) — never in libgnss_mi355x.so: there gm_acq_debug_stamps is
// GM_ERR_UNSUPPORTED and the flag below is a compile-time false.
#ifdef GM_DIAG_STAMPS
static bool g_corr_stamps_armed = false;
bool corr_stamps_built() { return true; }
void set_corr_stamps(long long* d_ptr) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_corr_stamps), &d_ptr, sizeof(d_ptr));
    g_corr_stamps_armed = d_ptr != nullptr;
}
#else
static constexpr bool g_corr_stamps_armed = false;
bool corr_stamps_built() { return false; }
void set_corr_stamps(long long*) {}
#endif

// ------------------------------------------------------------------------------------ stage C
// REF_MUL (gm_acq_cfg.reference_products): x conj(code) and |.|^2 formed exactly as num-complex does (separate multiplies and
// adds, :184-192) instead of with two fused multiply-adds each.
template <class PLX, bool KEEP_CODE, bool STAMPS, bool REF_MUL>
__global__ __launch_bounds__(CorrPlanOf<PLX>::type::T, CorrPlanOf<PLX>::type::WAVES_PER_EU) void acq_corr_kernel(
    const cf* __restrict__ spectra, const cf* __restrict__ code_fft, const cf* __restrict__ tw_inv,
    float* __restrict__ mmax, uint32_t* __restrict__ margmax, float* __restrict__ msum,
    const uint32_t* __restrict__ worker_list, int n_workers, int n_bins, int n_int, int map_mode,
    int split_from, int split_k, int split_items, float* __restrict__ split_scratch, uint32_t* __restrict__ split_counter,
    int strict_sum) {
    // XCD-aware tile map: blocks b and b+8 share an XCD (round-robin dispatch, speed only).  The (bin, worker)
    // items are numbered bin-major and every XCD takes one contiguous, EQUAL share of them: the workers of a
    // Doppler bin stay on one XCD (at most two), so that bin's M spectra (M*8N bytes) are served by that XCD's L2,
    // and no XCD gets a whole extra bin (41 bins over 8 XCDs used to give XCD 0 a full third round).
    // Grid tail: the items of an XCD beyond its last FULL round of resident workgroups (slot >= split_from) are cut into
    // n_int parts of ONE integration each, so the last round is made of short workgroups (1312 items on 512 slots are 2.56
    // rounds: whole items would take 3).  The parts of an item meet through HBM: each stores its integration's power plane
    // write-through, and the part whose ticket comes last adds the n_int planes in integration order — the very additions, in
    // the very order, an uncut item makes in registers — and does the reduction.  So a cell's {max, argmax, sum} do not
    // depend on whether its item was cut, i.e. not on how many workers share the launch: a sharded grid equals the
    // single-GPU grid word for word (tests/test_gpu_mixed_grid.py).  (Parts of several integrations would have to store a
    // plane per integration from inside the m loop; a store pending at the loop's back edge makes hipcc wait for every
    // pass-0 load at once — loads and stores share vmcnt — which cost N = 16368 23 % and N = 8000 4 % on EVERY item.)
    // Placement is for speed only: correctness does not depend on where the parts run.
    const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3;
    int slot = wslot, part = 0, parts = 1;
    if (wslot >= split_from) {
        const int h = wslot - split_from;
        slot = split_from + h / split_k;
        part = h % split_k;
        parts = split_k;
    }
    int d, p;
    if (map_mode == 0) {          // equal contiguous share of the bin-major item list per XCD
        const int items = n_bins * n_workers, share = (items + 7) >> 3;
        const int item = xcd * share + slot;
        if (slot >= share || item >= items) return;
        d = item / n_workers;
        p = int(worker_list[item - d * n_workers]);
    } else if (map_mode >= 16) {  // equal shares as in mode 0, walked in blocks of cb = map_mode >> 4 workers x the share's strip of bins
        // (the workgroups resident on an XCD at a time then share code spectra AND bin spectra; see comp_corr_kernel)
        const int cb = map_mode >> 4, rows_max = map_mode & 15;
        const int items = n_bins * n_workers, share = (items + 7) >> 3;
        const int it_lo = xcd * share, it_hi = it_lo + share < items ? it_lo + share : items;
        const int d_lo = it_lo / n_workers, per_blk = rows_max * cb;
        const int blk = slot / per_blk, rem = slot - blk * per_blk, dr = rem / cb, w = blk * cb + (rem - dr * cb);
        const int item = (d_lo + dr) * n_workers + w;
        if (w >= n_workers || item < it_lo || item >= it_hi) return;
        d = d_lo + dr;
        p = int(worker_list[w]);
    } else if (map_mode == 1) {   // whole bins per XCD (bin d on XCD d % 8)
        d = xcd + 8 * (slot / n_workers);
        if (d >= n_bins) return;
        p = int(worker_list[slot % n_workers]);
    } else {                      // whole bins for the first 8*floor(D/8) bins, the leftover bins' workers split evenly
        const int q = n_bins >> 3, whole = q * n_workers;
        if (slot < whole) {
            d = xcd + 8 * (slot / n_workers);
            p = int(worker_list[slot % n_workers]);
        } else {
            const int left = (n_bins - 8 * q) * n_workers, each = (left + 7) >> 3;
            const int j = slot - whole, item = xcd * each + j;
            if (j >= each || item >= left) return;
            d = 8 * q + item / n_workers;
            p = int(worker_list[item % n_workers]);
        }
    }

    // the inverse transform runs on the size's CORRELATION plan: the plain plan, its prime-factor form (pairwise coprime radices:
    // no twiddles, nothing to load) or a hybrid plan (constant twiddles in the code)
    using PL = typename CorrPlanOf<PLX>::type;
    constexpr bool PFA = CorrMode<PL>::PFA, HYB = CorrMode<PL>::HYBRID;
    __shared__ cf lds[PL::LDS_ELEMS + ((PFA || HYB) ? 0 : PL::TW_TOTAL)];
    cf* tw = lds + PL::LDS_ELEMS;
    const int tid = threadIdx.x;
    if constexpr (!PFA && !HYB) load_twiddles<PL>(tw, tw_inv, tid);
    constexpr int NB0 = PL::NB(0), NBL = PL::NB(PL::NP - 1);
    // does lane (tid, it) own outputs of the last pass, and which element is its output r
    auto last_active = [&](int it) { if constexpr (HYB) return PL::last_active(tid); else return tid + it * PL::T < NBL; };
    auto out_index = [&](int it, int r) {
        if constexpr (HYB) return PL::out_index(tid, r);
        else if constexpr (PFA) return Pfa<PL>::out_index(tid + it * PL::T, r);
        else return (tid + it * PL::T) + r * NBL;
    };

    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(spectra + size_t(d) * n_int * PL::N, unsigned(n_int) * PL::N * 8u);
    const __amdgpu_buffer_rsrc_t crs = make_rsrc(code_fft + size_t(p) * PL::N, PL::N * 8u);

    // conj(code spectrum) for this lane's pass-0 elements: resident in registers across the m loop
    // when the register budget allows (KEEP_CODE), else re-read from L2 next to the spectrum
    using PLd = PairLoad<PL>;
    constexpr bool CODE_PAIRED = CorrLayout<PL>::CODE_PAIRED;    // both arrays paired, or both natural (PairLayout)
    cf cc[KEEP_CODE ? PL::IT0 : 1][KEEP_CODE ? PL::R0 : 1];
    if constexpr (KEEP_CODE) {
#pragma unroll
        for (int it = 0; it < PL::IT0; ++it) {
            const int b = tid + it * PL::T;
            if (b < NB0) {
                PLd q;
                if constexpr (CODE_PAIRED) q.load(crs, b, 0);
#pragma unroll
                for (int r = 0; r < PL::R0; ++r) {
                    const cf c = CODE_PAIRED ? q.get(r) : buf_load_cf(crs, b * 8, r * NB0 * 8);
                    cc[it][r] = cf_make(c.x, -c.y);
                }
            }
        }
    }
    float acc[PL::ITL][PL::RL];
#pragma unroll
    for (int it = 0; it < PL::ITL; ++it)
#pragma unroll
        for (int r = 0; r < PL::RL; ++r) acc[it][r] = 0.0f;

    long long* stbase = nullptr;
    if constexpr (STAMPS) stbase = (blockIdx.x == 0 && (tid & 63) == 0 && PL::T / 64 <= 8) ? g_corr_stamps : nullptr;
    const int wv = tid >> 6;
    const int m_per = n_int / parts, m_begin = part * m_per, m_end = m_begin + m_per;
    constexpr bool CAN_SPLIT = !STAMPS;
    constexpr int RL4 = (PL::RL + 3) / 4;                                         // 16-byte groups of a lane's power values (the last one padded)
    auto acc_or_zero = [&](int it, int r) { return r < PL::RL ? acc[it][r < PL::RL ? r : 0] : 0.0f; };
    // PREFETCH (GM_CORR_PREFETCH_PAIRS > 0, hybrid plans): the first pairs of transform m + 1's spectrum are requested from
    // inside transform m (behind its middle pass: the point with registers to spare) and travel while its last pass runs; past
    // the last integration the request goes beyond the descriptor's range (no memory access).  The transform must stay a
    // FUNCTION taking a hook: the same statements written inline made hipcc spill 76 registers.
#ifndef GM_CORR_PREFETCH_PAIRS
#define GM_CORR_PREFETCH_PAIRS 4
#endif
    constexpr int NPF = (HYB && CODE_PAIRED && !STAMPS && !KEEP_CODE) ? (GM_CORR_PREFETCH_PAIRS < PLd::NPAIR ? GM_CORR_PREFETCH_PAIRS : PLd::NPAIR) : 0;
    PLd xq[PL::IT0];
    auto load_x_ahead = [&](int m_) {
#pragma unroll
        for (int it = 0; it < PL::IT0; ++it) xq[it].template load_range<0, NPF, false>(xrs, tid + it * PL::T, m_ * PL::N);
    };
    if constexpr (NPF > 0) load_x_ahead(m_begin);
    for (int m = m_begin; m < m_end; ++m) {
        // all (remaining) pass-0 loads of this transform are issued here, pairs as 16-byte loads (PairLayout)
        PLd cq[(KEEP_CODE || !CODE_PAIRED) ? 1 : PL::IT0];
#pragma unroll
        for (int it = 0; it < PL::IT0; ++it) {
            // no branch around the loads (values that live across control flow made hipcc spill 68 VGPRs): lanes without a
            // pass-0 butterfly get an offset beyond the descriptor's range, which the buffer unit drops without a memory request
            const int b = tid + it * PL::T;
            if constexpr (CODE_PAIRED) xq[it].template load_range<NPF, PLd::NPAIR, true>(xrs, b, m * PL::N);
            if constexpr (!KEEP_CODE && CODE_PAIRED) cq[it].load(crs, b, 0);
        }
        auto in = [&](int it, int r) {
            const cf a = CODE_PAIRED ? xq[it].get(r) : buf_load_cf(xrs, (tid + it * PL::T) * 8, (m * PL::N + r * NB0) * 8);
            cf c;
            if constexpr (KEEP_CODE) c = cc[it][r];
            else if constexpr (CODE_PAIRED) { const cf g = cq[it].get(r); c = cf_make(g.x, -g.y); }
            else { const cf g = buf_load_cf(crs, (tid + it * PL::T) * 8, r * NB0 * 8); c = cf_make(g.x, -g.y); }
            // result_buf[i] *= conj(code[i])  (:184-186).  num-complex multiplies without FMA; here two of the four products are
            // fused (one rounding less each).  The value feeds the inverse FFT, whose own rounding differs from rustfft's by more
            // than that, and nothing observable sits in between: 4 instead of 6 instructions per element.  REF_MUL: num-complex's
            // own form, re = a.re*b.re - a.im*b.im, im = a.re*b.im + a.im*b.re, every product and sum rounded on its own
            if constexpr (REF_MUL) return cf_make(a.x * c.x - a.y * c.y, a.x * c.y + a.y * c.x);
            else return cf_make(__builtin_fmaf(a.x, c.x, -(a.y * c.y)), __builtin_fmaf(a.x, c.y, a.y * c.x));
        };
        // acc += norm_sqr() (:190-192): the power with one fused multiply-add, then a plain add — the running sum must stay
        // `acc + p` with p complete, so that the planes of a cut item (each 0 + p) merge to the very same words (two fused
        // multiply-adds into acc would save one more instruction and break that)
        // REF_MUL: norm_sqr() = re*re + im*im with both products rounded
        auto out = [&](int it, int r, cf v) {
            if constexpr (REF_MUL) acc[it][r] = acc[it][r] + (v.x * v.x + v.y * v.y);
            else acc[it][r] = acc[it][r] + __builtin_fmaf(v.y, v.y, v.x * v.x);
        };
        if constexpr (NPF > 0) {
            lds_transform<PL, true, PFA>(in, out, lds, tw, tid, [&](int k) {
                if (k == 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    load_x_ahead(m + 1 < m_end ? m + 1 : n_int);       // n_int: one past the descriptor's range
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        } else if constexpr (!STAMPS) {
            lds_transform<PL, true, PFA>(in, out, lds, tw, tid);
        } else {   // diagnostic variant: the same phases with stamps next to the barriers
            auto st = [&](int k) { corr_stamp(stbase, m, wv, k); };
            st(0);
            {
                cf v0[PL::IT0][PL::R0];
                Fft<PL, true, PFA>::pass0_stage1(v0, in, tid);
                st(1);
                __syncthreads();
                st(2);
                Fft<PL, true, PFA>::pass0_stage2(v0, lds, tid);
            }
            st(3);
            __syncthreads();
            MiddlePasses<PL, true, 1, PFA>::run(lds, tw, tid, st);
            cf vl[PL::ITL][PL::RL];
            Fft<PL, true, PFA>::last_stage1(vl, lds, tw, tid);
            Fft<PL, true, PFA>::last_stage2(vl, out, tid);
        }
    }

    if constexpr (CAN_SPLIT) {
        if (parts > 1) {      // a part of a cut item (one integration): its plane goes out
            // The whole byte offset travels in the VECTOR offset operand and the scalar offset is the constant 0: a
            // `buffer_store_dwordx4 v[..], v, s[..], sN offen` (scalar offset REGISTER) directly followed by a VALU write of its
            // data registers stored the NEW values of a few lanes now and then on gfx950 — the plane sums came out 0.01-0.5 %
            // low (tools/split_probe.py) — and hipcc's hazard recogniser inserts its wait state only for the forms without a
            // scalar offset register.  With soffset = 0 it does.  The guard no longer rests on that: every store below is
            // followed by an explicit `s_nop 1` (two wait states before ANY later instruction, so also before a VALU write of
            // the registers the store is still reading), and nothing zeroes or reloads `acc` before the s_waitcnt vmcnt(0).
            constexpr int SLAB = PL::ITL * RL4 * 4 * PL::T;                       // floats per power plane, register order
            // the item's planes: [n_int][SLAB] floats at its own place in the scratch
            const size_t item_plane0 = (size_t(xcd) * split_items + (slot - split_from)) * size_t(n_int);
            const __amdgpu_buffer_rsrc_t srs = make_rsrc(split_scratch + item_plane0 * SLAB, unsigned(n_int) * SLAB * 4u);
            const int voff = tid * 16 + part * SLAB * 4;
#pragma unroll
            for (int it = 0; it < PL::ITL; ++it)
#pragma unroll
                for (int r4 = 0; r4 < RL4; ++r4) {
                    u32x4 v;
                    v.x = __float_as_uint(acc_or_zero(it, 4 * r4 + 0)); v.y = __float_as_uint(acc_or_zero(it, 4 * r4 + 1));
                    v.z = __float_as_uint(acc_or_zero(it, 4 * r4 + 2)); v.w = __float_as_uint(acc_or_zero(it, 4 * r4 + 3));
                    __builtin_amdgcn_raw_buffer_store_b128(v, srs, voff + (it * RL4 + r4) * PL::T * 16, 0, 16);   // sc1: write-through
                    asm volatile("s_nop 1" ::: "memory");      // store-data hazard guard (see above): not left to the compiler's recogniser
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have left
            __syncthreads();                                         // ... and every wave's
            __shared__ int s_last;
            if (tid == 0) {
                uint32_t* cnt = split_counter + size_t(xcd) * split_items + (slot - split_from);
                const uint32_t old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = (old == uint32_t(parts - 1)) ? 1 : 0;                  // tickets are zeroed by the launcher before every launch
            }
            __syncthreads();
            if (!s_last) return;
            // the last arriver: the n_int planes added in integration order — (((0 + p0) + p1) + ...), what an uncut item's
            // registers hold — whoever comes last; sc1 loads bypass this CU's L1.  (Its own stores completed at the vmcnt(0) above.)
#pragma unroll
            for (int it = 0; it < PL::ITL; ++it)
#pragma unroll
                for (int r = 0; r < PL::RL; ++r) acc[it][r] = 0.0f;
            for (int q = 0; q < n_int; ++q) {
#pragma unroll
                for (int it = 0; it < PL::ITL; ++it)
#pragma unroll
                    for (int r4 = 0; r4 < RL4; ++r4) {
                        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srs, tid * 16, (q * SLAB + (it * RL4 + r4) * PL::T * 4) * 4, 16);
                        if (4 * r4 + 0 < PL::RL) acc[it][4 * r4 + 0] += __uint_as_float(v.x);
                        if (4 * r4 + 1 < PL::RL) acc[it][4 * r4 + 1 < PL::RL ? 4 * r4 + 1 : 0] += __uint_as_float(v.y);
                        if (4 * r4 + 2 < PL::RL) acc[it][4 * r4 + 2 < PL::RL ? 4 * r4 + 2 : 0] += __uint_as_float(v.z);
                        if (4 * r4 + 3 < PL::RL) acc[it][4 * r4 + 3 < PL::RL ? 4 * r4 + 3 : 0] += __uint_as_float(v.w);
                    }
            }
        }
    }

    // strict_sum_order: is_good_satellite's plane sum in the reference's own order (do_acquisition.rs:229-235): eight
    // running f32 sums over chunks_exact(8) — lane l adds power[8c + l] for c = 0, 1, ... — then reduce_sum, an ordered
    // add of the eight lanes starting from -0.0.  The plane goes to LDS once (the transform buffer is free now) and eight
    // lanes of wave 0 walk it sequentially: N/8 dependent adds (~4 us at N = 8000), paid only when the caller asks for
    // bit-faithful detector sums; the default is the tree sum below.
    float strict_total = 0.0f;
    if (strict_sum) {
        __syncthreads();
        float* pl = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int it = 0; it < PL::ITL; ++it) {
            if (last_active(it)) {
#pragma unroll
                for (int r = 0; r < PL::RL; ++r) pl[out_index(it, r)] = acc[it][r];
            }
        }
        __syncthreads();
        if (tid < 64) {
            float ls = 0.0f;                                        // f32x8::splat(0.0)
            if (tid < 8) {
                constexpr int CHUNKS = PL::N / 8;                   // chunks_exact(8) drops a tail
#pragma unroll 8
                for (int c = 0; c < CHUNKS; ++c) ls = ls + pl[c * 8 + tid];
            }
            float t = -0.0f;                                        // simd_reduce_add_ordered(v, -0.0)
#pragma unroll
            for (int l = 0; l < 8; ++l) t = t + __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ls), l));
            strict_total = t;
        }
    }

    // per-lane: first strict maximum + partial sum
    float bv = 0.0f, sum = 0.0f;
    uint32_t bi = 0xffffffffu;
    if constexpr (!PFA) {      // plain and hybrid plans: the element index of a slot is an add (plain) or a handful of integer operations
#pragma unroll
        for (int it = 0; it < PL::ITL; ++it) {
            if (last_active(it)) {
#pragma unroll
                for (int r = 0; r < PL::RL; ++r) {
                    take_better(bv, bi, acc[it][r], uint32_t(out_index(it, r)));
                    sum += acc[it][r];
                }
            }
        }
    } else {
        // prime-factor order: a register slot's code-phase index is Pfa::out_index(b, r) (a dozen integer operations), so the
        // scan runs on values and the index is worked out for the winner alone — for every slot that holds the maximum only
        // when several do (then the lowest index wins, as in take_better: the reference's first strict maximum, :195-202)
        int bs = -1, ties = 0;
#pragma unroll
        for (int it = 0; it < PL::ITL; ++it) {
            if (tid + it * PL::T < NBL) {
#pragma unroll
                for (int r = 0; r < PL::RL; ++r) {
                    const float v = acc[it][r];
                    if (v > bv) { bv = v; bs = it * PL::RL + r; }
                    sum += v;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < PL::ITL; ++it) {
            if (tid + it * PL::T < NBL) {
#pragma unroll
                for (int r = 0; r < PL::RL; ++r) ties += acc[it][r] == bv ? 1 : 0;
            }
        }
        if (ties == 1 && bs >= 0) {
            bi = uint32_t(Pfa<PL>::out_index(tid + (bs / PL::RL) * PL::T, bs % PL::RL));
        } else if (ties >= 1) {
#pragma unroll
            for (int it = 0; it < PL::ITL; ++it) {
                const int b = tid + it * PL::T;
                if (b < NBL) {
#pragma unroll
                    for (int r = 0; r < PL::RL; ++r)
                        if (acc[it][r] == bv) {
                            const uint32_t i = uint32_t(Pfa<PL>::out_index(b, r));
                            bi = i < bi ? i : bi;
                        }
                }
            }
        }
    }
    // wavefront (64 lanes) butterfly reduction
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const uint32_t oi = uint32_t(__shfl_xor(int(bi), off, 64));
        const float os = __shfl_xor(sum, off, 64);
        take_better(bv, bi, ov, oi);
        sum += os;
    }
    __syncthreads();   // everyone is done with the LDS transform buffer: reuse it as scratch
    float* sv = reinterpret_cast<float*>(lds);
    uint32_t* si = reinterpret_cast<uint32_t*>(lds) + 64;
    float* ss = reinterpret_cast<float*>(lds) + 128;
    const int wave = tid >> 6, lane = tid & 63;
    constexpr int NW = PL::T / 64;
    if (lane == 0) { sv[wave] = bv; si[wave] = bi; ss[wave] = sum; }
    __syncthreads();
    if (tid == 0) {
        float fv = sv[0], fs = ss[0];
        uint32_t fi = si[0];
        for (int w = 1; w < NW; ++w) { take_better(fv, fi, sv[w], si[w]); fs += ss[w]; }
        if (fi == 0xffffffffu) fi = 0;   // all-NaN plane: the reference keeps (0.0, 0)
        const size_t o = size_t(p) * n_bins + d;
        mmax[o] = fv; margmax[o] = fi; msum[o] = strict_sum ? strict_total : fs;
    }
}

}  // namespace gm
// stage C of N = 16368 (prime-factor plan ending in radix 31): the wave-specialised kernel with the radix-31 pass on the matrix pipe
#include "acq_corr_ws31.h"
namespace gm {

// ------------------------------------------------------------------------------------ replica spectrum
template <class PL>
__global__ __launch_bounds__(PL::T) void acq_code_fft_kernel(const int8_t* __restrict__ code_samples,
                                                             const cf* __restrict__ tw_fwd,
                                                             cf* __restrict__ code_fft) {
    __shared__ cf lds[PL::LDS_ELEMS + PL::TW_TOTAL];
    cf* tw = lds + PL::LDS_ELEMS;
    const int tid = threadIdx.x;
    load_twiddles<PL>(tw, tw_fwd, tid);
    const int8_t* src = code_samples + size_t(blockIdx.x) * PL::N;
    cf* dst = code_fft + size_t(blockIdx.x) * PL::N;
    constexpr int NB0 = PL::NB(0), NBL = PL::NB(PL::NP - 1);
    lds_transform<PL, false>([&](int it, int r) { return cf_make(float(src[(tid + it * PL::T) + r * NB0]), 0.0f); },
                             [&](int it, int r, cf val) { dst[(tid + it * PL::T) + r * NBL] = val; }, lds, tw, tid);
}

// ------------------------------------------------------------------------------------ plain batched FFT
template <class PL, bool INV>
__global__ __launch_bounds__(PL::T) void fft_batch_kernel(cf* __restrict__ data, const cf* __restrict__ tw_g) {
    __shared__ cf lds[PL::LDS_ELEMS + PL::TW_TOTAL];
    cf* tw = lds + PL::LDS_ELEMS;
    const int tid = threadIdx.x;
    load_twiddles<PL>(tw, tw_g, tid);
    cf* x = data + size_t(blockIdx.x) * PL::N;
    constexpr int NB0 = PL::NB(0), NBL = PL::NB(PL::NP - 1);
    // in place: every input is in registers (pass-0 stage 1) before the first barrier, outputs are
    // written after the last pass
    lds_transform<PL, INV>([&](int it, int r) { return x[(tid + it * PL::T) + r * NB0]; },
                           [&](int it, int r, cf val) { x[(tid + it * PL::T) + r * NBL] = val; }, lds, tw, tid);
}

// ------------------------------------------------------------------------------------ long power-of-two FFT (FFT<T>, fft.rs:5-30)
// Lengths above one LDS buffer, L = N1 * N2 with both factors in-LDS power-of-two plans (four-step; the same decomposition as
// the fine-Doppler transform below, without its fused inputs and reduction): n = N2*n1 + n2, k = k1 + N1*k2,
//   X[k] = sum_n2 W_N2^{n2 k2} * [ W_L^{n2 k1} * sum_n1 x[N2 n1 + n2] W_N1^{n1 k1} ].
// API parity for FFT<T> of any length (through Bluestein, gm_api.hip), not a hot path: strided accesses as they come.
template <class PL, bool INV>
__global__ __launch_bounds__(PL::T) void big_cols_kernel(const cf* __restrict__ x, cf* __restrict__ B, const cf* __restrict__ tw_g, uint32_t N2) {
    __shared__ cf lds[PL::LDS_ELEMS + PL::TW_TOTAL];     // grid N2; PL::N == N1
    cf* tw = lds + PL::LDS_ELEMS;
    const int tid = threadIdx.x;
    const uint32_t n2 = blockIdx.x, L = uint32_t(PL::N) * N2;
    load_twiddles<PL>(tw, tw_g, tid);
    cf* dst = B + size_t(n2) * PL::N;
    constexpr int NB0 = PL::NB(0), NBL = PL::NB(PL::NP - 1);
    const float inv_l = 2.0f / float(L);
    lds_transform<PL, INV>(
        [&](int it, int r) { return x[size_t(N2) * uint32_t((tid + it * PL::T) + r * NB0) + n2]; },
        [&](int it, int r, cf v) {
            const uint32_t k1 = uint32_t((tid + it * PL::T) + r * NBL);
            const uint32_t t = (n2 * k1) & (L - 1u);                 // exact phase index mod L (L <= 2^28)
            float sn, cs;
            sincospif(float(t) * inv_l, &sn, &cs);                   // W_L^{-+ n2 k1}
            if (INV) sn = -sn;
            dst[k1] = cf_make(__builtin_fmaf(v.x, cs, v.y * sn), __builtin_fmaf(v.y, cs, -(v.x * sn)));
        },
        lds, tw, tid);
}
template <class PL, bool INV>
__global__ __launch_bounds__(PL::T) void big_rows_kernel(const cf* __restrict__ B, cf* __restrict__ X, const cf* __restrict__ tw_g, uint32_t N1) {
    __shared__ cf lds[PL::LDS_ELEMS + PL::TW_TOTAL];     // grid N1; PL::N == N2
    cf* tw = lds + PL::LDS_ELEMS;
    const int tid = threadIdx.x;
    const uint32_t k1 = blockIdx.x;
    load_twiddles<PL>(tw, tw_g, tid);
    constexpr int NB0 = PL::NB(0), NBL = PL::NB(PL::NP - 1);
    lds_transform<PL, INV>(
        [&](int it, int r) { return B[size_t(uint32_t((tid + it * PL::T) + r * NB0)) * N1 + k1]; },
        [&](int it, int r, cf v) { X[size_t(N1) * uint32_t((tid + it * PL::T) + r * NBL) + k1] = v; },
        lds, tw, tid);
}

// ------------------------------------------------------------------------------------ fine Doppler
// finer_doppler (acquisition_bk.rs:215-302): X = FFT_{N1*N2}( zero-pad( (s[cp+n] - mean) * chip(n) ) ), peak of |X|.
// n = N2*n1 + n2, k = k1 + N1*k2:  X[k] = sum_n2 W_N2^{n2 k2} * [ W_N^{n2 k1} * sum_n1 x[N2 n1 + n2] W_N1^{n1 k1} ].
template <class PL>
__global__ __launch_bounds__(PL::T) void fine_cols_kernel(FineArgs a) {     // grid (N2, S); PL::N == N1
    __shared__ cf lds[PL::LDS_ELEMS + PL::TW_TOTAL];
    cf* tw = lds + PL::LDS_ELEMS;
    const int tid = threadIdx.x;
    const uint32_t n2 = blockIdx.x, sat = blockIdx.y;
    load_twiddles<PL>(tw, a.tw1, tid);
    const uint32_t cp = a.sat_code_phase[sat];
    const int8_t* chips = a.chips + size_t(a.sat_worker[sat]) * a.code_len;
    const float mre = a.mean[0], mim = a.mean[1];
    cf* dst = a.B + (size_t(sat) * a.N2 + n2) * PL::N;
    constexpr int NB0 = PL::NB(0), NBL = PL::NB(PL::NP - 1);
    const float inv_n = 2.0f / float(a.N1 * a.N2);
    lds_transform<PL, false>(
        [&](int it, int r) {
            const uint32_t n1 = uint32_t((tid + it * PL::T) + r * NB0);
            const uint32_t n = a.N2 * n1 + n2;
            if (n >= a.size_use) return cf_make(0.0f, 0.0f);                  // zero padding (:255, :273)
            const cf s = load_sample(a.samples, a.fmt, size_t(cp) + n);
            const uint32_t ind = uint32_t(floorf((float(n) * a.code_rate) / a.fs)) % a.code_len;   // :241-247
            const float c = float(chips[ind]);
            return cf_make((s.x - mre) * c, (s.y - mim) * c);                 // :237, :266-272
        },
        [&](int it, int r, cf v) {
            const uint32_t k1 = uint32_t((tid + it * PL::T) + r * NBL);
            const uint32_t t = (n2 * k1) & (a.N1 * a.N2 - 1u);                // exact phase index mod N (N <= 2^24)
            float sn, cs;
            sincospif(float(t) * inv_n, &sn, &cs);                            // W_N^{n2 k1} = cos - j sin
            dst[k1] = cf_make(__builtin_fmaf(v.x, cs, v.y * sn), __builtin_fmaf(v.y, cs, -(v.x * sn)));
        },
        lds, tw, tid);
}

// RT adjacent rows per workgroup: B is [n2][k1], so a row's elements are N1*8 bytes apart; one lane's loads for RT rows are
// RT*8 contiguous bytes (a whole 64-byte line at RT = 8) instead of eight strided 8-byte reads of eight workgroups.
template <class PL> struct FineRows {
    static constexpr int PER = PL::IT0 * PL::R0;                       // pass-0 elements a lane owns
    static constexpr int RT = PL::T >= 1024 ? 1 : (PER <= 8 ? 8 : (PER <= 16 ? 4 : (PER <= 32 ? 2 : 1)));   // 1024 lanes: 128-VGPR cap
};

template <class PL>
__global__ __launch_bounds__(PL::T) void fine_rows_kernel(FineArgs a) {     // grid (N1 / RT, S); PL::N == N2
    constexpr int RT = FineRows<PL>::RT;
    __shared__ cf lds[PL::LDS_ELEMS + PL::TW_TOTAL];
    __shared__ float s_p[PL::T / 64];
    __shared__ uint32_t s_k[PL::T / 64];
    cf* tw = lds + PL::LDS_ELEMS;
    const int tid = threadIdx.x;
    const uint32_t k1_0 = blockIdx.x * RT, sat = blockIdx.y;
    load_twiddles<PL>(tw, a.tw2, tid);
    const cf* src = a.B + size_t(sat) * a.N2 * a.N1 + k1_0;
    constexpr int NB0 = PL::NB(0), NBL = PL::NB(PL::NP - 1);
    // the RT rows' pass-0 inputs of this lane, loaded up front (static indices: registers)
    cf tile[RT][PL::IT0][PL::R0];
#pragma unroll
    for (int it = 0; it < PL::IT0; ++it) {
        const int b = tid + it * PL::T;
        if (b < NB0) {
#pragma unroll
            for (int r = 0; r < PL::R0; ++r) {
                const cf* p = src + size_t(b + r * NB0) * a.N1;
#pragma unroll
                for (int j = 0; j < RT; ++j) tile[j][it][r] = p[j];
            }
        }
    }
    float best = -1.0f;
    uint32_t bestk = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < RT; ++j) {
        lds_transform<PL, false>(
            [&](int it, int r) { return tile[j][it][r]; },
            [&](int it, int r, cf v) {
                const uint32_t k = (k1_0 + j) + a.N1 * uint32_t((tid + it * PL::T) + r * NBL);
                const float p = __builtin_fmaf(v.x, v.x, v.y * v.y);
                if (p > best || (p == best && k < bestk)) { best = p; bestk = k; }   // first index of the maximum (:279-282)
            },
            lds, tw, tid);
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float op = __shfl_xor(best, off);
        const uint32_t ok = __shfl_xor(bestk, off);
        if (op > best || (op == best && ok < bestk)) { best = op; bestk = ok; }
    }
    if ((tid & 63) == 0) { s_p[tid >> 6] = best; s_k[tid >> 6] = bestk; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < PL::T / 64; ++w)
            if (s_p[w] > best || (s_p[w] == best && s_k[w] < bestk)) { best = s_p[w]; bestk = s_k[w]; }
        a.rowmax[size_t(sat) * (a.N1 / RT) + blockIdx.x] = best;
        a.rowarg[size_t(sat) * (a.N1 / RT) + blockIdx.x] = bestk;
    }
}

__global__ __launch_bounds__(256) void fine_final_kernel(const float* __restrict__ rowmax, const uint32_t* __restrict__ rowarg,
                                                         uint32_t n_rows, float* __restrict__ peak_pow,
                                                         uint32_t* __restrict__ peak_idx) {
    __shared__ float s_p[4];
    __shared__ uint32_t s_k[4];
    const int tid = threadIdx.x;
    const uint32_t sat = blockIdx.x;
    float best = -1.0f;
    uint32_t bestk = 0xFFFFFFFFu;
    for (uint32_t i = tid; i < n_rows; i += 256) {
        const float p = rowmax[size_t(sat) * n_rows + i];
        const uint32_t k = rowarg[size_t(sat) * n_rows + i];
        if (p > best || (p == best && k < bestk)) { best = p; bestk = k; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float op = __shfl_xor(best, off);
        const uint32_t ok = __shfl_xor(bestk, off);
        if (op > best || (op == best && ok < bestk)) { best = op; bestk = ok; }
    }
    if ((tid & 63) == 0) { s_p[tid >> 6] = best; s_k[tid >> 6] = bestk; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (s_p[w] > best || (s_p[w] == best && s_k[w] < bestk)) { best = s_p[w]; bestk = s_k[w]; }
        peak_pow[sat] = best;
        peak_idx[sat] = bestk;
    }
}
void launch_fine_final(hipStream_t st, const float* rowmax, const uint32_t* rowarg, uint32_t n_rows, int n_sats,
                       float* peak_pow, uint32_t* peak_idx) {
    hipLaunchKernelGGL(fine_final_kernel, dim3(n_sats), dim3(256), 0, st, rowmax, rowarg, n_rows, peak_pow, peak_idx);
}

// mean of the whole snapshot (:236): per-lane f32 partial sums, tree-combined (the legacy adds sequentially in f32;
// the difference is far below one part in 1e5 of the mean and the mean itself is ~1e-3 of the signal)
__global__ __launch_bounds__(1024) void fine_mean_kernel(const void* __restrict__ samples, int fmt, uint32_t n, float* __restrict__ mean) {
    __shared__ float s_re[16], s_im[16];
    const int tid = threadIdx.x;
    float re = 0.0f, im = 0.0f;
    for (uint32_t i = tid; i < n; i += 1024) {
        const cf v = load_sample(samples, fmt, i);
        re += v.x; im += v.y;
    }
    for (int off = 32; off > 0; off >>= 1) { re += __shfl_xor(re, off); im += __shfl_xor(im, off); }
    if ((tid & 63) == 0) { s_re[tid >> 6] = re; s_im[tid >> 6] = im; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) { re += s_re[w]; im += s_im[w]; }
        mean[0] = re / float(n);
        mean[1] = im / float(n);
    }
}
void launch_fine_mean(hipStream_t st, const void* samples, int fmt, uint32_t n, float* d_mean) {
    hipLaunchKernelGGL(fine_mean_kernel, dim3(1), dim3(1024), 0, st, samples, fmt, n, d_mean);
}

// ------------------------------------------------------------------------------------ launchers
template <class PL> struct Launch {
    static void fill_tw(cf* tw, bool inverse) {
        fill_twiddles<PL>(tw, inverse, [](double a) { return ::cos(a); }, [](double a) { return ::sin(a); });
    }
    using CP = typename CorrPlanOf<PL>::type;      // the plan acq_corr_kernel's inverse runs on (acq_device.h)
    static constexpr bool CORR_TW = !CorrMode<CP>::PFA && !CorrMode<CP>::HYBRID;     // the kernel loads a table (plain plan with twiddles)
    static constexpr int corr_tw_total() { if constexpr (CORR_TW) return CP::TW_TOTAL; else return 0; }
    static void fill_tw_corr(cf* tw) {
        if constexpr (CORR_TW) fill_twiddles<CP>(tw, true, [](double a) { return ::cos(a); }, [](double a) { return ::sin(a); });
    }
    static_assert(CP::T == PL::T && CP::N == PL::N, "the correlation plan keeps the size and the workgroup");
    using MP = typename MixPlanOf<PL>::type;       // the plan stage F's forward transform runs on (tw_fwd: ITS table, fill_tw_mix)
    static void fill_tw_mix(cf* tw, bool inverse) {
        fill_twiddles<MP>(tw, inverse, [](double a) { return ::cos(a); }, [](double a) { return ::sin(a); });
    }
    static void mix_fft(hipStream_t st, const void* samples, int fmt, const cf* tables, const cf* tw_fwd,
                        cf* spectra, int n_bins, int n_int, uint32_t* clear_tickets, const uint16_t* order, const DecideArgs* dec) {
        // dec (may be null): the previous dwell's deferred decision rides along as dec->n_prn trailing workgroups
        const int n_items = n_bins * n_int, n_dec = dec ? dec->n_prn : 0;
        DecideArgs none{};
        hipLaunchKernelGGL(acq_mix_fft_kernel<PL>, dim3(n_items + n_dec), dim3(MP::T), 0, st, samples, fmt,
                           tables, tw_fwd, spectra, n_int, clear_tickets, order, n_items, dec ? *dec : none);
    }
    static int fill_order(uint16_t* order) { return fill_order_table<CP>(order); }
    static constexpr bool WS31 = Ws31<CP>::USE && CorrMode<CP>::PFA;      // stage C runs acq_corr_ws31_kernel (acq_corr_ws31.h)
    static constexpr int slab() {
        constexpr int generic = CP::ITL * ((CP::RL + 3) / 4) * 4 * CP::T;
        if constexpr (WS31) return ws31_split_slab<CP>() > generic ? ws31_split_slab<CP>() : generic;
        else return generic;
    }
    static constexpr int SPLIT_SLAB = slab();                             // floats per power plane of the tail split
    static void corr(hipStream_t st, const cf* spectra, const cf* code_fft, const cf* tw_inv, float* mmax,
                     uint32_t* margmax, float* msum, const uint32_t* worker_list, int n_workers, int n_bins,
                     int n_int, float* split_scratch, int split_planes, uint32_t* split_counter, int strict_sum, int tickets_cleared,
                     int ref_mul) {
        if (n_workers <= 0) return;
        // Tile map: whole Doppler bins per XCD (best L2 locality) unless that costs an XCD an extra round of
        // workgroups (2 x 32 resident per XCD) compared with equal shares of the item list.  Measured on configs[1]
        // geometry: P = 12: 125 -> 88 us, P = 24: 195 -> 165 us with equal shares; P = 32: whole bins 3 % faster.
        static const int forced = diag_int("GM_CORR_MAP", -1);   // diagnostic override (GM_DIAGNOSTICS=1 only)
        const int per_xcd_bins = ((n_bins + 7) / 8) * n_workers, per_xcd_even = (n_bins * n_workers + 7) / 8;
        const int slots = 32 * CP::WG_PER_CU;
        const int q = n_bins / 8, left = (n_bins - 8 * q) * n_workers;
        const int per_xcd_mixed = q * n_workers + (left + 7) / 8;
        int map_mode = ((per_xcd_bins + slots - 1) / slots > (per_xcd_even + slots - 1) / slots) ? 0 : 1;
        if (map_mode == 1 && per_xcd_mixed < per_xcd_bins) map_mode = 2;   // same locality, balanced leftovers
        if (forced >= 0) map_mode = forced;
        // GM_CORR_CB = cb (diagnostic, and the default for one-workgroup-per-CU plans below): tiled walk of the equal shares
        static const int cb_env = diag_int("GM_CORR_CB", -1);
        int cb = cb_env >= 0 ? cb_env : 0;
        int share = map_mode == 0 ? per_xcd_even : (map_mode == 1 ? per_xcd_bins : per_xcd_mixed);
        if (cb > 0) {
            const int rows_max = (per_xcd_even + n_workers - 2) / n_workers + 1;
            if (rows_max <= 15 && cb <= 1000) {
                map_mode = (cb << 4) | rows_max;
                share = ((n_workers + cb - 1) / cb) * rows_max * cb;       // slots per XCD, the ragged ends' empty ones included
            }
        }
        // Grid tail.  When the queue of an XCD runs dry its resident workgroups finish one by one, and the launch ends a
        // whole workgroup duration (M transforms, ~45 us alone on a CU) after the last one started: about half a duration
        // of idle slots.  The LAST items of every XCD are therefore cut into n_int parts of one integration each, enough of
        // them that every resident slot ends on a short workgroup (items * n_int ~ slots).  Measured at configs[1]:
        // 0.229 -> 0.206 ms per launch; the part count and the item count barely matter between (2, 4) and (10, 8).
        // GM_CORR_SPLIT = 0 / 1 (no cut) and GM_CORR_SPLIT_ITEMS override for diagnostics.
        static const int split_env = diag_int("GM_CORR_SPLIT", -1);
        static const int items_env = diag_int("GM_CORR_SPLIT_ITEMS", -1);
        int split_from = share, split_k = 1, split_items = 0;
        // (strict_sum_order also keeps the reference's sequential accumulation over the integrations: no split)
        if (SPLIT_SLAB && split_scratch && split_counter && split_env != 0 && !g_corr_stamps_armed && !strict_sum &&
            (share > slots || share <= GM_CORR_SPLIT_MAX_ITEMS / 8)) {
            // share <= slots: the whole grid is resident at once and (for few workers, e.g. the reference's single-PRN
            // search) leaves most of the chip idle: then EVERY item is cut, which multiplies the parallelism by k
            const bool all = share <= slots;
            // one integration per part (see the kernel): k = n_int, when the scratch holds the planes
            if (n_int >= 2 && n_int <= GM_CORR_SPLIT_MAX_K && (!all || share * n_int <= split_planes / 8)) split_k = n_int;
            // one workgroup per CU (N = 16368 ...): the cut tail does not pay on a grid of several rounds (same box, back to back:
            // 0.440 against 0.432 ms per 32-PRN launch with and without; comparisons ACROSS gpurun boxes are worthless for a
            // 2 % question — the chips differ by more); it stays for grids that fit the chip at once, where it multiplies the parallelism
            if (CP::WG_PER_CU == 1 && !all && split_env < 0) split_k = 1;
            if (split_env == 1) split_k = 1;
            if (split_k > 1) {
                split_items = items_env > 0 ? items_env : (all ? share : (slots + split_k - 1) / split_k);
                if (split_items > share) split_items = share;
                if (split_items > GM_CORR_SPLIT_MAX_ITEMS / 8) split_items = GM_CORR_SPLIT_MAX_ITEMS / 8;
                if (split_items * n_int > split_planes / 8) split_items = split_planes / 8 / n_int;   // one plane per integration; the scratch holds split_planes of them
                if (split_items <= 0) { split_items = 0; split_k = 1; }
                split_from = share - split_items;
            }
        }
        const int grid = 8 * (split_from + split_items * split_k);
        // the merge tickets start from zero in EVERY launch (a launch that was aborted, or two host threads on one handle,
        // must not leave a count behind that makes a later dwell merge early): the whole ticket block, a multiple of 16 bytes
        // at the start of its allocation (cdna_hip_programming.md G16 "Re-initialise every call")
        if (split_k > 1 && !tickets_cleared) (void)hipMemsetAsync(split_counter, 0, size_t(GM_CORR_SPLIT_MAX_ITEMS) * sizeof(uint32_t), st);
        if constexpr (WS31) {
            if (!g_corr_stamps_armed) {       // (always, in the product build: the generic kernel of this size exists in a diagnostic build only)
                if (ref_mul)
                    hipLaunchKernelGGL((acq_corr_ws31_kernel<CP, true>), dim3(grid), dim3(1024), 0, st, spectra, code_fft, mmax, margmax, msum,
                                       worker_list, n_workers, n_bins, n_int, map_mode, split_from, split_k, split_items, split_scratch, split_counter, strict_sum);
                else
                    hipLaunchKernelGGL((acq_corr_ws31_kernel<CP, false>), dim3(grid), dim3(1024), 0, st, spectra, code_fft, mmax, margmax, msum,
                                       worker_list, n_workers, n_bins, n_int, map_mode, split_from, split_k, split_items, split_scratch, split_counter, strict_sum);
                return;
            }
        }
#ifdef GM_DIAG_STAMPS
        if (g_corr_stamps_armed) {  // diagnostic build of the same kernel (gm_acq_debug_stamps)
            hipLaunchKernelGGL((acq_corr_kernel<PL, CP::KEEP_CODE, true, false>), dim3(grid), dim3(PL::T), 0, st,
                               spectra, code_fft, tw_inv, mmax, margmax, msum, worker_list, n_workers, n_bins, n_int, map_mode,
                               split_from, split_k, split_items, split_scratch, split_counter, strict_sum);
            return;
        }
#endif
        // the generic kernel of a wave-specialised size is not even instantiated in the product build
#ifdef GM_DIAG_STAMPS
        constexpr bool GENERIC = true;
#else
        constexpr bool GENERIC = !WS31;
#endif
        if constexpr (GENERIC) {
            if (ref_mul)          // gm_acq_cfg.reference_products: the reference's own rounding of x conj(code) and |.|^2
                hipLaunchKernelGGL((acq_corr_kernel<PL, CP::KEEP_CODE, false, true>), dim3(grid), dim3(PL::T), 0, st,
                                   spectra, code_fft, tw_inv, mmax, margmax, msum, worker_list, n_workers, n_bins, n_int, map_mode,
                                   split_from, split_k, split_items, split_scratch, split_counter, strict_sum);
            else
                hipLaunchKernelGGL((acq_corr_kernel<PL, CP::KEEP_CODE, false, false>), dim3(grid), dim3(PL::T), 0, st,
                                   spectra, code_fft, tw_inv, mmax, margmax, msum, worker_list, n_workers, n_bins, n_int, map_mode,
                                   split_from, split_k, split_items, split_scratch, split_counter, strict_sum);
        }
    }
    static void code_fft(hipStream_t st, const int8_t* code_samples, const cf* tw_fwd, cf* out, int n_codes) {
        hipLaunchKernelGGL(acq_code_fft_kernel<PL>, dim3(n_codes), dim3(PL::T), 0, st, code_samples, tw_fwd, out);
    }
    static void pair_codes(hipStream_t st, const cf* nat, cf* paired, int n_codes) {
        hipLaunchKernelGGL(relayout_kernel<CP>, dim3(n_codes * 4 < 1024 ? n_codes * 4 : 1024), dim3(256), 0, st, nat, paired, n_codes);
    }
    static void fft_batch(hipStream_t st, cf* data, const cf* tw, int inverse, int batch) {
        if (inverse) hipLaunchKernelGGL((fft_batch_kernel<PL, true>), dim3(batch), dim3(PL::T), 0, st, data, tw);
        else hipLaunchKernelGGL((fft_batch_kernel<PL, false>), dim3(batch), dim3(PL::T), 0, st, data, tw);
    }
    static constexpr bool POW2 = (PL::N & (PL::N - 1)) == 0;
    // four-step passes of the long power-of-two FFT: this plan as N1 (columns, grid N2) / as N2 (rows, grid N1)
    static void big_cols(hipStream_t st, const cf* x, cf* B, const cf* tw, uint32_t n2, int inverse) {
        if constexpr (POW2) {
            if (inverse) hipLaunchKernelGGL((big_cols_kernel<PL, true>), dim3(n2), dim3(PL::T), 0, st, x, B, tw, n2);
            else hipLaunchKernelGGL((big_cols_kernel<PL, false>), dim3(n2), dim3(PL::T), 0, st, x, B, tw, n2);
        }
    }
    static void big_rows(hipStream_t st, const cf* B, cf* X, const cf* tw, uint32_t n1, int inverse) {
        if constexpr (POW2) {
            if (inverse) hipLaunchKernelGGL((big_rows_kernel<PL, true>), dim3(n1), dim3(PL::T), 0, st, B, X, tw, n1);
            else hipLaunchKernelGGL((big_rows_kernel<PL, false>), dim3(n1), dim3(PL::T), 0, st, B, X, tw, n1);
        }
    }
    static void fine_cols(hipStream_t st, const FineArgs& a, int n_sats) {
        if constexpr (POW2) hipLaunchKernelGGL(fine_cols_kernel<PL>, dim3(a.N2, n_sats), dim3(PL::T), 0, st, a);
    }
    static void fine_rows(hipStream_t st, const FineArgs& a, int n_sats) {
        if constexpr (POW2) hipLaunchKernelGGL(fine_rows_kernel<PL>, dim3(a.N1 / FineRows<PL>::RT, n_sats), dim3(PL::T), 0, st, a);
    }
    static constexpr PlanOps ops() {
        return PlanOps{PL::N, PL::T, PL::TW_TOTAL, int(sizeof(cf)) * (CP::LDS_ELEMS + corr_tw_total()), SPLIT_SLAB,
                       CorrLayout<CP>::RELAYOUT ? 1 : 0,
                       &fill_tw, &fill_order, &mix_fft, &corr, &code_fft, &pair_codes, &fft_batch,
                       POW2 ? &fine_cols : nullptr, POW2 ? &fine_rows : nullptr, FineRows<PL>::RT,
                       POW2 ? &big_cols : nullptr, POW2 ? &big_rows : nullptr, MP::TW_TOTAL, &fill_tw_mix, corr_tw_total(), &fill_tw_corr};
    }
};

#define GM_PLAN_ENTRY(PL) Launch<PL>::ops(),
static const PlanOps g_plans[] = {GM_FOR_EACH_PLAN(GM_PLAN_ENTRY)};

const PlanOps* find_plan(int n) {
    for (const PlanOps& p : g_plans)
        if (p.n == n) return &p;
    return nullptr;
}
int list_plans(uint32_t* sizes, int cap) {
    int k = 0;
    for (const PlanOps& p : g_plans) {
        if (k < cap && sizes) sizes[k] = uint32_t(p.n);
        ++k;
    }
    return k;
}

// ------------------------------------------------------------------------------------ elementwise
__global__ void apply_doppler_kernel(const cf* __restrict__ s, const cf* __restrict__ t, cf* __restrict__ out,
                                     size_t n4) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += size_t(gridDim.x) * blockDim.x) {
        const cf a = s[i], b = t[i];
        out[i] = cf_make(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
    }
}
void launch_apply_doppler(hipStream_t st, const cf* s, const cf* t, cf* out, size_t n) {
    const size_t n4 = (n / 4) * 4;   // doppler_shift.rs:26: only whole groups of four
    if (!n4) return;
    const int blocks = int((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(apply_doppler_kernel, dim3(blocks), dim3(256), 0, st, s, t, out, n4);
}

__global__ void power_kernel(const cf* __restrict__ x, float* __restrict__ p, size_t n) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const cf v = x[i];
        p[i] = v.x * v.x + v.y * v.y;
    }
}
void launch_power(hipStream_t st, const cf* x, float* p, size_t n) {
    if (!n) return;
    const int blocks = int((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(power_kernel, dim3(blocks), dim3(256), 0, st, x, p, n);
}

void launch_decide(hipStream_t st, const DecideArgs& a) {
    if (a.n_prn <= 0) return;
    hipLaunchKernelGGL(decide_kernel, dim3(a.n_prn), dim3(64), size_t(a.n_bins) * 12, st, a);
}

}  // namespace gm
