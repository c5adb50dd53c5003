"""Oracle parity of the persistent tracking kernel at BASELINE's own shapes (TrackingManager::process_channels,
do_tracking.rs:351-371, batched on the device) and over every workgroups-per-channel value G the launcher can pick.

  cfg3  : 32 channels x 25 Msps, E/P/L, both code-index modes            -> G = 16 (one DPP row per arm)
  cfg5  : 36 channels x 50 Msps, 4092-chip BOC(1,1), VE/E/P/L/VL, n = 200 000 samples per period -> G = 8 (LDS totals)
  sweep : C in {1, 3, 9, 17, 32, 36, 64, 100, 200, 257}                  -> G in {32, 16, 8, 4, 2, 1}

Two comparisons per case, both through gm_trk_update_all on the device ring:

  * TEACHER-FORCED (the strict one): the oracle channel computes each epoch's correlator sums from its own state, they are
    compared with the device's sums of that epoch at 1e-5 of the prompt envelope (north_star's tolerance), and then the
    oracle advances its loop state with the DEVICE's sums (orc_trk_update_forced).  By induction both sides enter every
    epoch with identical state, so 1e-5 holds at EVERY epoch, not just the first, and the final state words must agree:
    every word of it, bit for bit (the epilogue is IEEE-exact f32 arithmetic and its one libm call, f32::atan, runs
    glibc's algorithm on the device: csrc/gm_libm.h).
    The same sums are also held to 5e-7 of the oracle's f64 accumulation of the same f32 products (the kernel's own
    error; the rest of the 1e-5 is the reference's sequential f32 summation order).
  * FREE-RUNNING: the oracle's plain update() from the same start.  The sums at epoch e then also carry the loop
    feedback of the earlier epochs' (sub-1e-5) differences: a 1-ulp difference of carrier_freq (2.4e-4 Hz at 3 kHz) turns
    the carrier by 1.5e-6 rad per epoch and accumulates in carrier_phase, so the bound is FREE_REL = 2.5e-5 over >= 10 epochs (measured: <= 1.3e-5).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REL = 1e-5          # north_star: correlator I/Q within 1e-5 relative (to the prompt envelope)
FREE_REL = 2.5e-5     # free-running bound (see the module docstring)
F64_REL = 5e-7      # against the same per-sample f32 products accumulated in f64: what the kernel's own error is


def _acq_result(prn, carrier_freq, fs, idx):
    return dict(prn=prn, code_phase_samples=0, code_phase_chips=0.0, carrier_freq=carrier_freq, fs=fs, mag_relative=10.0,
                sample_global_index=idx, doppler_bin=0)


def _ulps(a, b):
    a, b = np.float32(a), np.float32(b)
    if a == b:
        return 0
    ia, ib = int(a.view(np.int32)), int(b.view(np.int32))
    if (ia < 0) != (ib < 0):
        return abs(ia & 0x7fffffff) + abs(ib & 0x7fffffff)
    return abs(ia - ib)


def _compare(mgr, ring, oring, starts, make_oracle, arms, epochs_req, epochs_expect, rel=REL, free_rel=FREE_REL):
    """Run update_all once; replay it against teacher-forced and free-running oracle channels."""
    forced, free = [], []
    for i, r in enumerate(starts):
        mgr.channels[i].start(r)
        for lst in (forced, free):
            oc = make_oracle(i)
            oc.start(r)
            lst.append(oc)
    outs, proc, lost, done = mgr.update_all(ring, epochs_req)
    assert done == epochs_expect and not lost.any()
    nv = 2 * arms
    worst = dict(forced=0.0, free=0.0, f64=0.0, ulps=0)
    for i in range(len(starts)):
        for ep in range(epochs_req):
            rc_f, comp, comp64, _ = forced[i].update_forced(oring, outs[ep, i])
            rc_r, exp, _ = free[i].update_ex(oring)
            assert (rc_f != 0) == bool(proc[ep, i]) == (rc_r != 0), (i, ep)
            if not rc_f:
                continue
            env = float(np.hypot(comp[0], comp[1]))
            assert env > 100.0, (i, ep, env)                      # a signal is under the correlator
            e1 = float(np.max(np.abs(outs[ep, i] - comp[:nv]))) / env
            e2 = float(np.max(np.abs(outs[ep, i] - exp[:nv]))) / float(np.hypot(exp[0], exp[1]))
            e64 = float(np.max(np.abs(outs[ep, i] - comp64[:nv]))) / env
            worst["forced"], worst["free"], worst["f64"] = max(worst["forced"], e1), max(worst["free"], e2), max(worst["f64"], e64)
            assert e1 <= rel, ("teacher-forced", i, ep, e1)
            assert e64 <= F64_REL, ("teacher-forced vs f64 accumulation", i, ep, e64)
            assert e2 <= free_rel, ("free-running", i, ep, e2)
        s, o = mgr.channels[i].state, forced[i].c
        assert s.next_sample_index == o.next_sample_index == free[i].c.next_sample_index
        assert s.num_samples_per_code == o.num_samples_per_code
        assert s.lost_counter == o.lost_counter == 0 and s.prn == o.prn
        assert s.i_prompt == o.i_prompt and s.q_prompt == o.q_prompt
        for k in ("carrier_freq", "carrier_phase", "carrier_error", "carrier_nco", "code_phase", "code_error", "code_nco",
                  "code_rate"):
            u = _ulps(getattr(s, k), getattr(o, k))
            worst["ulps"] = max(worst["ulps"], u)
            assert u == 0, (i, k, getattr(s, k), getattr(o, k))
    return worst


@pytest.mark.parametrize("mode,exchange", [(0, "l2"), (1, "l2"), (1, "write-through")])
def test_cfg3_32_channels_25msps(gpu, oracle, mode, exchange, monkeypatch):
    """BASELINE configs[2]: 32 channels x 25 Msps, E/P/L (G = 16: the DPP-row totals of trk_persistent_kernel).
    exchange: "l2" — a channel's workgroups agree on their XCD after the first epoch and publish with plain stores (the
    form the benchmark runs); "write-through" — the cross-XCD form kept for every epoch (GM_TRK_FORCE_SC1=1), which is
    what a grid whose workgroups land on different XCDs would run."""
    from gnss_sdr_rs_amd import tracking as T, synth
    if exchange == "write-through":
        monkeypatch.setenv("GM_TRK_FORCE_SC1", "1")
    else:
        monkeypatch.delenv("GM_TRK_FORCE_SC1", raising=False)
    fs, n, C, E = 25.0e6, 25000, 32, 12
    t = oracle.ca_code_table()
    # FAITHFUL indexes GPS_CA_CODE_32_PRN[prn] (do_tracking.rs:276): PRN 32 would index row 32 (the reference panics), so
    # the channels cycle PRN 1..31 and the air carries row `prn` (the NEXT satellite's code)
    prns = [1 + (i % 31) for i in range(C)]
    uniq = sorted(set(prns))
    rows = [p if mode == 0 else p - 1 for p in uniq]
    sc = synth.tracking_scene(t, fs, 0.0, uniq, E + 2, config_id=3, cn0=47.0, code_rows=rows)
    x = synth.to_c32(sc["x"])
    ring, oring = T.MulticastRingBuffer(1 << 19), oracle.MulticastRingBuffer(1 << 19)
    ring.write_samples(x[:(E + 1) * n])
    oring.write_samples(x[:(E + 1) * n])
    mgr = T.TrackingManager(fs, n_channels=C, code_index_mode=mode)
    by_prn = {s["prn"]: s for s in sc["sats"]}
    starts = [_acq_result(p, by_prn[p]["doppler_hz"] + 20.0 - 1.5 * (i // 31), fs, by_prn[p]["code_start"])
              for i, p in enumerate(prns)]
    w = _compare(mgr, ring, oring, starts, lambda i: oracle.TrackingChannel(i, fs, code_index_mode=mode), 3, E, E)
    print("cfg3 mode", mode, w)
    mgr.close(); ring.close()


def _boc_scene(codes, fs, rate, L, n_samples, dopp, starts, amp=0.6, sigma=8.0, seed=5):
    tt = np.arange(n_samples, dtype=np.float64)
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples)) * sigma
    for c in range(codes.shape[0]):
        cp = ((tt - starts[c]) * rate / fs) % L
        sub = np.where((cp - np.floor(cp)) < 0.5, 1.0, -1.0)
        x += amp * codes[c][np.floor(cp).astype(np.int64)] * sub * np.exp(2j * np.pi * dopp[c] * tt / fs + 0.3j * c)
    return x.astype(np.complex64)


def test_cfg5_36_channels_50msps_boc_five_arms(gpu, oracle):
    """BASELINE configs[4]: 36 channels x 50 Msps, 4092-chip code at 1.023 Mcps (200 000 samples per period), BOC(1,1),
    five arms (G = 8, NV = 10: the LDS-staged totals).  No reference code exists for this: GPU vs the
    generalised oracle."""
    from gnss_sdr_rs_amd import tracking as T
    fs, L, rate, C, E = 50.0e6, 4092, 1.023e6, 36, 10
    n = int(round(fs / (rate / L)))
    assert n == 200000
    rng = np.random.default_rng(55)
    codes = np.where(rng.integers(0, 2, (C, L)) > 0, 1, -1).astype(np.int8)
    dopp = rng.uniform(-2000, 2000, C)
    cstart = rng.integers(0, 5000, C)
    x = _boc_scene(codes, fs, rate, L, (E + 1) * n, dopp, cstart)
    ring, oring = T.MulticastRingBuffer(1 << 22), oracle.MulticastRingBuffer(1 << 22)
    ring.write_samples(x)
    oring.write_samples(x)
    kw = dict(n_arms=5, early_late_space=0.25, very_early_late_space=0.6, boc11=True, codes=codes)
    mgr = T.TrackingManager(fs, n_channels=C, code_index_mode=T.CODE_INDEX_FIXED, nominal_code_rate=rate, **kw)
    starts = [_acq_result(c + 1, float(dopp[c]) + 10.0, fs, int(cstart[c])) for c in range(C)]

    def mk(i):
        return oracle.TrackingChannel(i, fs, code_index_mode=1, n_arms=5, el_space=0.25, vel_space=0.6, boc11=True,
                                      codes=codes, code_rate=rate)
    # n = 200 000: the sequential f32 sum the oracle inherits from the reference (do_tracking.rs:256-262) carries
    # ~eps/2 * sqrt(n/3) = 1.5e-5 of the envelope in rounding noise by itself, above north_star's 1e-5.  The device's
    # tree sum is held to 5e-7 of the f64-accumulated products (F64_REL, as in every other case) and to 4e-5 of the
    # sequential f32 sum.
    w = _compare(mgr, ring, oring, starts, mk, 5, E, E, rel=4e-5, free_rel=1e-4)
    print("cfg5", w)
    mgr.close(); ring.close()


@pytest.mark.parametrize("C", [1, 3, 9, 17, 32, 36, 64, 100, 200, 257])
def test_channel_count_sweep_covers_every_G(gpu, oracle, C):
    """Every workgroups-per-channel value the launcher can choose (G = 32, 16, 8, 4, 2, 1) at 8 Msps: channel i tracks
    satellite i % 8 from its own initial carrier offset, so no two channels hold the same state."""
    from gnss_sdr_rs_amd import tracking as T, synth
    fs, n, E = 8.0e6, 8000, 11
    t = oracle.ca_code_table()
    prns = [2, 5, 9, 13, 17, 22, 26, 30]
    sc = synth.tracking_scene(t, fs, 0.0, prns, E + 2, config_id=61, cn0=50.0)
    x = synth.to_c32(sc["x"])
    ring, oring = T.MulticastRingBuffer(1 << 17), oracle.MulticastRingBuffer(1 << 17)
    ring.write_samples(x[:(E + 1) * n])
    oring.write_samples(x[:(E + 1) * n])
    mgr = T.TrackingManager(fs, n_channels=C, code_index_mode=T.CODE_INDEX_FIXED)
    starts = []
    for i in range(C):
        s = sc["sats"][i % 8]
        starts.append(_acq_result(s["prn"], s["doppler_hz"] + 25.0 - 0.17 * (i // 8), fs, s["code_start"]))
    w = _compare(mgr, ring, oring, starts, lambda i: oracle.TrackingChannel(i, fs, code_index_mode=1), 3, E, E)
    print("sweep C", C, w)
    mgr.close(); ring.close()


@pytest.mark.parametrize("C", [1, 8, 15, 32, 100])
def test_share_device_grid_same_parity(gpu, oracle, C):
    """gm_trk_cfg.share_device (ABI 6: a receiver's setting — a quarter of the device's resident places, so that the front-end's and
    the acquisition's kernels run beside a tracking launch): other workgroups-per-channel counts than the default for the same C
    (8 instead of 32 at 15 channels), same parity bars — teacher-forced sums within 1e-5 of the oracle's and 5e-7 of an f64
    accumulation, loop state 0 ulps — and the asynchronous entry gives the same words as the synchronous one on this grid too."""
    from gnss_sdr_rs_amd import tracking as T, synth
    fs, n, E = 8.0e6, 8000, 11
    t = oracle.ca_code_table()
    prns = [2, 5, 9, 13, 17, 22, 26, 30]
    sc = synth.tracking_scene(t, fs, 0.0, prns, E + 2, config_id=63, cn0=50.0)
    x = synth.to_c32(sc["x"])
    ring, oring = T.MulticastRingBuffer(1 << 17), oracle.MulticastRingBuffer(1 << 17)
    ring.write_samples(x[:(E + 1) * n])
    oring.write_samples(x[:(E + 1) * n])
    starts = []
    for i in range(C):
        s = sc["sats"][i % 8]
        starts.append(_acq_result(s["prn"], s["doppler_hz"] + 25.0 - 0.17 * (i // 8), fs, s["code_start"]))
    mgr = T.TrackingManager(fs, n_channels=C, code_index_mode=T.CODE_INDEX_FIXED, share_device=True)
    w = _compare(mgr, ring, oring, starts, lambda i: oracle.TrackingChannel(i, fs, code_index_mode=1), 3, E, E)
    print("share_device C", C, w)
    states = [bytes(mgr.channels[i].state) for i in range(C)]
    mgr.close()
    m2 = T.TrackingManager(fs, n_channels=C, code_index_mode=T.CODE_INDEX_FIXED, share_device=True)
    for i, r in enumerate(starts):
        m2.channels[i].start(r)
    outs, proc, lost, done = m2.collect(m2.update_all_async(ring, E), wait=True)
    assert done == E and not lost.any() and proc.all()
    assert [bytes(m2.channels[i].state) for i in range(C)] == states
    m2.close(); ring.close()


def test_two_managers_on_two_streams_do_not_strand_each_other(gpu, oracle):
    """The persistent kernel needs its whole grid resident; two handles (two streams) launched back to back used to be able
    to hold half of each other's workgroups until the 0.2 s exchange time-out (GM_ERR_HIP).  Persistent launches are now
    chained per device on the GPU: both calls complete, with exactly the results of running them one after the other."""
    from gnss_sdr_rs_amd import tracking as T, synth
    fs, n, E, C = 8.0e6, 8000, 40, 32
    t = oracle.ca_code_table()
    prns = [2, 5, 9, 13, 17, 22, 26, 30]
    sc = synth.tracking_scene(t, fs, 0.0, prns, E + 2, config_id=62, cn0=50.0)
    ring = T.MulticastRingBuffer(1 << 19)
    ring.write_samples(synth.to_c32(sc["x"])[:(E + 1) * n])

    def fresh(offset):
        m = T.TrackingManager(fs, n_channels=C, code_index_mode=T.CODE_INDEX_FIXED)     # own stream each
        for i in range(C):
            s = sc["sats"][i % 8]
            m.channels[i].start(_acq_result(s["prn"], s["doppler_hz"] + offset - 0.3 * (i // 8), fs, s["code_start"]))
        return m
    # reference: one after the other, synchronised in between
    a, b = fresh(20.0), fresh(-15.0)
    a.update_all_dev(ring, E); a.synchronize()
    b.update_all_dev(ring, E); b.synchronize()
    want = [[m.channels[i].state for i in range(C)] for m in (a, b)]
    a.close(); b.close()
    # both in flight: launch, launch, then wait (repeated: the interleaving is up to the dispatcher)
    for _ in range(3):
        a, b = fresh(20.0), fresh(-15.0)
        a.update_all_dev(ring, E)
        b.update_all_dev(ring, E)
        a.synchronize(); b.synchronize()          # raises GmError on an exchange time-out
        for m, w in zip((a, b), want):
            for i in range(C):
                s = m.channels[i].state
                assert s.next_sample_index == w[i].next_sample_index and s.lost_counter == w[i].lost_counter == 0
                for k in ("carrier_freq", "carrier_phase", "code_phase", "code_rate", "i_prompt", "q_prompt"):
                    assert _ulps(getattr(s, k), getattr(w[i], k)) == 0, (i, k)
        a.close(); b.close()
    ring.close()


def test_epoch_length_that_flips_between_two_values(gpu, oracle):
    """fs chosen so that fs / (code_rate / 1023) sits at a half-integer (4000.5): the DLL's parts-per-million steps of
    the code rate move the rounded epoch length n between 4000 and 4001 from epoch to epoch.  The kernel keeps n without the
    two divisions only while the new rate lies in an interval proven to round to the same n (gm_libm.h spc_rate_bounds)
    and evaluates the definition otherwise; the bookkeeping (num_samples_per_code, next_sample_index) must equal the
    oracle's at every epoch and the lengths must in fact change during the run."""
    from gnss_sdr_rs_amd import tracking as T, synth
    fs, C, E = 4_000_500.0, 4, 60
    t = oracle.ca_code_table()
    prns = [4, 9, 17, 26]
    sc = synth.tracking_scene(t, fs, 0.0, prns, E + 3, config_id=61, cn0=48.0)
    x = synth.to_c32(sc["x"])
    ring, oring = T.MulticastRingBuffer(1 << 19), oracle.MulticastRingBuffer(1 << 19)
    ring.write_samples(x[:(E + 2) * 4001])
    oring.write_samples(x[:(E + 2) * 4001])
    mgr = T.TrackingManager(fs, n_channels=C, code_index_mode=1)
    starts = [_acq_result(s["prn"], s["doppler_hz"] + 15.0, fs, s["code_start"]) for s in sc["sats"]]
    forced = []
    for i, r in enumerate(starts):
        mgr.channels[i].start(r)
        oc = oracle.TrackingChannel(i, fs, code_index_mode=1)
        oc.start(r)
        forced.append(oc)
    lengths = set()
    for ep in range(E):
        outs, proc, lost, done = mgr.update_all(ring, 1)          # one pass per call: the state is read back after each
        assert done == 1 and not lost.any()
        for i, oc in enumerate(forced):
            rc, comp, comp64, _ = oc.update_forced(oring, outs[0, i])
            assert rc != 0 and proc[0, i]
            env = float(np.hypot(comp[0], comp[1]))
            assert float(np.max(np.abs(outs[0, i] - comp[:6]))) <= REL * env, (i, ep)
            s = mgr.channels[i].state
            assert s.num_samples_per_code == oc.c.num_samples_per_code, (i, ep, s.num_samples_per_code, oc.c.num_samples_per_code)
            assert s.next_sample_index == oc.c.next_sample_index, (i, ep)
            assert _ulps(s.code_rate, oc.c.code_rate) == 0 and _ulps(s.code_phase, oc.c.code_phase) == 0
            lengths.add(int(s.num_samples_per_code))
    assert lengths == {4000, 4001}, lengths
    # the same run as ONE persistent launch (the length changes inside the launch): identical state, word for word
    one = T.TrackingManager(fs, n_channels=C, code_index_mode=1)
    for i, r in enumerate(starts):
        one.channels[i].start(r)
    outs1, proc1, lost1, done1 = one.update_all(ring, E)
    assert done1 == E and proc1[:E, :C].all() and not lost1.any()
    for i in range(C):
        a, b = mgr.channels[i].state, one.channels[i].state
        for k in ("next_sample_index", "num_samples_per_code", "lost_counter", "prn"):
            assert getattr(a, k) == getattr(b, k), (i, k)
        for k in ("carrier_freq", "carrier_phase", "carrier_error", "carrier_nco", "code_phase", "code_error", "code_nco", "code_rate",
                  "i_prompt", "q_prompt"):
            assert _ulps(getattr(a, k), getattr(b, k)) == 0, (i, k)
    one.close(); mgr.close(); ring.close()


def test_phase_beyond_the_fast_range_takes_the_general_forms(gpu, oracle):
    """A 20 MHz IF at 50 Msps turns the carrier by 1.26e5 rad per epoch: beyond the 1e5 rad the fast sin/cos admits
    (fast_car_ok), so every epoch runs correlate_sample<FAST = false> (f64 argument reduction, library fmod) inside the
    persistent kernel.  Same teacher-forced comparison as the BASELINE shapes."""
    from gnss_sdr_rs_amd import tracking as T, synth
    fs, f_if, n, C, E = 50.0e6, 20.0e6, 50000, 3, 6
    t = oracle.ca_code_table()
    prns = [2, 12, 30]
    sc = synth.tracking_scene(t, fs, f_if, prns, E + 2, config_id=71, cn0=50.0)
    x = synth.to_c32(sc["x"])
    ring, oring = T.MulticastRingBuffer(1 << 19), oracle.MulticastRingBuffer(1 << 19)
    ring.write_samples(x[:(E + 1) * n])
    oring.write_samples(x[:(E + 1) * n])
    mgr = T.TrackingManager(fs, n_channels=C, code_index_mode=1)
    starts = [_acq_result(s["prn"], f_if + s["doppler_hz"] + 10.0, fs, s["code_start"]) for s in sc["sats"]]
    w = _compare(mgr, ring, oring, starts, lambda i: oracle.TrackingChannel(i, fs, code_index_mode=1), 3, E, E,
                 rel=2e-5, free_rel=2e-4)     # 50 000 terms per sum: the reference's own sequential f32 order is 1.3e-5 away from f64 (cfg5 note)
    print("20 MHz IF", w)
    mgr.close(); ring.close()


# ------------------------------------------------------------------------------------------------ gm_trk_cfg.strict_libm
# The carrier's cos / sin as glibc 2.35's cosf / sinf, restated on the device (csrc/gm_libm.h sincosf_glibc; the CPU suite
# checks the restatement against this host's libm on 1e8 arguments).  The oracle calls the host's cosf / sinf — what the
# reference's `phase.cos()` / `phase.sin()` (do_tracking.rs:234-235) resolve to — so with the switch on every sample's
# products are the SAME f32 values on both sides and what is left of the 1e-5 is the summation order alone.
STRICT_F64_REL = 2.5e-7    # device tree sum of the identical products against their f64 accumulation (measured 1.2e-7; 5e-7 without the switch)
STRICT_FREE_REL = FREE_REL  # free-running with identical products: measured 0.9e-5 (FAITHFUL) / 1.35e-5 (FIXED) over 12 epochs, against
                            # 1.3e-5 without the switch — NO tighter: what feeds the loops is the reference's own sequential f32 summation
                            # order (7.4e-6 teacher-forced, with or without the switch), which no parallel sum reproduces


@pytest.mark.parametrize("fs,f_if,doppler", [(25.0e6, 0.0, 1830.0), (16_367_600.0, 4_130_400.0, -2210.0), (8.0e6, 0.0, 3.0)])
def test_strict_libm_single_sample_products_equal_the_hosts(gpu, oracle, fs, f_if, doppler):
    """One-hot sample vectors through early_late_correlation (gm_trk_correlate): with a single sample (1 + 0j) at index i the
    prompt sums ARE that sample's products, cos(phase_i) * chip and -sin(phase_i) * chip (adding zeros is exact) — so the
    device's cos / sin of 3 x 200 carrier phases are compared with the host libm's BIT FOR BIT.  Phases: below pi/4 and up
    to ~20 rad (reduce_fast), and up to 2.6e4 rad at the reference capture's 4.1304 MHz IF (reduce_large); the state the
    calls advance (carrier_phase, code_phase) must stay word-equal as well."""
    from gnss_sdr_rs_amd import tracking as T
    n = int(oracle.num_samples_per_code(1.023e6, fs))
    mgr = T.TrackingManager(fs, n_channels=2, code_index_mode=1, strict_libm=True)
    ch, oc = mgr.channels[1], oracle.TrackingChannel(1, fs, code_index_mode=1)
    for c in (ch, oc):
        c.start(dict(prn=9, code_phase_samples=0, code_phase_chips=0.0, carrier_freq=f_if + doppler, fs=fs, mag_relative=10.0,
                     sample_global_index=0, doppler_bin=0))
    rng = np.random.default_rng(int(fs) % 1000)
    idx = np.concatenate([[0, 1, 2, n - 1], rng.integers(0, n, 196)])
    seen_nonzero = 0
    for i in idx:
        seg = np.zeros(n, np.complex64)
        seg[int(i)] = 1.0
        got = np.array(ch.early_late_correlation(seg), np.float32)
        exp = np.array(oc.early_late_correlation(seg), np.float32)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (int(i), got, exp)
        seen_nonzero += int(abs(got[0]) > 0) + int(abs(got[1]) > 0)
        s = ch.state
        assert s.carrier_phase == oc.c.carrier_phase and s.code_phase == oc.c.code_phase
    assert seen_nonzero >= 390
    # the default (strict_libm = 0) forms differ from the host's in the last bit on a quarter of the arguments: the same probe must see it
    dflt = T.TrackingManager(fs, n_channels=2, code_index_mode=1)
    dc, oc2 = dflt.channels[1], oracle.TrackingChannel(1, fs, code_index_mode=1)
    for c in (dc, oc2):
        c.start(dict(prn=9, code_phase_samples=0, code_phase_chips=0.0, carrier_freq=f_if + doppler, fs=fs, mag_relative=10.0,
                     sample_global_index=0, doppler_bin=0))
    differ = 0
    for i in idx:
        seg = np.zeros(n, np.complex64)
        seg[int(i)] = 1.0
        got = np.array(dc.early_late_correlation(seg), np.float32)
        exp = np.array(oc2.early_late_correlation(seg), np.float32)
        assert np.max(np.abs(got[:2].astype(np.float64) - exp[:2])) <= 1.3e-7         # 2 ulp of values below 1
        differ += int(not np.array_equal(got[:2].view(np.uint32), exp[:2].view(np.uint32)))
    print("strict probe", fs, "default forms differ on", differ, "of", len(idx))
    if f_if > 0 or doppler > 100:
        assert differ > 0
    mgr.close(); dflt.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_strict_libm_cfg3_free_running(gpu, oracle, mode):
    """BASELINE configs[2] (32 channels x 25 Msps) with strict_libm: identical products, so the teacher-forced sums sit
    within the device tree sum's own error of the f64 accumulation (STRICT_F64_REL: 1.2e-7 measured, 5e-7 without the
    switch).  The FREE-RUNNING comparison — the reference's plain update() from the same start, 12 epochs of loop feedback —
    does NOT tighten (0.9e-5 / 1.35e-5 for FAITHFUL / FIXED against 1.3e-5 without the switch): the teacher-forced 7.4e-6 is the
    reference's sequential f32 summation order against the device's tree, untouched by the switch, and that is what the
    loops integrate.  So the answer to "does free-running hold 1e-5 once cos / sin are glibc's" is no; the bound stays FREE_REL."""
    from gnss_sdr_rs_amd import tracking as T, synth
    fs, n, C, E = 25.0e6, 25000, 32, 12
    t = oracle.ca_code_table()
    prns = [1 + (i % 31) for i in range(C)]
    uniq = sorted(set(prns))
    rows = [p if mode == 0 else p - 1 for p in uniq]
    sc = synth.tracking_scene(t, fs, 0.0, uniq, E + 2, config_id=3, cn0=47.0, code_rows=rows)
    x = synth.to_c32(sc["x"])
    ring, oring = T.MulticastRingBuffer(1 << 19), oracle.MulticastRingBuffer(1 << 19)
    ring.write_samples(x[:(E + 1) * n])
    oring.write_samples(x[:(E + 1) * n])
    mgr = T.TrackingManager(fs, n_channels=C, code_index_mode=mode, strict_libm=True)
    by_prn = {s["prn"]: s for s in sc["sats"]}
    starts = [_acq_result(p, by_prn[p]["doppler_hz"] + 20.0 - 1.5 * (i // 31), fs, by_prn[p]["code_start"])
              for i, p in enumerate(prns)]
    w = _compare(mgr, ring, oring, starts, lambda i: oracle.TrackingChannel(i, fs, code_index_mode=mode), 3, E, E,
                 rel=REL, free_rel=STRICT_FREE_REL)
    print("cfg3 strict mode", mode, w)
    assert w["f64"] <= STRICT_F64_REL, w
    mgr.close(); ring.close()


def test_strict_libm_large_phase_and_five_arms(gpu, oracle):
    """strict_libm on the two other sample paths: (a) a 20 MHz IF at 50 Msps (phases to 1.26e5 rad: glibc's reduce_large on
    the device, inside correlate_sample<FAST = false>), (b) the BOC(1,1) five-arm geometry of BASELINE configs[4] at a
    reduced channel count.  Teacher-forced against the oracle; the f64 bound is the strict one."""
    from gnss_sdr_rs_amd import tracking as T, synth
    fs, f_if, n, C, E = 50.0e6, 20.0e6, 50000, 3, 5
    t = oracle.ca_code_table()
    sc = synth.tracking_scene(t, fs, f_if, [2, 12, 30], E + 2, config_id=71, cn0=50.0)
    x = synth.to_c32(sc["x"])
    ring, oring = T.MulticastRingBuffer(1 << 19), oracle.MulticastRingBuffer(1 << 19)
    ring.write_samples(x[:(E + 1) * n])
    oring.write_samples(x[:(E + 1) * n])
    mgr = T.TrackingManager(fs, n_channels=C, code_index_mode=1, strict_libm=True)
    starts = [_acq_result(s["prn"], f_if + s["doppler_hz"] + 10.0, fs, s["code_start"]) for s in sc["sats"]]
    w = _compare(mgr, ring, oring, starts, lambda i: oracle.TrackingChannel(i, fs, code_index_mode=1), 3, E, E,
                 rel=2e-5, free_rel=2e-5)     # 50 000 terms per sum: the reference's sequential f32 order alone is 1.3e-5 from f64
    print("20 MHz IF strict", w)
    assert w["f64"] <= STRICT_F64_REL, w
    mgr.close(); ring.close()
    # (b)
    fs, L, rate, C, E = 50.0e6, 4092, 1.023e6, 6, 4
    n = 200000
    rng = np.random.default_rng(56)
    codes = np.where(rng.integers(0, 2, (C, L)) > 0, 1, -1).astype(np.int8)
    dopp = rng.uniform(-2000, 2000, C)
    cstart = rng.integers(0, 5000, C)
    x = _boc_scene(codes, fs, rate, L, (E + 1) * n, dopp, cstart)
    ring, oring = T.MulticastRingBuffer(1 << 21), oracle.MulticastRingBuffer(1 << 21)
    ring.write_samples(x)
    oring.write_samples(x)
    mgr = T.TrackingManager(fs, n_channels=C, code_index_mode=T.CODE_INDEX_FIXED, nominal_code_rate=rate, n_arms=5,
                            early_late_space=0.25, very_early_late_space=0.6, boc11=True, codes=codes, strict_libm=True)
    starts = [_acq_result(c + 1, float(dopp[c]) + 10.0, fs, int(cstart[c])) for c in range(C)]
    w = _compare(mgr, ring, oring, starts,
                 lambda i: oracle.TrackingChannel(i, fs, code_index_mode=1, n_arms=5, el_space=0.25, vel_space=0.6, boc11=True,
                                                  codes=codes, code_rate=rate), 5, E, E, rel=4e-5, free_rel=4e-5)
    print("cfg5 geometry strict", w)
    assert w["f64"] <= STRICT_F64_REL, w
    mgr.close(); ring.close()


# ------------------------------------------------------------------------------- strict_libm + strict_sum_order: bit for bit
def _assert_bit_identical_free_running(mgr, ring, oring, starts, make_oracle, arms, epochs):
    """update_all against the oracle's plain update() from the same start (FREE-RUNNING, no teacher forcing): every
    correlator sum of every epoch and every word of the final channel state must be the same bits."""
    free = []
    for i, r in enumerate(starts):
        mgr.channels[i].start(r)
        oc = make_oracle(i)
        oc.start(r)
        free.append(oc)
    outs, proc, lost, done = mgr.update_all(ring, epochs)
    assert done == epochs and not lost.any() and proc.all()
    nv = 2 * arms
    for i in range(len(starts)):
        for ep in range(epochs):
            rc, exp, _ = free[i].update_ex(oring)
            assert rc != 0
            got = np.ascontiguousarray(outs[ep, i, :nv], np.float32)
            want = np.ascontiguousarray(np.asarray(exp[:nv], np.float32))
            assert float(np.hypot(want[0], want[1])) > 100.0
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (i, ep, got, want)
        s, o = mgr.channels[i].state, free[i].c
        assert s.next_sample_index == o.next_sample_index and s.num_samples_per_code == o.num_samples_per_code
        assert s.lost_counter == o.lost_counter == 0 and s.prn == o.prn
        for k in ("carrier_freq", "carrier_phase", "carrier_error", "carrier_nco", "code_phase", "code_error", "code_nco",
                  "code_rate", "i_prompt", "q_prompt"):
            assert _ulps(getattr(s, k), getattr(o, k)) == 0, (i, k, getattr(s, k), getattr(o, k))


@pytest.mark.parametrize("mode", [0, 1])
def test_strict_modes_cfg3_free_running_bit_identical(gpu, oracle, mode):
    """BASELINE configs[2] (32 channels x 25 Msps, E/P/L) with gm_trk_cfg.strict_libm + strict_sum_order: glibc's cos / sin
    and the reference's sequential sums (do_tracking.rs:231-263) on the device.  40 epochs FREE-RUNNING from the acquisition
    hand-over — PLL / DLL feedback and all — and every sum of every epoch and every state word equals the oracle's bit for
    bit: the tolerance of this path is zero when the caller asks for it."""
    from gnss_sdr_rs_amd import tracking as T, synth
    fs, n, C, E = 25.0e6, 25000, 32, 40
    t = oracle.ca_code_table()
    prns = [1 + (i % 31) for i in range(C)]
    uniq = sorted(set(prns))
    rows = [p if mode == 0 else p - 1 for p in uniq]
    sc = synth.tracking_scene(t, fs, 0.0, uniq, E + 2, config_id=3, cn0=47.0, code_rows=rows)
    x = synth.to_c32(sc["x"])
    ring, oring = T.MulticastRingBuffer(1 << 21), oracle.MulticastRingBuffer(1 << 21)
    ring.write_samples(x[:(E + 1) * n])
    oring.write_samples(x[:(E + 1) * n])
    mgr = T.TrackingManager(fs, n_channels=C, code_index_mode=mode, strict_libm=True, strict_sum_order=True)
    by_prn = {s["prn"]: s for s in sc["sats"]}
    starts = [_acq_result(p, by_prn[p]["doppler_hz"] + 20.0 - 1.5 * (i // 31), fs, by_prn[p]["code_start"])
              for i, p in enumerate(prns)]
    _assert_bit_identical_free_running(mgr, ring, oring, starts, lambda i: oracle.TrackingChannel(i, fs, code_index_mode=mode), 3, E)
    mgr.close(); ring.close()


def test_strict_modes_other_paths_bit_identical(gpu, oracle):
    """The same zero-tolerance comparison on the other sample paths: (a) the reference capture's geometry, 16.3676 Msps with
    a 4.1304 MHz IF (phases to 2.6e4 rad: glibc's reduce_large; n = 16368, not a multiple of the LDS stage), (b) a 20 MHz
    IF at 50 Msps (the general forms: library fmod, IEEE division), (c) BOC(1,1), five arms, a 4092-chip custom code at
    50 Msps (n = 200 000: 196 stages per sum), (d) the unit entries early_late_correlation / do_work on caller samples."""
    from gnss_sdr_rs_amd import tracking as T, synth
    t = oracle.ca_code_table()
    for fs, f_if, prns, E, cid in ((16_367_600.0, 4_130_400.0, [3, 11, 19, 27], 15, 72), (50.0e6, 20.0e6, [2, 12, 30], 5, 71)):
        n = int(oracle.num_samples_per_code(1.023e6, fs))
        sc = synth.tracking_scene(t, fs, f_if, prns, E + 2, config_id=cid, cn0=50.0)
        x = synth.to_c32(sc["x"])
        ring, oring = T.MulticastRingBuffer(1 << 19), oracle.MulticastRingBuffer(1 << 19)
        ring.write_samples(x[:(E + 1) * n])
        oring.write_samples(x[:(E + 1) * n])
        mgr = T.TrackingManager(fs, n_channels=len(prns), code_index_mode=1, strict_libm=True, strict_sum_order=True)
        starts = [_acq_result(s["prn"], f_if + s["doppler_hz"] + 10.0, fs, s["code_start"]) for s in sc["sats"]]
        _assert_bit_identical_free_running(mgr, ring, oring, starts, lambda i: oracle.TrackingChannel(i, fs, code_index_mode=1), 3, E)
        # (d) unit entries (a fresh handle: start() keeps the code rate the loops left behind, as the reference's does)
        mgr.close()
        mgr = T.TrackingManager(fs, n_channels=2, code_index_mode=1, strict_libm=True, strict_sum_order=True)
        ch, oc = mgr.channels[0], oracle.TrackingChannel(0, fs, code_index_mode=1)
        for c in (ch, oc):
            c.start(_acq_result(sc["sats"][0]["prn"], f_if + sc["sats"][0]["doppler_hz"] + 10.0, fs, 0))
        seg = x[sc["sats"][0]["code_start"]:sc["sats"][0]["code_start"] + n]
        got = np.array(ch.early_late_correlation(seg), np.float32)
        want = np.array(oc.early_late_correlation(seg), np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (got, want)
        seg2 = x[sc["sats"][0]["code_start"] + n:sc["sats"][0]["code_start"] + 2 * n]
        got2, msg = ch.do_work(seg2)
        want2, wmsg = oc.do_work(seg2)
        assert np.array_equal(np.array(got2, np.float32).view(np.uint32), np.array(want2, np.float32).view(np.uint32)) and msg == wmsg
        for k in ("carrier_freq", "carrier_phase", "carrier_nco", "code_phase", "code_rate", "code_nco"):
            assert _ulps(getattr(ch.state, k), getattr(oc.c, k)) == 0, k
        mgr.close(); ring.close()
    # (c)
    fs, L, rate, C, E = 50.0e6, 4092, 1.023e6, 5, 4
    n = 200000
    rng = np.random.default_rng(57)
    codes = np.where(rng.integers(0, 2, (C, L)) > 0, 1, -1).astype(np.int8)
    dopp = rng.uniform(-2000, 2000, C)
    cstart = rng.integers(0, 5000, C)
    x = _boc_scene(codes, fs, rate, L, (E + 1) * n, dopp, cstart)
    ring, oring = T.MulticastRingBuffer(1 << 21), oracle.MulticastRingBuffer(1 << 21)
    ring.write_samples(x)
    oring.write_samples(x)
    mgr = T.TrackingManager(fs, n_channels=C, code_index_mode=T.CODE_INDEX_FIXED, nominal_code_rate=rate, n_arms=5,
                            early_late_space=0.25, very_early_late_space=0.6, boc11=True, codes=codes, strict_libm=True,
                            strict_sum_order=True)
    starts = [_acq_result(c + 1, float(dopp[c]) + 10.0, fs, int(cstart[c])) for c in range(C)]
    _assert_bit_identical_free_running(
        mgr, ring, oring, starts,
        lambda i: oracle.TrackingChannel(i, fs, code_index_mode=1, n_arms=5, el_space=0.25, vel_space=0.6, boc11=True, codes=codes,
                                         code_rate=rate), 5, E)
    mgr.close(); ring.close()


def test_strict_sum_order_reports_a_code_period_beyond_its_streams(gpu, oracle):
    """strict_sum_order sizes its per-sample product streams for the nominal code period + 1 %.  A channel whose code rate
    has been set 10 % low (n = 8889 samples against streams of 8144) must be REPORTED — GM_ERR_OUT_OF_RANGE from the call —
    not skipped silently; a handle with the right nominal rate runs the same state, equal to the oracle's bit for bit."""
    from gnss_sdr_rs_amd import tracking as T, synth, _lib
    fs, E = 8.0e6, 3
    t = oracle.ca_code_table()
    sc = synth.tracking_scene(t, fs, 0.0, [7], E + 3, config_id=63, cn0=50.0)
    x = synth.to_c32(sc["x"])
    ring, oring = T.MulticastRingBuffer(1 << 16), oracle.MulticastRingBuffer(1 << 16)
    ring.write_samples(x[:40000]); oring.write_samples(x[:40000])
    s = sc["sats"][0]
    r0 = _acq_result(s["prn"], s["doppler_hz"] + 5.0, fs, s["code_start"])
    slow = 1.023e6 * 0.9
    mgr = T.TrackingManager(fs, n_channels=1, code_index_mode=1, strict_libm=True, strict_sum_order=True)
    mgr.channels[0].start(r0)
    mgr.channels[0].set_state(code_rate=slow)
    with pytest.raises(_lib.GmError) as ei:
        mgr.update_all(ring, 1)
    assert ei.value.status == -5, ei.value            # GM_ERR_OUT_OF_RANGE
    mgr.close()
    ok = T.TrackingManager(fs, n_channels=1, code_index_mode=1, strict_libm=True, strict_sum_order=True, nominal_code_rate=slow)
    ok.channels[0].start(r0)
    ok.channels[0].set_state(code_rate=slow)
    oc = oracle.TrackingChannel(0, fs, code_index_mode=1)
    oc.start(r0)
    oc.c.code_rate = slow
    outs, proc, lost, done = ok.update_all(ring, E)
    assert done == E and proc.all()
    for ep in range(E):
        rc, exp, _ = oc.update_ex(oring)
        assert rc != 0 and oc.c.num_samples_per_code in (8888, 8889, 8890)
        assert np.array_equal(np.ascontiguousarray(outs[ep, 0, :6], np.float32).view(np.uint32), np.asarray(exp[:6], np.float32).view(np.uint32))
    assert ok.channels[0].state.next_sample_index == oc.c.next_sample_index
    ok.close(); ring.close()


def test_strict_modes_cfg5_full_shape_bit_identical(gpu, oracle):
    """BASELINE configs[4] at its full shape — 36 channels x 50 Msps, 4092-chip code, BOC(1,1), VE/E/P/L/VL, n = 200 000 samples
    per period (288 MB of per-sample products per pass) — with both strict switches: three code periods free-running, every
    one of the 36 x 3 x 10 sums and every state word equal to the generalised oracle's bit for bit."""
    from gnss_sdr_rs_amd import tracking as T
    fs, L, rate, C, E = 50.0e6, 4092, 1.023e6, 36, 3
    n = 200000
    rng = np.random.default_rng(55)
    codes = np.where(rng.integers(0, 2, (C, L)) > 0, 1, -1).astype(np.int8)
    dopp = rng.uniform(-2000, 2000, C)
    cstart = rng.integers(0, 5000, C)
    x = _boc_scene(codes, fs, rate, L, (E + 1) * n, dopp, cstart)
    ring, oring = T.MulticastRingBuffer(1 << 20), oracle.MulticastRingBuffer(1 << 20)
    ring.write_samples(x)
    oring.write_samples(x)
    mgr = T.TrackingManager(fs, n_channels=C, code_index_mode=T.CODE_INDEX_FIXED, nominal_code_rate=rate, n_arms=5,
                            early_late_space=0.25, very_early_late_space=0.6, boc11=True, codes=codes, strict_libm=True,
                            strict_sum_order=True)
    starts = [_acq_result(c + 1, float(dopp[c]) + 10.0, fs, int(cstart[c])) for c in range(C)]
    _assert_bit_identical_free_running(
        mgr, ring, oring, starts,
        lambda i: oracle.TrackingChannel(i, fs, code_index_mode=1, n_arms=5, el_space=0.25, vel_space=0.6, boc11=True, codes=codes,
                                         code_rate=rate), 5, E)
    mgr.close(); ring.close()


@pytest.mark.parametrize("strict", [True, False])
def test_loss_of_lock_inside_update_all(gpu, oracle, strict):
    """strict_libm + strict_sum_order through the loss-of-lock branch of do_work (do_tracking.rs:195-209) INSIDE update_all:
    three channels on one ring — the satellite that is there, a code that is not in the air and a channel started far off in
    Doppler and phase — 26 passes free-running over a stream that falls silent after three code periods.  The pass in which
    each channel gives up (20 unlocked epochs -> reset + SatelliteLost), its `lost` flag, every sum before it, the processed
    flags after it (a reset channel is skipped) and the final state must equal the oracle's exactly."""
    from gnss_sdr_rs_amd import tracking as T, synth
    fs, n, E = 8.0e6, 8000, 26
    t = oracle.ca_code_table()
    sc = synth.tracking_scene(t, fs, 0.0, [5], E + 3, config_id=65, cn0=50.0)
    x = synth.to_c32(sc["x"])
    s = sc["sats"][0]
    # LOCK_THRESHOLD = 15 (:16) is far below what int8-scale noise alone puts into the prompt sums, so the air goes SILENT after
    # three code periods: power 0 <= 15 from pass 3 on, 20 unlocked passes, reset + SatelliteLost in pass 22
    x[s["code_start"] + 3 * n:] = 0
    ring, oring = T.MulticastRingBuffer(1 << 18), oracle.MulticastRingBuffer(1 << 18)
    ring.write_samples(x[:(E + 2) * n]); oring.write_samples(x[:(E + 2) * n])
    starts = [_acq_result(5, s["doppler_hz"] + 8.0, fs, s["code_start"]),
              _acq_result(17, s["doppler_hz"], fs, s["code_start"]),                 # PRN 17 is not in the scene
              _acq_result(5, s["doppler_hz"] + 2500.0, fs, s["code_start"] + 37)]   # wrong Doppler, wrong phase
    # strict = False: the same scenario through the persistent kernel (tree sums: the three live epochs within FREE_REL, everything
    # from the silence on — zeros, flags, the pass of the reset, the reset state — exactly equal)
    mgr = T.TrackingManager(fs, n_channels=3, code_index_mode=1, strict_libm=strict, strict_sum_order=strict)
    ocs = []
    for i, r in enumerate(starts):
        mgr.channels[i].start(r)
        oc = oracle.TrackingChannel(i, fs, code_index_mode=1)
        oc.start(r)
        ocs.append(oc)
    outs, proc, lost, done = mgr.update_all(ring, E)
    lost_at = {}
    for i, oc in enumerate(ocs):
        for ep in range(E):
            rc, exp, msg = oc.update_ex(oring)
            assert (rc != 0) == bool(proc[ep, i]), (i, ep, rc)
            if rc and (strict or ep >= 4):
                assert np.array_equal(np.ascontiguousarray(outs[ep, i, :6], np.float32).view(np.uint32), np.asarray(exp[:6], np.float32).view(np.uint32)), (i, ep)
            elif rc:
                assert float(np.max(np.abs(outs[ep, i, :6] - exp[:6]))) <= FREE_REL * max(float(np.hypot(exp[0], exp[1])), 1.0e4), (i, ep)
            assert bool(lost[ep, i]) == (msg is not None), (i, ep, msg)
            if msg is not None:
                lost_at[i] = ep
        st = mgr.channels[i].state
        assert mgr.channels[i].is_active() == oc.is_active() and st.prn == oc.c.prn and st.lost_counter == oc.c.lost_counter
        assert st.next_sample_index == oc.c.next_sample_index
        for k in ("carrier_freq", "carrier_phase", "code_phase", "code_rate", "i_prompt", "q_prompt"):
            assert _ulps(getattr(st, k), getattr(oc.c, k)) == 0, (i, k)         # reset(): all zero on both sides
    assert lost_at.get(0) == 22 and lost_at.get(1) == 22 and 2 in lost_at, lost_at
    assert not any(c.is_active() for c in mgr.channels)
    mgr.close(); ring.close()
