"""End-to-end on the GPU: samples -> device ring -> acquisition on a ring snapshot (do_acquisition.rs:297-313) ->
AcquisitionResult -> TrackingChannel::start -> process_channels on the same ring (do_tracking.rs:351-371), i.e. the
two stages of src/main.rs:204-227 chained through the reference's own message types.  FIXED code indexing (the
FAITHFUL mode correlates PRN p against PRN p+1's code and cannot hold lock on a real signal)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_acquire_then_track_from_device_ring(gpu, oracle):
    from gnss_sdr_rs_amd import acquisition as A, tracking as T, synth
    t = oracle.ca_code_table()
    fs, N, M, n_ms = 4_096_000.0, 4096, 10, 75
    truth = {4: (-1730.0, 1111), 11: (640.0, 4000), 23: (2210.0, 77), 30: (-420.0, 2500)}
    sats = [dict(prn=p, prn_row=p - 1, cn0_dbhz=49.0, doppler_hz=d, code_start=c, phase=0.1 * p) for p, (d, c) in truth.items()]
    x = synth.to_c32(synth.make_scene(t, fs, 0.0, n_ms * N, sats, config_id=61))
    ring = T.MulticastRingBuffer(1 << 19)
    dop = np.arange(-2500.0, 2500.1, 100.0, dtype=np.float32)          # fine grid: the PLL pulls in from <= 50 Hz
    eng = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=dop, n_integrations=M)
    # not enough samples yet: the reference skips the round (:299)
    ring.write_samples(x[:5 * N])
    assert eng.search_ring(ring) == (None, None)
    ring.write_samples(x[5 * N:12 * N])
    res, local_tail = eng.search_ring(ring)
    assert local_tail == 2 * N                                            # head - M*N
    found = {r["prn"]: r for r in res if r}
    assert set(found) == set(truth)
    # identical to searching the same samples through the host-buffer entry
    assert eng.search(x[2 * N:12 * N], local_tail=local_tail) == res
    mx, am, sm = eng.metrics()
    mgr = T.TrackingManager(fs, n_channels=6, code_index_mode=T.CODE_INDEX_FIXED)
    for i, (prn, r) in enumerate(sorted(found.items())):
        assert r["code_phase_samples"] == truth[prn][1] and r["sample_global_index"] == local_tail + truth[prn][1]
        best = int(np.argmax(mx[prn - 1]))                                # strongest bin (the early exit may stop earlier)
        assert abs(dop[best] - truth[prn][0]) <= 50.0
        r = dict(r, carrier_freq=float(eng.table_freq[best]))
        mgr.channels[i].start(r)
        assert mgr.channels[i].is_active()
    ring.write_samples(x[12 * N:])
    outs, proc, lost, done = mgr.update_all(ring, 80)
    assert not lost.any()
    for i, (prn, r) in enumerate(sorted(found.items())):
        n_run = int(proc[:, i].sum())
        assert n_run == (n_ms * N - r["sample_global_index"]) // N        # every whole code period in the ring
        s = mgr.channels[i].state
        assert s.active and s.prn == prn and s.lost_counter == 0
        assert abs(s.carrier_freq - truth[prn][0]) < 15.0                 # PLL settled on the true Doppler
        assert abs(s.code_rate - 1.023e6) < 30.0
        ip, qp = outs[:n_run, i, 0], outs[:n_run, i, 1]
        # carrier phase lock: the prompt energy ends up in I
        assert np.mean(np.abs(ip[-20:])) > 4.0 * np.mean(np.abs(qp[-20:]))
        # code lock: early and late envelopes balanced, prompt above both
        e = np.hypot(outs[n_run - 20:n_run, i, 2], outs[n_run - 20:n_run, i, 3]).mean()
        l = np.hypot(outs[n_run - 20:n_run, i, 4], outs[n_run - 20:n_run, i, 5]).mean()
        p = np.hypot(ip[-20:], qp[-20:]).mean()
        assert abs(e - l) < 0.2 * p and p > e and p > l
    assert not proc[:, 4:].any()
    mgr.close(); eng.close(); ring.close()


def test_best_bin_decision_mode(gpu, oracle):
    """GM_DECIDE_BEST_BIN (not the reference's early exit): the strongest bin of the grid is reported when it passes
    the ratio test; GM_DECIDE_REFERENCE on the same metrics stops at the first passing bin."""
    from gnss_sdr_rs_amd import acquisition as A, synth
    t = oracle.ca_code_table()
    fs, N, M = 4_096_000.0, 4096, 4
    sats = [dict(prn_row=6, cn0_dbhz=55.0, doppler_hz=1210.0, code_start=321)]
    x = synth.to_c32(synth.make_scene(t, fs, 0.0, M * N, sats, config_id=62))
    dop = np.arange(-2500.0, 2500.1, 100.0, dtype=np.float32)
    ref = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=dop, prn_ids=[7, 8], n_integrations=M)
    best = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=dop, prn_ids=[7, 8], n_integrations=M, decision_mode=1)
    r_ref, r_best = ref.search(x), best.search(x)
    mx, am, sm = best.metrics()
    assert r_ref[1] is None and r_best[1] is None
    b = int(np.argmax(mx[0]))
    assert r_best[0]["doppler_bin"] == b and abs(dop[b] - 1210.0) <= 50.0 and r_best[0]["code_phase_samples"] == 321
    assert r_best[0]["mag_relative"] == mx[0, b] and r_best[0]["code_phase_samples"] == am[0, b]
    # the reference mode stops earlier on this strong signal (a sinc sidelobe already passes peak/mean > 7)
    assert r_ref[0]["doppler_bin"] < b and r_ref[0]["code_phase_samples"] == 321
    ref.close(); best.close()


def test_full_chain_frontend_to_nav_bits(gpu, oracle):
    """SURVEY §8 f1-f4 chained around the path: int8 IF samples -> DigitalFrontend (DC removal + NCO down-mix, f2) ->
    device ring -> acquisition on a ring snapshot -> fine Doppler (f3) -> TrackingChannel::start -> 3 s of
    process_channels -> bit sync / nav bits / preamble on the prompt I (f4).  Checked against the simulated truth."""
    from gnss_sdr_rs_amd import acquisition as A, decoding as Dm, frontend as F, tracking as T, synth
    t = oracle.ca_code_table()
    fs, N, M, f_if = 2_048_000.0, 2048, 10, 256_000.0        # f_if / fs * 2048 = 256: the LUT NCO is exact at this IF
    n_ms = 3100
    rng = np.random.default_rng(5)
    data = rng.integers(0, 2, 160) * 2 - 1
    for at in (20, 60, 100, 130):
        data[at:at + 8] = Dm.GPS_CA_PREAMBLE
    sat = dict(prn=9, prn_row=8, cn0_dbhz=50.0, doppler_hz=1337.0, code_start=700, phase=0.4, data_bits=data, bit_edge_ms=13)
    # mix_simd with the negated sine table (nco_lut.rs:8-15,31) computes conj(x) * exp(-j theta): it brings a spectrally
    # INVERTED IF stream (carrier at -(f_if + doppler)) to baseband at +doppler.  That is the stream fed here.
    x = np.conj(synth.make_scene(t, fs, f_if, n_ms * N, [sat], config_id=77))
    x = x + (6.0 - 4.0j)                                       # a DC offset for the front-end to remove
    xi8 = synth.to_i8_iq(np.clip(x.real, -127, 127) + 1j * np.clip(x.imag, -127, 127))

    ring = T.MulticastRingBuffer(1 << 23)
    fe = F.DigitalFrontend(f_if, fs, fs)
    for off in range(0, n_ms * N, 1 << 18):                    # rf_thread's block pump (larger blocks, same arithmetic)
        fe.write_ring(ring, xi8[off:off + (1 << 18)])
    ring.flush()
    assert ring.get_head() == n_ms * N
    _, br, bi = fe.state()
    assert abs(br.mean() - 6.0) < 1.0 and abs(bi.mean() + 4.0) < 1.0      # the IIR settled on the injected offset

    dop = np.arange(-2500.0, 2500.1, 250.0, dtype=np.float32)
    eng = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=dop, n_integrations=M, decision_mode=A.DECIDE_BEST_BIN)
    snap = ring.copy_to_slice(0, 12 * N)
    res = eng.search(snap[:M * N])
    found = {r["prn"]: r for r in res if r}
    assert set(found) == {9}
    r = found[9]
    assert r["code_phase_samples"] == 700
    fine = eng.finer_doppler(res)[8]
    assert abs(fine["freq_hz"] - 1337.0) < 60.0             # nav-bit edges inside the 9 ms window cost some accuracy
    mgr = T.TrackingManager(fs, n_channels=2, code_index_mode=T.CODE_INDEX_FIXED)
    mgr.channels[0].start(dict(r, carrier_freq=fine["freq_hz"]))
    outs, proc, lost, done = mgr.update_all(ring, 3090)
    n_run = int(proc[:, 0].sum())
    assert n_run >= 3080 and not lost.any()
    s = mgr.channels[0].state
    assert s.active and abs(s.carrier_freq - 1337.0) < 15.0

    nav = Dm.NavSyncStatus(Dm.NAV_FIXED)
    ip = outs[:n_run, 0, 0]
    old, st = 0.0, None
    for e in range(n_run):
        st = nav.update(float(old), float(ip[e]), e)
        old = ip[e]
    assert st["flag_bit_sync"] and st["frame_sync_ind"] == 13   # the bit edge of the simulated data
    assert st["flag_frame_sync"]
    bits = nav.frame_bits()
    assert bits.size > 30
    # the decoded bits are the transmitted ones up to the PLL's 180 degree ambiguity, which the preamble polarity resolves
    start = next(k for k in range(data.size - bits.size + 1) if (data[k:k + bits.size] == st["polarity"] * bits).all())
    assert start > 50
    mgr.close(); eng.close(); fe.close(); ring.close()


def test_decision_over_more_than_64_doppler_bins(gpu, oracle):
    """decide_kernel forms the reference's running best (first strict maximum, do_acquisition.rs:195-202) as a wave prefix scan,
    64 bins per trip with a carry between trips: 151 bins (three trips), satellites whose first passing bin lies in the
    first, second and third trip, one absent code, both decision modes.  Every Option<AcquisitionResult> must equal the oracle's
    decision replayed on the DEVICE's own metrics (orc_decide_from_metrics: the sequential scan as the reference writes it)."""
    from gnss_sdr_rs_amd import acquisition as A, synth
    t = oracle.ca_code_table()
    fs, N, M = 2.048e6, 2048, 3
    dop = np.arange(-3750.0, 3750.1, 50.0, dtype=np.float32)
    assert dop.size == 151
    sats = [dict(prn_row=2, cn0_dbhz=52.0, doppler_hz=-3100.0, code_start=100),      # passes inside the first 64 bins
            dict(prn_row=11, cn0_dbhz=50.0, doppler_hz=600.0, code_start=1500),      # ... in the second trip (bin 87)
            dict(prn_row=25, cn0_dbhz=50.0, doppler_hz=3700.0, code_start=9)]        # ... in the third (its lobe starts beyond bin 128)
    x = synth.to_c32(synth.make_scene(t, fs, 0.0, M * N, sats, config_id=64))
    prns = [3, 12, 26, 30]
    for mode in (0, 1):
        eng = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=dop, prn_ids=prns, n_integrations=M, decision_mode=mode)
        res = eng.search(x, local_tail=777)
        mx, am, sm = eng.metrics()
        trips = set()
        for i, prn in enumerate(prns):
            if mode == 0:
                exp = oracle.decide_from_metrics(mx[i], am[i], sm[i], dop, N, prn, fs, 777)
            else:      # strongest bin, tested once
                b = int(np.argmax(mx[i]))
                avg = (np.float32(sm[i, b]) - np.float32(mx[i, b])) / np.float32(N - 1)
                exp = dict(doppler_bin=b, code_phase_samples=int(am[i, b])) if np.float32(mx[i, b]) / avg > np.float32(7.0) else None
            got = res[i]
            assert (got is None) == (exp is None), (mode, prn, got, exp)
            if exp:
                assert got["code_phase_samples"] == exp["code_phase_samples"], (mode, prn)
                if mode == 0:
                    assert got["carrier_freq"] == exp["carrier_freq"] and got["mag_relative"] == exp["mag_relative"]
                    assert got["sample_global_index"] == exp["sample_global_index"]
                else:
                    assert got["doppler_bin"] == exp["doppler_bin"]
                trips.add(got["doppler_bin"] // 64)
        assert res[3] is None and trips == {0, 1, 2}, (mode, trips)
        eng.close()


def test_tracking_async_tickets_equal_the_per_block_path(gpu, oracle):
    """gm_trk_update_all_async + gm_trk_collect: a feeder writes block after block through the ring's
    asynchronous writer and enqueues the tracking passes behind each block WITHOUT waiting — ordered on the device by the ring's
    enqueue event, data gate on the enqueued head — and drains the tickets when they are ready.  Channel states and every
    correlator sum equal the synchronous per-block path (write, flush, gm_trk_update_all), bit for bit; ticket misuse is refused."""
    from gnss_sdr_rs_amd import tracking as T, synth, _lib
    t = oracle.ca_code_table()
    fs, N, n_ms, BLK = 4_096_000.0, 4096, 96, 6 * 4096 + 1000       # blocks that do NOT end on code-period boundaries
    truth = {4: (-1730.0, 1111), 11: (640.0, 4000), 23: (2210.0, 77)}
    sats = [dict(prn=p, prn_row=p - 1, cn0_dbhz=49.0, doppler_hz=d, code_start=c, phase=0.1 * p) for p, (d, c) in truth.items()]
    x = synth.to_c32(synth.make_scene(t, fs, 0.0, n_ms * N, sats, config_id=62))
    starts = [dict(prn=p, code_phase_samples=c, code_phase_chips=0.0, carrier_freq=d + 20.0, fs=fs, mag_relative=1.0,
                   sample_global_index=c, doppler_bin=0) for p, (d, c) in sorted(truth.items())]
    E = BLK // N + 3

    def run(asynchronous):
        ring = T.MulticastRingBuffer(1 << 19)
        mgr = T.TrackingManager(fs, n_channels=5, code_index_mode=T.CODE_INDEX_FIXED)
        for i, r in enumerate(starts):
            mgr.channels[i].start(r)
        got = [[] for _ in starts]

        def take(res):
            outs, proc, lost, done = res
            assert not lost.any()
            for i in range(len(starts)):
                got[i].append(outs[proc[:, i] != 0, i, :])
        tickets = []
        for off in range(0, x.size, BLK):
            ring.write_samples_async(x[off:off + BLK])
            if asynchronous:
                tickets.append(mgr.update_all_async(ring, E))
                while tickets:
                    res = mgr.collect(tickets[0], wait=len(tickets) >= 7)
                    if res is None:
                        break
                    tickets.pop(0); take(res)
            else:
                ring.flush()
                take(mgr.update_all(ring, E))
        ring.flush()
        if asynchronous:
            tickets.append(mgr.update_all_async(ring, E))
            for tk in tickets:
                take(mgr.collect(tk, wait=True))
            with pytest.raises(_lib.GmError):                # collected already
                mgr.collect(tickets[-1], wait=True)
        else:
            take(mgr.update_all(ring, E))
        states = [bytes(mgr.channels[i].state) for i in range(len(starts))]
        sums = [np.concatenate(g, axis=0) for g in got]
        mgr.close(); ring.close()
        return states, sums
    s_sync, o_sync = run(False)
    s_async, o_async = run(True)
    for i, r in enumerate(starts):
        assert o_sync[i].shape[0] == (n_ms * N - r["sample_global_index"]) // N      # every whole code period
        assert o_async[i].shape == o_sync[i].shape
        assert (o_async[i].view(np.uint32) == o_sync[i].view(np.uint32)).all(), i
        assert s_async[i] == s_sync[i], i
    # nine calls without a collect: the ninth is refused (eight result slots), nothing is launched for it
    ring = T.MulticastRingBuffer(1 << 16)
    mgr = T.TrackingManager(fs, n_channels=2, code_index_mode=T.CODE_INDEX_FIXED)
    ring.write_samples_async(x[:8 * N])
    tks = [mgr.update_all_async(ring, 2) for _ in range(8)]
    assert len(set(tks)) == 8 and 0 not in tks
    with pytest.raises(_lib.GmError) as e:
        mgr.update_all_async(ring, 2)
    assert e.value.status == -5                              # GM_ERR_OUT_OF_RANGE
    for tk in reversed(tks):                                 # any order
        assert mgr.collect(tk, wait=True) is not None
    mgr.close(); ring.close()


def test_bulk_states_enqueued_head_and_collected_states(gpu, oracle):
    """The ABI 7 entries on their own terms.  gm_trk_get_states == the per-channel gm_trk_get_state records; gm_trk_set_states
    with a `which` mask writes the flagged channels only (runs of them, the ends included) and every channel without one;
    gm_ring_get_enqueued_head is what the asynchronous writer has been handed — at once, before the copy has landed — never
    behind gm_ring_get_head and equal to it after a flush; the states gm_trk_collect hands over are the channel records as
    they stood behind THAT call's passes (a later call in flight does not show in them) and, for the last call, equal
    gm_trk_get_states word for word; null pointers are refused."""
    import ctypes as C
    from gnss_sdr_rs_amd import tracking as T, synth, _lib
    L = _lib.lib()
    t = oracle.ca_code_table()
    fs, N, n_ms = 4_096_000.0, 4096, 40
    truth = {7: (-900.0, 300), 19: (1500.0, 2222)}
    sats = [dict(prn=p, prn_row=p - 1, cn0_dbhz=50.0, doppler_hz=d, code_start=c, phase=0.2 * p) for p, (d, c) in truth.items()]
    x = synth.to_c32(synth.make_scene(t, fs, 0.0, n_ms * N, sats, config_id=64))
    ring = T.MulticastRingBuffer(1 << 18)
    mgr = T.TrackingManager(fs, n_channels=6, code_index_mode=T.CODE_INDEX_FIXED)
    raw = lambda s: bytes(C.string_at(C.addressof(s), C.sizeof(s)))
    # ---- bulk read == per-channel reads; masked bulk write
    base = mgr.get_states()
    assert [raw(s) for s in base] == [raw(mgr.channels[c].state) for c in range(6)]
    want = mgr.get_states()
    for c, s in enumerate(want):
        s.carrier_freq, s.code_phase, s.lost_counter = 100.0 + c, 0.25 * c, 3 * c
    mgr.set_states(want, which=[1, 0, 0, 1, 1, 1])             # a run at the start, a run that reaches the end
    got = mgr.get_states()
    for c in range(6):
        assert raw(got[c]) == raw(want[c] if c in (0, 3, 4, 5) else base[c]), c
    mgr.set_states(want, which=[0] * 6)                           # nothing flagged: nothing written
    assert [raw(s) for s in mgr.get_states()] == [raw(s) for s in got]
    mgr.set_states(base)                                          # no mask: every channel
    assert [raw(s) for s in mgr.get_states()] == [raw(s) for s in base]
    st = (_lib.TrkState * 6)()
    assert L.gm_trk_get_states(mgr._h, None) == -1 and L.gm_trk_set_states(mgr._h, None, None) == -1
    assert L.gm_ring_get_enqueued_head(ring._h, None) == -1
    # ---- the enqueued head
    assert ring.get_enqueued_head() == ring.get_head() == 0
    half = (n_ms // 2) * N + 777
    ring.write_samples_async(x[:half])
    assert ring.get_enqueued_head() == half >= ring.get_head()
    ring.flush()
    assert ring.get_enqueued_head() == ring.get_head() == half
    # ---- the states a collect hands over belong to ITS call
    for i, (p, (d, c0)) in enumerate(sorted(truth.items())):
        mgr.channels[i].start(dict(prn=p, code_phase_samples=c0, code_phase_chips=0.0, carrier_freq=d + 15.0, fs=fs, mag_relative=1.0,
                                   sample_global_index=c0, doppler_bin=0))
    t1 = mgr.update_all_async(ring, n_ms // 2 + 2)
    ring.write_samples_async(x[half:])
    assert ring.get_enqueued_head() == x.size
    t2 = mgr.update_all_async(ring, n_ms // 2 + 2)
    o1, p1, l1, d1, s1 = mgr.collect(t1, wait=True, with_states=True)
    o2, p2, l2, d2, s2 = mgr.collect(t2, wait=True, with_states=True)
    assert not l1.any() and not l2.any()
    for i, (p, (d, c0)) in enumerate(sorted(truth.items())):
        n1, n2 = int(p1[:, i].sum()), int(p2[:, i].sum())
        assert n1 == (half - c0) // N and n1 + n2 == (x.size - c0) // N
        assert s1[i].next_sample_index == c0 + n1 * N and s2[i].next_sample_index == c0 + (n1 + n2) * N   # (code_rate stays within a sample of N here)
        assert s1[i].active and s2[i].active and abs(s2[i].carrier_freq - d) < 15.0
    assert [raw(s) for s in s2] == [raw(s) for s in mgr.get_states()]
    assert [raw(s) for s in s1[2:]] == [raw(s) for s in base[2:]]                                          # idle channels: untouched
    mgr.close(); ring.close()


def test_ring_head_by_pull_with_concurrent_readers(gpu):
    """The asynchronous writer's head is published by PULL (ABI 6: no host callback on the copy stream): whoever looks retires the
    blocks whose copies have completed.  A writer thread and two reader threads on one ring — the single-writer / many-readers
    use of the reference ring (multicast_ring_buffer.rs:36-130) — : every head a reader sees is monotonic, never ahead of what the
    writer has enqueued, and the samples below it HAVE landed (copy_to_slice returns exactly what was written); wait_head returns
    at the head it was asked for (the Condvar wait of do_tracking.rs:392-406), and times out when nothing comes."""
    import threading
    from gnss_sdr_rs_amd import tracking as T
    rng = np.random.default_rng(5)
    BLK, NB = 3000, 160                                           # blocks that do not divide the ring or the staging slots
    x = (rng.standard_normal(BLK * NB) + 1j * rng.standard_normal(BLK * NB)).astype(np.complex64)
    ring = T.MulticastRingBuffer(1 << 20)
    written = [0]
    errors = []

    def writer():
        try:
            for b in range(NB):
                ring.write_samples_async(x[b * BLK:(b + 1) * BLK])
                written[0] = (b + 1) * BLK
            ring.flush()
        except Exception as e:                                    # pragma: no cover
            errors.append(repr(e))

    def reader(step):
        try:
            last, want = 0, step
            while last < BLK * NB:
                assert ring.wait_head(want, 5000), ("timed out waiting for", want)
                h = ring.get_head()
                assert h >= want and h >= last, (h, want, last)
                # (the writer's counter is bumped after its call returns: the head may run at most one call ahead of it)
                assert h <= written[0] + BLK, (h, written[0])
                lo = max(last, h - 4096)
                got = ring.copy_to_slice(lo, h - lo)
                assert (got.view(np.uint32) == x[lo:h].view(np.uint32)).all(), (lo, h)
                last = h
                want = min(h + step, BLK * NB)
        except Exception as e:
            errors.append(repr(e))
    th = [threading.Thread(target=writer), threading.Thread(target=reader, args=(7001,)), threading.Thread(target=reader, args=(20011,))]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join(120)
    assert not errors, errors
    assert ring.get_head() == BLK * NB
    assert not ring.wait_head(BLK * NB + 1, 20)                   # nothing more is coming: the wait times out
    ring.close()
