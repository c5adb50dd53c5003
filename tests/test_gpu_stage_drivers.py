"""The DELIVERABLE stage drivers — gnss::run_tracking / gnss::run_acquisition of host/gnss_sdr.hpp, which stand in for
do_tracking::run (do_tracking.rs:384-415) and do_acquisition::run (do_acquisition.rs:241-327) — driven through the small
extern "C" harness host/receiver_harness.cpp: the ticket loop (no host wait per block) must leave every
channel exactly where the synchronous loop leaves it, and the whole chain of main.rs:182-227 built from those drivers must run
far faster than real time with every satellite tracked."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _state_bytes(s):
    return bytes(C.string_at(C.addressof(s), C.sizeof(s)))


def _assert_same_states(ab, n):
    """every word of every channel record equal; on a mismatch the message names the fields (not a wall of bytes)"""
    diff = []
    for c in range(n):
        a, b = ab["sync"][c], ab["async_"][c]
        if _state_bytes(a) != _state_bytes(b):
            diff += [(c, k, getattr(a, k), getattr(b, k)) for k, _ in a._fields_ if k != "reserved" and getattr(a, k) != getattr(b, k)]
    assert not diff, "synchronous loop vs ticket loop (channel, field, sync, tickets): %r; epochs %r" % (diff, ab["epochs"])


def test_cpp_tracking_driver_tickets_equal_the_synchronous_loop_on_the_full_chain_scene(gpu, oracle):
    """The scene of test_full_chain_frontend_to_nav_bits (int8 IF -> DigitalFrontend -> ring, one satellite with 50 bit/s data,
    3.1 s): acquisition + fine Doppler give the hand-over; then gnss::run_tracking alone on that ring, once with the synchronous
    process_channels per loop turn and once with process_channels_async / collect (passes planned from the collected states,
    ordered on the device behind the ring's writer).  Final channel states: equal bit for bit; same epochs, same messages."""
    from gnss_sdr_rs_amd import acquisition as A, decoding as Dm, frontend as F, receiver as R, tracking as T, synth
    t = oracle.ca_code_table()
    fs, N, M, f_if = 2_048_000.0, 2048, 10, 256_000.0
    n_ms = 3100
    rng = np.random.default_rng(5)
    data = rng.integers(0, 2, 160) * 2 - 1
    for at in (20, 60, 100, 130):
        data[at:at + 8] = Dm.GPS_CA_PREAMBLE
    sat = dict(prn=9, prn_row=8, cn0_dbhz=50.0, doppler_hz=1337.0, code_start=700, phase=0.4, data_bits=data, bit_edge_ms=13)
    x = np.conj(synth.make_scene(t, fs, f_if, n_ms * N, [sat], config_id=77))
    x = x + (6.0 - 4.0j)
    xi8 = synth.to_i8_iq(np.clip(x.real, -127, 127) + 1j * np.clip(x.imag, -127, 127))
    ring = T.MulticastRingBuffer(1 << 23)
    fe = F.DigitalFrontend(f_if, fs, fs)
    for off in range(0, n_ms * N, 1 << 18):
        fe.write_ring(ring, xi8[off:off + (1 << 18)])
    ring.flush()
    base = ring.copy_to_slice(0, n_ms * N)                   # the front-end's output: what both driver runs are fed
    dop = np.arange(-2500.0, 2500.1, 250.0, dtype=np.float32)
    eng = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=dop, n_integrations=M, decision_mode=A.DECIDE_BEST_BIN)
    res = eng.search(base[:M * N])
    r = res[8]
    assert r and r["code_phase_samples"] == 700
    fine = eng.finer_doppler(res)[8]
    eng.close(); fe.close(); ring.close()
    hand_over = dict(r, carrier_freq=fine["freq_hz"])
    ab = R.tracking_ab(base, [hand_over], fs, n_channels=15, write_block=(1 << 18) + 1234)
    assert ab["epochs"][0] == ab["epochs"][1] >= 3080, ab["epochs"]
    assert ab["locked"] == (1, 1) and ab["lost"] == (0, 0)
    _assert_same_states(ab, 15)
    s = ab["async_"][0]
    assert s.active and s.prn == 9 and abs(s.carrier_freq - 1337.0) < 15.0
    assert s.next_sample_index + s.num_samples_per_code > n_ms * N          # every whole code period in the ring was run
    assert not any(ab["async_"][c].active for c in range(1, 15))
    # the ticket loop makes a call per ~planned block instead of one per 10 passes and never waits for a pass: much less wall clock
    assert ab["seconds"][1] < ab["seconds"][0], ab["seconds"]


def test_cpp_tracking_driver_tickets_several_channels_and_a_lost_one(gpu, oracle):
    """Four satellites handed over at different sample indices (one of them a PRN that is NOT in the scene: its channel loses
    lock after MAX_LOST_EPOCHS and reports SatelliteLost), more results than the driver has channels for the last one, blocks
    that end inside code periods.  Ticket loop == synchronous loop, word for word."""
    from gnss_sdr_rs_amd import receiver as R, synth
    t = oracle.ca_code_table()
    fs, N, n_ms = 4_096_000.0, 4096, 160
    truth = {4: (-1730.0, 1111), 11: (640.0, 4000), 23: (2210.0, 77)}
    sats = [dict(prn=p, prn_row=p - 1, cn0_dbhz=49.0, doppler_hz=d, code_start=c, phase=0.1 * p) for p, (d, c) in truth.items()]
    # scaled so that a noise-only prompt (mean power ~2) sits under LOCK_THRESHOLD = 15 (:16) and a tracked one (~170) above it
    x = (1e-3 * synth.to_c32(synth.make_scene(t, fs, 0.0, n_ms * N, sats, config_id=91))).astype(np.complex64)
    results = [dict(prn=p, code_phase_samples=c, code_phase_chips=0.0, carrier_freq=d + 20.0, fs=fs, mag_relative=1.0,
                    sample_global_index=c + k * 3 * N, doppler_bin=0) for k, (p, (d, c)) in enumerate(truth.items())]
    results.append(dict(prn=30, code_phase_samples=5, code_phase_chips=0.0, carrier_freq=100.0, fs=fs, mag_relative=1.0,
                        sample_global_index=2 * N + 5, doppler_bin=0))                   # absent: noise only -> lost
    results.append(dict(prn=31, code_phase_samples=9, code_phase_chips=0.0, carrier_freq=0.0, fs=fs, mag_relative=1.0,
                        sample_global_index=9, doppler_bin=0))                           # a fifth result for four channels: dropped (:353-362)
    ab = R.tracking_ab(x, results, fs, n_channels=4, ring_log2=20, write_block=7 * N + 321)
    assert ab["locked"] == (4, 4) and ab["lost"] == (1, 1)
    assert ab["epochs"][0] == ab["epochs"][1] > 3 * 150
    _assert_same_states(ab, 4)
    for c, (p, (d, _)) in enumerate(truth.items()):
        s = ab["async_"][c]
        assert s.active and s.prn == p and abs(s.carrier_freq - d) < 15.0 and s.lost_counter == 0
    assert not ab["async_"][3].active                                                    # reset() after the loss (:196-201)


def test_cpp_receiver_chain_from_the_stage_drivers(gpu, oracle):
    """main.rs:182-227 as the C++ drop-in wires it: feeder -> DigitalFrontend::write_ring -> device ring -> gnss::run_acquisition
    (signal-time pacing, fine Doppler) on its thread -> gnss::run_tracking (ticket loop) on its thread -> nav bits.  2.5 s of
    int8 IQ at 4.096 Msps, five satellites with data bits: every one is handed over once, tracked on its true Doppler, bit-
    synchronised on its bit edge; the chain runs several times faster than real time (bench.py reports the 16.4 Msps figure)."""
    from gnss_sdr_rs_amd import decoding as Dm, receiver as R, synth
    t = oracle.ca_code_table()
    fs, N, f_if, n_ms = 4_096_000.0, 4096, 512_000.0, 2500            # f_if / fs * 2048 = 256: the LUT NCO is exact
    rng = np.random.default_rng(3)
    sats = []
    for i, prn in enumerate((3, 8, 14, 21, 27)):
        data = rng.integers(0, 2, 140) * 2 - 1
        data[5 + i:13 + i] = Dm.GPS_CA_PREAMBLE
        sats.append(dict(prn=prn, prn_row=prn - 1, cn0_dbhz=49.0, doppler_hz=float(rng.uniform(-3000, 3000)), code_start=int(rng.integers(0, N)),
                         phase=0.2 * i, data_bits=data, bit_edge_ms=int(rng.integers(0, 20))))
    x = np.conj(synth.make_scene(t, fs, f_if, n_ms * N, sats, config_id=93)) + (4.0 - 2.0j)
    xi8 = synth.to_i8_iq(np.clip(x.real, -127, 127) + 1j * np.clip(x.imag, -127, 127))
    rep = R.receiver_run(xi8, fs, f_if, freq_search_hz=8e3, freq_step_hz=500.0, block_samples=1 << 17, ring_log2=22, warmup_calls=4)
    truth = {s["prn"]: s for s in sats}
    chans = [c for c in rep["channels"] if c["prn"]]
    assert sorted(c["prn"] for c in chans) == sorted(truth), rep
    for c in chans:
        s = truth[c["prn"]]
        assert c["active"] and c["lost_counter"] == 0 and abs(c["carrier_freq"] - s["doppler_hz"]) < 25.0, c
        # the channel's epoch 0 is the code period it was handed over at: the bit edge is seen relative to that
        assert (c["start_index"] - s["code_start"]) % N == 0
        m0 = (c["start_index"] - s["code_start"]) // N
        assert c["bit_sync"] and c["frame_sync_ind"] == (s["bit_edge_ms"] - m0) % 20, (c, s["bit_edge_ms"], m0)
        assert c["epochs"] >= n_ms - 80, c                             # handed over at the first round (10 ms + a block), tracked to the end
    assert rep["dwells"] >= 2 and rep["channels_started"] == 5
    assert rep["channel_epochs"] == sum(c["epochs"] for c in chans)
    assert rep["signal_seconds"] / rep["wall_seconds"] > 3.0, rep
