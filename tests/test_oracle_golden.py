"""CPU: pin the oracle (oracle/) against every known answer the reference's own tests and constants hold
(SURVEY.md §8c-4).  These run without a GPU."""
import hashlib

import numpy as np
import pytest

from conftest import golden


def test_ca_code_prn1_known_answer(oracle):
    # src/bk/gps_ca_prn.rs:72-124 (test_prn_code)
    g = golden("ca_code_known_answers.json")
    assert (oracle.ca_code_row(0) == np.array(g["prn1_chips"], np.int8)).all()


def test_ca_code_table_digest_and_first10(oracle):
    # src/constants/gps_ca_constants.rs:1-1346 — all 32 rows, by digest, packed bits and IS-GPS-200 octals
    g = golden("ca_code_known_answers.json")
    t = oracle.ca_code_table()
    assert t.shape == (32, 1023) and set(np.unique(t)) == {-1, 1}
    assert hashlib.sha256(t.tobytes()).hexdigest() == g["table_sha256"]
    bits = np.unpackbits(np.frombuffer(bytes.fromhex(g["table_bits_hex"]), np.uint8).reshape(32, 128), axis=1)[:, :1023]
    assert ((t > 0).astype(np.uint8) == bits).all()
    for r in range(32):
        v = 0
        for b in (t[r, :10] > 0):
            v = (v << 1) | int(b)
        assert oct(v)[2:] == g["first10_octal"][r]
    assert g["first10_octal"][0] == "1440" and g["first10_octal"][31] == "1712"   # IS-GPS-200 Table 3-Ia


def test_ca_code_row_out_of_range(oracle):
    with pytest.raises(IndexError):
        oracle.ca_code_row(32)      # GPS_CA_CODE_32_PRN[32] panics in the reference


def test_generate_ca_code_samples(oracle):
    # ca_code.rs:12-27: n = round(fs/(rate/1023)), idx = floor(i*rate/fs)
    for fs, n in [(16_367_600.0, 16368), (8.0e6, 8000), (4_096_000.0, 4096), (2.048e6, 2048), (25.0e6, 25000)]:
        s = oracle.generate_ca_code_samples(3, 1.023e6, fs)
        assert s.size == n
        i = np.arange(n, dtype=np.float32)
        idx = np.floor((i * np.float32(1.023e6)) / np.float32(fs)).astype(np.int64)
        assert (s == oracle.ca_code_row(2)[idx]).all()
    with pytest.raises(IndexError):
        oracle.generate_ca_code_samples(0, 1.023e6, 8e6)     # prn as usize - 1 underflows
    with pytest.raises(IndexError):
        oracle.generate_ca_code_samples(33, 1.023e6, 8e6)


def test_acquisition_manager_known_answers(oracle):
    # do_acquisition.rs:339-395
    g = golden("manager_known_answers.json")
    m = oracle.AcquisitionManager()
    assert m.mode == m.COLD
    names = {"ColdStart": m.COLD, "WarmStart": m.WARM, "SteadyState": m.STEADY}
    for n in ("3", "5", "0"):
        m.update_mode(int(n))
        assert m.mode == names[g["mode_for_tracked"][n]]
    m = oracle.AcquisitionManager()
    assert m.get_pacing_and_list(set(g["cold_start"]["active"])) == (g["cold_start"]["interval_ms"], g["cold_start"]["mask"])
    m.update_mode(g["warm_start"]["update_mode"])
    assert m.get_pacing_and_list(set(g["warm_start"]["active"])) == (g["warm_start"]["interval_ms"], g["warm_start"]["mask"])
    m.update_mode(7)   # steady: 2000 ms, first 5 inactive
    assert m.get_pacing_and_list({1, 3}) == (2000, 0b1111010)


def test_ring_buffer_vectors(oracle):
    # multicast_ring_buffer.rs:147-209
    g = golden("ring_buffer_vectors.json")
    rb = oracle.MulticastRingBuffer(g["buf_size"])
    rng = lambda a: np.arange(a[0], a[1]).astype(np.complex64)
    for st in g["steps"]:
        rb.write_samples(rng(st["write"]))
        assert rb.get_head() == st["head"]
        raw = rb.raw()
        for key, sl in (("buffer_1020_1024", slice(1020, 1024)), ("buffer_0_6", slice(0, 6)), ("buffer_6_16", slice(6, 16))):
            if key in st:
                assert (raw[sl] == rng(st[key])).all()
        if "copy_to_slice" in st:
            c = st["copy_to_slice"]
            assert (rb.copy_to_slice(c["start"], c["n"]) == rng(c["expect"])).all()
    with pytest.raises(AssertionError):
        oracle.MulticastRingBuffer(1000)   # "Buffer size must be a power of two"


def test_loop_filter_constants(oracle):
    # do_tracking.rs:16-28, 60-64
    g = golden("loop_filter_constants.json")
    for k in ("pll", "dll"):
        f = oracle.loop_filter_new(g[k]["bw"], g[k]["zeta"], g[k]["gain"])
        assert f.tau1 == pytest.approx(g[k]["tau1"], rel=0, abs=0) and f.tau2 == pytest.approx(g[k]["tau2"], rel=0, abs=0)
    assert g["pll"]["tau1"] == pytest.approx(1.117551e-4, rel=1e-6) and g["dll"]["tau1"] == pytest.approx(0.06984694, rel=1e-6)


def test_oracle_fft_against_float64(oracle):
    # rustfft semantics: forward e^{-j..}, inverse e^{+j..}, unnormalised (do_acquisition.rs:137,182,188)
    rng = np.random.default_rng(1)
    for n in (8, 100, 1023, 2048, 4096, 8000, 16368):
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        ref = np.fft.fft(x.astype(np.complex128))
        assert np.linalg.norm(oracle.fft(x) - ref) / np.linalg.norm(ref) < 4e-7
        refi = np.fft.ifft(x.astype(np.complex128)) * n
        assert np.linalg.norm(oracle.fft(x, inverse=True) - refi) / np.linalg.norm(refi) < 4e-7
    r = rng.standard_normal(64).astype(np.float32)
    assert np.allclose(oracle.rfft(r), np.fft.rfft(r.astype(np.float64)), atol=1e-4)


def test_doppler_table_and_apply(oracle):
    # doppler_shift.rs:10-58 against a float32 numpy restatement of the same formulas
    f = np.float32
    t = oracle.DopplerShiftTable(4_130_400.0, -7000.0, 16_367_600.0, 16368)
    assert t.doppler_freq_hz == float(f(4_130_400.0) + f(-7000.0))          # stores IF + Doppler (:20)
    step = f(f(f(2.0) * f(np.pi)) * f(t.doppler_freq_hz)) / f(16_367_600.0)
    ph = np.arange(16368, dtype=np.float32) * step
    # glibc cosf/sinf vs numpy's float32 kernels can differ in the last bit
    assert np.max(np.abs(t.table.real - np.cos(ph))) < 2e-7 and np.max(np.abs(t.table.imag + np.sin(ph))) < 2e-7
    rng = np.random.default_rng(2)
    s = (rng.integers(-127, 128, 1003) + 1j * rng.integers(-127, 128, 1003)).astype(np.complex64)
    out = np.full(1003, 99 + 99j, np.complex64)
    oracle.apply_doppler_shift(s, t.table[:1003], out)
    a, b, c, d = s.real, s.imag, t.table[:1003].real, t.table[:1003].imag
    re = (a * c).astype(f) + (-(b * d).astype(f))
    im = (a * d).astype(f) + (b * c).astype(f)
    assert (out[:1000].real == re[:1000]).all() and (out[:1000].imag == im[:1000]).all()
    assert (out[1000:] == 99 + 99j).all()      # only 4*floor(n/4) elements are written (:26)


def test_is_good_satellite_lane_order(oracle):
    # do_acquisition.rs:229-238: 8 lane sums over chunks_exact(8), then an ordered reduce
    rng = np.random.default_rng(3)
    p = (rng.random(8003) * 1e6).astype(np.float32)
    lanes = np.zeros(8, np.float32)
    for c in range(8003 // 8):
        lanes = (lanes + p[c * 8:c * 8 + 8]).astype(np.float32)
    s = np.float32(-0.0)
    for l in lanes:
        s = np.float32(s + l)
    ok, got = oracle.is_good_satellite(p, float(p.max()))
    assert got == float(s)
    avg = np.float32(np.float32(s - p.max()) / np.float32(8002))
    assert ok == bool(np.float32(p.max() / avg) > 7.0)


def test_process_channels_equals_per_channel_update():
    """orc_trk_process_channels (the rayon fan-out of do_tracking.rs:364-371 as an OpenMP loop, used by bench.py's
    tracking cpu_baseline) leaves exactly the state that per-channel update() calls leave."""
    import numpy as np
    from oracle import oracle as O
    from gnss_sdr_rs_amd import synth
    ca = O.ca_code_table()
    fs = 4.0e6
    sc = synth.tracking_scene(ca, fs, 0.0, [3, 7, 12], 8, config_id=9, cn0=50.0)
    ring = O.MulticastRingBuffer(1 << 16)
    ring.write_samples(synth.to_c32(sc["x"]))

    def fresh():
        out = []
        for i, s in enumerate(sc["sats"]):
            ch = O.TrackingChannel(i, fs, code_index_mode=O.CODE_INDEX_FIXED)
            ch.start(dict(prn=s["prn"], code_phase_samples=0, code_phase_chips=0.0, carrier_freq=s["doppler_hz"] + 15.0, fs=fs,
                          mag_relative=1.0, sample_global_index=s["code_start"]))
            out.append(ch)
        return out
    a, b = fresh(), fresh()
    done = O.process_channels(a, ring, 100, n_threads=3)
    n_seq = 0
    for ch in b:
        while ch.update(ring)[0] > 0:
            n_seq += 1
    assert done == n_seq and done >= 3 * 5
    for x, y in zip(a, b):
        for f in ("carrier_freq", "carrier_phase", "code_phase", "code_rate", "i_prompt", "q_prompt", "next_sample_index",
                  "lost_counter"):
            assert getattr(x.c, f) == getattr(y.c, f), f


def test_frontend_oracle_vs_numpy_float32_twin():
    """The C restatement of rf/frontend.rs:33-62 against an independent numpy float32 twin written from the same lines
    (the reference has no test or vector for the front-end: parity unpinned beyond this)."""
    import numpy as np
    from oracle import oracle as O
    f32 = np.float32
    f_if, fs = f32(4.1304e6), f32(16.3676e6)
    fe = O.DigitalFrontend(float(f_if), float(fs), float(fs))
    ang = (f32(2.0) * f32(np.pi)) * np.arange(2048, dtype=f32) / f32(2048)
    assert np.abs(np.array(fe.s.lut_re[:]) - np.cos(ang.astype(np.float64))).max() < 1e-7
    assert np.abs(np.array(fe.s.lut_im[:]) + np.sin(ang.astype(np.float64))).max() < 1e-7
    step = (f_if / fs) * f32(2048)
    assert f32(fe.s.phase_step) == step
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(16 * 300 + 5) * 10 + 2).astype(f32)
    want = x.copy()
    lre, lim = np.array(fe.s.lut_re[:], f32), np.array(fe.s.lut_im[:], f32)
    bre, bim, phase = np.zeros(8, f32), np.zeros(8, f32), f32(0)
    alpha = f32(0.001)
    con = f32(1.0) - alpha
    for c in range(0, x.size - 15, 16):
        re, im = want[c:c + 16:2].copy(), want[c + 1:c + 16:2].copy()
        bre = bre * con + re * alpha
        bim = bim * con + im * alpha
        re, im = re - bre, im - bim
        idx = np.zeros(8, np.int64)
        for j in range(8):
            idx[j] = int(phase) % 2048 if phase > 0 else 0
            phase = f32(np.fmod(f32(phase + step), f32(2048)))
        lc, ls = lre[idx], lim[idx]
        want[c:c + 16:2] = re * lc + im * ls
        want[c + 1:c + 16:2] = re * ls - im * lc
    got = fe.process_block(x.copy())
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    assert f32(fe.s.phase_accumulator) == phase and (np.array(fe.s.bias_re[:], f32) == bre).all()


def _regression_vectors():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "restatement_vectors.json")))


def _regen_acq(O, synth, case):
    import numpy as np
    t = O.ca_code_table()
    x = synth.to_c32(synth.make_scene(t, case["fs"], case["f_if"], case["M"] * case["N"], case["sats"], config_id=case["config_id"]))
    tables = [O.DopplerShiftTable(case["f_if"], float(d), case["fs"], case["N"]) for d in case["doppler_hz"]]
    return x, tables


def test_oracle_reproduces_its_committed_regression_vectors():
    """tests/golden/restatement_vectors.json (SURVEY §8 c4's list: Doppler tables, mix output, per-(p, d) metrics of two small
    scenes, AcquisitionResults, 5 tracking epochs in both code-index modes) — outputs of THIS restatement frozen at commit
    time, regenerated here bit for bit from the seeds.  Guards the oracle against accidental change."""
    import hashlib
    import numpy as np
    from oracle import oracle as O
    from gnss_sdr_rs_amd import synth
    vec = _regression_vectors()
    bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32).tolist()
    for key in ("acq_scene_2048", "acq_scene_8000"):
        case = vec[key]
        x, tables = _regen_acq(O, synth, case)
        if key == "acq_scene_2048":
            assert [hashlib.sha256(tables[i].table.tobytes()).hexdigest() for i in (1, 2, 3)] == vec["doppler_tables"]["sha256"]
            mixed = np.zeros(2048, np.complex64)
            O.apply_doppler_shift(x[:2048], tables[3], mixed)
            assert hashlib.sha256(mixed.tobytes()).hexdigest() == vec["mix"]["sha256"]
        for prn, w in zip(case["prns"], case["workers"]):
            exp, (bmax, barg, bsum, _) = O.AcquisitionWorker(prn, case["N"], case["fs"]).search_satellite(
                x, tables, 1000, case["M"], want_planes=True, no_early_exit=True)
            assert bits(bmax) == w["max_bits"] and np.asarray(barg).tolist() == w["argmax"] and bits(bsum) == w["sum_bits"]
            assert exp == w["result"]
    trk = vec["tracking"]
    t = O.ca_code_table()
    for mode in (0, 1):
        chans = trk["modes"][str(mode)]
        prns = [c["prn"] for c in chans]
        rows = [p if mode == 0 else p - 1 for p in prns]
        sc = synth.tracking_scene(t, trk["fs"], 0.0, prns, 7, config_id=trk["config_id"], cn0=50.0, code_rows=rows)
        ring = O.MulticastRingBuffer(1 << 16)
        ring.write_samples(synth.to_c32(sc["x"])[:6 * trk["n"] + 4000])
        for i, c in enumerate(chans):
            ch = O.TrackingChannel(i, trk["fs"], code_index_mode=mode)
            ch.start(dict(prn=c["prn"], code_phase_samples=0, code_phase_chips=0.0, carrier_freq=c["doppler_hz"] + 25.0,
                          fs=trk["fs"], mag_relative=1.0, sample_global_index=c["code_start"]))
            for ep in c["epochs"]:
                rc, out6, _ = ch.update(ring)
                assert rc == 1 and bits(out6) == ep["out_bits"]
                assert bits([ch.c.carrier_freq, ch.c.code_rate, ch.c.carrier_phase, ch.c.code_phase]) == ep["state_bits"]
                assert int(ch.c.next_sample_index) == ep["next_sample_index"]
