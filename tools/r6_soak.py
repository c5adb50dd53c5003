"""Soak of round 6's new paths on one GPU: the receiver chain from the C++ stage drivers N times (speculative front-end, ticket loop,
signal-time acquisition), the packed five-arm tracking launch M times with the same answer every time, and the sync-vs-ticket A/B.
Every result must repeat exactly; any time-out of the persistent kernel's exchange surfaces as GmError."""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gnss_sdr_rs_amd import _lib, acquisition as A, decoding as Dm, receiver as R, tracking as T, synth
_lib.init(0)
synth.DEFAULT_GENERATOR = "xoshiro"
t0 = time.time()
ca = A.ca_code_table()
# ---- receiver chain, 1.6 s at 16.3676 Msps, N times
fs, f_if, N = 16_367_600.0, 4_130_400.0, 16368
rng = np.random.default_rng(11)
sats = []
for i, (prn, cn0) in enumerate([(2, 50.0), (5, 48.0), (9, 47.0), (13, 46.0), (17, 46.0), (22, 45.0)]):
    data = rng.integers(0, 2, 100) * 2 - 1
    sats.append(dict(prn=prn, prn_row=prn - 1, cn0_dbhz=cn0, doppler_hz=float(rng.uniform(-5500, 5500)), code_start=int(rng.integers(0, N)), phase=0.3 * i,
                     data_bits=data, bit_edge_ms=int(rng.integers(0, 20))))
x = np.conj(synth.make_scene(ca, fs, f_if, 1600 * N, sats, config_id=12)) + (5.0 - 3.0j)
xi8 = synth.to_i8_iq(np.clip(x.real, -127, 127) + 1j * np.clip(x.imag, -127, 127)); del x
ref = None
for it in range(int(os.environ.get("SOAK_RX", "12"))):
    rep = R.receiver_run(xi8, fs, f_if, warmup_calls=4 if it else 48)
    key = sorted((c["prn"], c["active"], c["bit_sync"], c["frame_sync_ind"]) for c in rep["channels"] if c["prn"])
    locked = sum(1 for c in rep["channels"] if c["prn"] and c["active"])
    print("rx %2d: %.1f x real time, %d channels, %d active, front-end runs done again %d, dwells %d" % (it, rep["signal_seconds"] / rep["wall_seconds"], len(key), locked, rep["fe_runs_repaired"], rep["dwells"]), flush=True)
    assert len(key) == 6 and locked == 6, key
    assert [k[0] for k in key] == [2, 5, 9, 13, 17, 22]
# ---- packed five-arm launch, M times: identical states every time
fs5, L, rate, C, periods = 50.0e6, 4092, 1.023e6, 36, 6
n = 200000
codes = np.where(rng.integers(0, 2, (C, L)) > 0, 1, -1).astype(np.int8)
t1p = np.arange(n, dtype=np.float64)
cp = (t1p * rate / fs5) % L
sub = np.where((cp - np.floor(cp)) < 0.5, 1.0, -1.0).astype(np.float32)
ci = np.floor(cp).astype(np.int64)
xx = (rng.standard_normal((periods + 1) * n) + 1j * rng.standard_normal((periods + 1) * n)).astype(np.complex64) * np.float32(8.0)
dopp = rng.uniform(-2000, 2000, C)
xv = xx.reshape(periods + 1, n)
for c in range(C):
    one = (np.float32(0.6) * codes[c][ci] * sub * np.exp(2j * np.pi * dopp[c] * t1p / fs5)).astype(np.complex64)
    xv += one[None, :] * (np.exp(2j * np.pi * dopp[c] * n / fs5) ** np.arange(periods + 1))[:, None].astype(np.complex64)
ring = T.MulticastRingBuffer(1 << 21); ring.write_samples(xx)
mgr = T.TrackingManager(fs5, n_channels=C, n_arms=5, code_index_mode=T.CODE_INDEX_FIXED, early_late_space=0.25, very_early_late_space=0.6, boc11=True,
                        codes=codes, nominal_code_rate=rate)
ref = None
for it in range(int(os.environ.get("SOAK_TRK", "200"))):
    for j in range(C):
        mgr.channels[j].start(dict(prn=j + 1, code_phase_samples=0, code_phase_chips=0.0, carrier_freq=float(dopp[j]) + 10.0, fs=fs5, mag_relative=1.0,
                                   sample_global_index=0, doppler_bin=0))
        mgr.channels[j].set_state(code_rate=rate, num_samples_per_code=n, carrier_phase=0.0, code_error=0.0, carrier_error=0.0, lost_counter=0)
    outs, proc, lost, done = mgr.update_all(ring, periods)
    h = hashlib.sha256(outs.tobytes() + proc.tobytes() + lost.tobytes() + b"".join(bytes(s) for s in mgr.get_states())).hexdigest()
    ref = ref or h
    assert h == ref and done == periods and proc.all() and not lost.any(), (it, done)
print("packed five-arm launch: %d launches of %d periods, identical words every time" % (it + 1, periods))
print("soak done in %.0f s" % (time.time() - t0))
