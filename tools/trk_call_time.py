"""Diagnostic: what one gm_trk_update_all call costs on the host beside its kernel, alone and with a front-end block in flight on
another ring's copy stream (the receiver leg's situation).  Prints the median wall clock of launch, synchronisation and the whole
call, and the kernel's own duration from the handle's HIP events."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnss_sdr_rs_amd import _lib, acquisition as A, tracking as T, frontend as F, synth
_lib.init(0)
synth.DEFAULT_GENERATOR = "xoshiro"
fs, f_if, N = 16_367_600.0, 4_130_400.0, 16368
ca = A.ca_code_table()
rng = np.random.default_rng(3)
sats = [dict(prn=p, prn_row=p - 1, cn0_dbhz=48.0, doppler_hz=float(rng.uniform(-4000, 4000)), code_start=int(rng.integers(0, N)), phase=0.1 * p)
        for p in (2, 5, 9, 13, 17, 22, 26, 30)]
n_ms = 900
x = np.conj(synth.make_scene(ca, fs, f_if, n_ms * N, sats, config_id=12))
xi8 = synth.to_i8_iq(np.clip(x.real, -127, 127) + 1j * np.clip(x.imag, -127, 127)); del x
ring = T.MulticastRingBuffer(1 << 24)
fe = F.DigitalFrontend(f_if, fs, fs)
fe.write_ring(ring, xi8); ring.flush()
dop = np.arange(-7000.0, 7000.1, 500.0, dtype=np.float32)
eng = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=dop, n_integrations=10, decision_mode=A.DECIDE_BEST_BIN)
mgr = T.TrackingManager(fs, n_channels=15, code_index_mode=T.CODE_INDEX_FIXED)
# channels start on a snapshot ending 20 ms into the stream: acquisition results need a head there, so a second small ring
r0 = T.MulticastRingBuffer(1 << 19); f0 = F.DigitalFrontend(f_if, fs, fs)
f0.write_ring(r0, xi8[:20 * N]); r0.flush()
res, _ = eng.search_ring(r0)
fine = eng.finer_doppler(res)
ch = 0
for r, f_ in zip(res, fine):
    if r and f_:
        mgr.channels[ch].start(dict(r, carrier_freq=f_["freq_hz"])); ch += 1
print("channels started:", ch)
mgr.enable_timing(True)

def run(tag, busy):
    oring = T.MulticastRingBuffer(1 << 20); ofe = F.DigitalFrontend(f_if, fs, fs)
    ofe.write_ring(oring, xi8[:1 << 18]); oring.flush()
    rows = []
    for i in range(12):
        if busy:
            ofe.write_ring(oring, xi8[:1 << 18])          # a 16 ms block in flight on ANOTHER ring's copy stream
        t0 = time.perf_counter()
        mgr.update_all_dev(ring, 16)
        t1 = time.perf_counter()
        mgr.synchronize()
        t2 = time.perf_counter()
        ms, _ = mgr.last_timing()
        t3 = time.perf_counter()
        o = mgr.update_all(ring, 16)
        t4 = time.perf_counter()
        rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, ms, (t4 - t3) * 1e3, int(o[1].sum())))
        oring.flush()
    d = np.array(rows[2:])
    print("%-22s launch %.3f ms, synchronise %.3f ms, kernel (events) %.3f ms | whole update_all call %.3f ms, channel-epochs %d"
          % (tag, *np.median(d[:, :4], axis=0), int(np.median(d[:, 4]))))
    ofe.close(); oring.close()

run("tracking alone", False)
run("front-end in flight", True)
run("tracking alone", False)
