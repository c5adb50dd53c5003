"""Soak of the persistent tracking kernel: 32 channels, fs = 2.048 MHz, 3 s of signal in 4095-epoch launches, twice over.
Checks: no exchange time-out (gm_trk_synchronize raises), every channel processed every epoch it could, lock kept."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gnss_sdr_rs_amd import _lib, acquisition as A, synth, tracking as T
_lib.init(0)
ca = A.ca_code_table()
fs, n, n_ms = 2.048e6, 2048, 3000
prns = list(range(1, 33))
t0 = time.time()
sc = synth.tracking_scene(ca, fs, 0.0, prns, n_ms + 2, config_id=5, cn0=47.0)
x = synth.to_c32(sc["x"])
print("scene %.1f s" % (time.time() - t0), flush=True)
ring = T.MulticastRingBuffer(1 << 23)
ring.write_samples(x)
for rep in range(2):
    mgr = T.TrackingManager(fs, n_channels=32, code_index_mode=T.CODE_INDEX_FIXED)
    for i, s in enumerate(sc["sats"]):
        mgr.channels[i].start(dict(prn=s["prn"], code_phase_samples=0, code_phase_chips=0.0, carrier_freq=s["doppler_hz"] + 15.0,
                                   fs=fs, mag_relative=1.0, sample_global_index=s["code_start"], doppler_bin=0))
    t0 = time.time()
    mgr.update_all_dev(ring, n_ms)            # 3000 passes in one persistent launch
    mgr.synchronize()                          # raises on an exchange time-out
    dt = time.time() - t0
    ok = sum(1 for c in mgr.channels if c.is_active() and c.lost_counter == 0 and
             abs(c.carrier_freq - sc["sats"][c.id]["doppler_hz"]) < 20.0)
    nxt = min(c.next_sample_index for c in mgr.channels)
    print("rep %d: %.1f ms for %d passes (%.2f us per pass), %d / 32 channels locked, min next_sample_index %d of %d" %
          (rep, dt * 1e3, n_ms, dt / n_ms * 1e6, ok, nxt, x.size), flush=True)
    assert ok == 32 and nxt > (n_ms - 2) * n
    mgr.close()
print("soak ok")
