"""tools/trk_host_time.py — what the HOST-VISIBLE tracking entry costs per call (gm_trk_update_all: one persistent launch, a stream
synchronisation, three device-to-host copies, through the ctypes wrapper): 67 us for one pass of 32 channels x 25 Msps, 98 us for
ten (9.8 us per epoch against 3.1 us of device time) — three orders of magnitude inside the 1 ms real-time budget of an epoch."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gnss_sdr_rs_amd import tracking as T, synth, acquisition as A
fs, n, C = 25.0e6, 25000, 32
E = 400
ca = A.ca_code_table()
sc = synth.tracking_scene(ca, fs, 0.0, list(range(1, 33)), E + 2, config_id=3, cn0=47.0)
ring = T.MulticastRingBuffer(1 << 24)
ring.write_samples(synth.to_c32(sc["x"]))
mgr = T.TrackingManager(fs, n_channels=C, code_index_mode=T.CODE_INDEX_FIXED)
for i in range(C):
    s = sc["sats"][i % 32]
    mgr.channels[i].start(dict(prn=s["prn"], code_phase_samples=0, code_phase_chips=0.0, carrier_freq=s["doppler_hz"] + 20.0,
                               fs=fs, mag_relative=1.0, sample_global_index=s["code_start"], doppler_bin=0))
for per in (1, 10):
    mgr.update_all(ring, per)
    t0 = time.perf_counter(); k = 0
    for _ in range(15):
        mgr.update_all(ring, per); k += 1
    dt = (time.perf_counter() - t0) / k
    print("update_all(%d epochs): %.1f us per call, %.1f us per epoch" % (per, dt * 1e6, dt * 1e6 / per), flush=True)
mgr.close(); ring.close()
