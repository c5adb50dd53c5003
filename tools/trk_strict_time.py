"""tools/trk_strict_time.py — the strict tracking modes (gm_trk_cfg.strict_libm / strict_sum_order) at BASELINE configs[2]'s shape:
40 passes of 32 channels x 25 Msps, wall-clock per pass for the four combinations.  (Run it plainly: under rocprofv3 the process
printed its four lines and then never exited on this pool — every handle is closed explicitly below since.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gnss_sdr_rs_amd import tracking as T, synth, acquisition as A

fs, n, C, E = 25.0e6, 25000, 32, 40
ca = A.ca_code_table() if hasattr(A, "ca_code_table") else None
sc = synth.tracking_scene(ca, fs, 0.0, list(range(1, 33)), E + 2, config_id=3, cn0=47.0)
ring = T.MulticastRingBuffer(1 << 21)
ring.write_samples(synth.to_c32(sc["x"]))
for kw in (dict(), dict(strict_libm=True), dict(strict_libm=True, strict_sum_order=True), dict(strict_sum_order=True)):
    mgr = T.TrackingManager(fs, n_channels=C, code_index_mode=T.CODE_INDEX_FIXED, **kw)
    ts = []
    for rep in range(3):
        for i in range(C):
            s = sc["sats"][i % 32]
            mgr.channels[i].start(dict(prn=s["prn"], code_phase_samples=0, code_phase_chips=0.0, carrier_freq=s["doppler_hz"] + 20.0,
                                       fs=fs, mag_relative=1.0, sample_global_index=s["code_start"], doppler_bin=0))
        mgr.synchronize()
        t0 = time.perf_counter()
        mgr.update_all_dev(ring, E)
        mgr.synchronize()
        ts.append(time.perf_counter() - t0)
    print(kw, "us per epoch: %.1f" % (min(ts) / E * 1e6), flush=True)
    mgr.close()
ring.close()
