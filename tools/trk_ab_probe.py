"""Diagnostic: host/receiver_harness.cpp's gmrx_tracking_ab (gnss::run_tracking, synchronous loop against ticket loop) several times on the
four-channel scene of tests/test_gpu_stage_drivers.py; prints every field of a channel record in which the two loops differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gnss_sdr_rs_amd import _lib, acquisition as A, receiver as R, synth
_lib.init(0)
t = A.ca_code_table()
fs, N, n_ms = 4_096_000.0, 4096, 160
truth = {4: (-1730.0, 1111), 11: (640.0, 4000), 23: (2210.0, 77)}
sats = [dict(prn=p, prn_row=p - 1, cn0_dbhz=49.0, doppler_hz=d, code_start=c, phase=0.1 * p) for p, (d, c) in truth.items()]
x = (1e-3 * synth.to_c32(synth.make_scene(t, fs, 0.0, n_ms * N, sats, config_id=91))).astype(np.complex64)
results = [dict(prn=p, code_phase_samples=c, code_phase_chips=0.0, carrier_freq=d + 20.0, fs=fs, mag_relative=1.0,
                sample_global_index=c + k * 3 * N, doppler_bin=0) for k, (p, (d, c)) in enumerate(truth.items())]
results.append(dict(prn=30, code_phase_samples=5, code_phase_chips=0.0, carrier_freq=100.0, fs=fs, mag_relative=1.0,
                    sample_global_index=2 * N + 5, doppler_bin=0))
ref = None
for it in range(int(os.environ.get("AB_RUNS", "6"))):
    ab = R.tracking_ab(x, results, fs, n_channels=4, ring_log2=20, write_block=7 * N + 321)
    print("run", it, "epochs", ab["epochs"], "locked", ab["locked"], "lost", ab["lost"], "seconds", ab["seconds"])
    for c in range(4):
        a, b = ab["sync"][c], ab["async_"][c]
        for k, _ in a._fields_:
            va, vb = getattr(a, k), getattr(b, k)
            if k != "reserved" and va != vb:
                print("   ch", c, k, "sync", va, "tickets", vb)
    cur = [[getattr(s, k) for k, _ in s._fields_ if k != "reserved"] for s in ab["sync"]]
    if ref is not None and cur != ref:
        print("   the SYNCHRONOUS loop differs from its own first run")
    ref = ref or cur
