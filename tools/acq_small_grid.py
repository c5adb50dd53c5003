"""Latency of small acquisition grids (few workers): the reference's own case is ONE PRN x 29 bins x 16368 phases."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import numpy as np
from gnss_sdr_rs_amd import _lib, acquisition as A, synth
_lib.init(0)
ca = A.ca_code_table()
cap = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "capture_config.json")))
sc = synth.cfg1_scene(ca, cap)
x = synth.to_i8_real(sc["x"])
for prns in ([sc["sats"][0]["prn"]], [s["prn"] for s in sc["sats"][:4]], list(range(1, 13))):
    eng = A.AcquisitionEngine(sc["fs"], sc["f_if"], sc["N"], doppler_hz=sc["doppler_hz"], n_integrations=sc["M"], prn_ids=prns)
    eng.search(x)
    eng.enable_timing(1)
    t0 = time.perf_counter()
    for _ in range(20):
        res = eng.search(x)
    dt = (time.perf_counter() - t0) / 20
    tm = eng.last_timing()
    mix, corr = tm["mix_fft_ms"], tm["corr_ms"]
    print(len(prns), "PRN: host-call %.3f ms  corr kernel %.3f ms  mix %.3f ms  found %s" % (dt * 1e3, corr, mix, [r["prn"] for r in res if r]), flush=True)
    eng.close()
