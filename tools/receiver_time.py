"""Diagnostic: bench.py's `receiver` leg alone (the whole chain of SURVEY §8 f1: feeder -> front-end -> ring -> acquisition + fine
Doppler -> tracking -> nav bits); RX_MS = milliseconds of signal (default 2200), RX_CPU=1 adds the CPU oracle chain,
RX_DIAG=1 the per-block split of the tracking call (kernel time by HIP events against the call's wall clock)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from gnss_sdr_rs_amd import _lib, acquisition as A, tracking as T, synth
_lib.init(0)
synth.DEFAULT_GENERATOR = "xoshiro"
diag = []
if os.environ.get("RX_DIAG") == "1":
    import time
    _ua = T.TrackingManager.update_all
    def timed_update_all(self, ring, max_epochs=1):
        self.enable_timing(True)
        t0 = time.perf_counter()
        r = _ua(self, ring, max_epochs)
        wall = time.perf_counter() - t0
        ms, _ = self.last_timing()
        diag.append((wall * 1e3, ms, int(r[1].sum()), int(r[1].any(axis=1).sum())))
        return r
    T.TrackingManager.update_all = timed_update_all
out = bench.receiver_leg(A.ca_code_table(), A, T, synth, os.environ.get("RX_CPU") == "1", n_ms=int(os.environ.get("RX_MS", "2200")))
print(json.dumps(out, indent=1))
if diag:
    import numpy as np
    d = np.array(diag[2:])
    print("tracking calls: %d; per call median wall %.3f ms, kernel %.3f ms, channel-epochs %.0f, epochs %.0f; kernel us per epoch %.2f"
          % (len(d), np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 2]), np.median(d[:, 3]), 1e3 * d[:, 1].sum() / max(d[:, 3].sum(), 1)))
