"""Diagnostic: bench.py's `receiver` leg alone (the whole chain of SURVEY §8 f1: feeder -> front-end -> ring -> acquisition + fine
Doppler -> tracking -> nav bits); RX_MS = milliseconds of signal (default 2200), RX_CPU=1 adds the CPU oracle chain,
RX_DIAG=1 the per-block split of the tracking call (kernel time by HIP events against the call's wall clock)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_rt = os.environ.get("RX_TORCH", "0")      # diagnostics: what the presence of PyTorch in the process costs the loop (1: imported, 2: CUDA initialised, 3: + one thread)
if _rt != "0":
    import torch
    if _rt in ("2", "3"):
        torch.cuda.init(); _s = torch.cuda.Stream(); _x = torch.zeros(10, device="cuda")
    if _rt == "3":
        torch.set_num_threads(1)
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
calls = {}
if os.environ.get("RX_CALLS") == "1":      # wall clock of every library call of the loop, by entry
    import time as _t
    def _wrap(cls, name):
        f = getattr(cls, name)
        def g(self, *a, **k):
            t0 = _t.perf_counter(); r = f(self, *a, **k); calls.setdefault(cls.__name__ + "." + name, []).append(_t.perf_counter() - t0); return r
        setattr(cls, name, g)
    from gnss_sdr_rs_amd import frontend as _F, decoding as _D
    for cls, names in ((T.TrackingManager, ("update_all_async", "collect")), (T.MulticastRingBuffer, ("get_head", "flush")), (_F.DigitalFrontend, ("write_ring",)),
                       (A.AcquisitionEngine, ("search_ring", "finer_doppler")), (_D.NavSyncStatus, ("update_many",))):
        for n in names:
            _wrap(cls, n)
out = bench.receiver_leg(A.ca_code_table(), A, T, synth, os.environ.get("RX_CPU") == "1", n_ms=int(os.environ.get("RX_MS", "2200")))
print(json.dumps(out, indent=1))
if calls:
    import numpy as np
    for k, v in calls.items():
        v = np.array(v) * 1e6
        slow = [(i, round(float(x))) for i, x in enumerate(v) if x > 1000]
        if slow and "write_ring" not in k: print("      calls above 1 ms (index, us):", slow)
        print("%-40s n %4d  total %8.0f us  median %7.1f  p90 %7.1f  max %8.1f" % (k, v.size, v.sum(), np.median(v), np.percentile(v, 90), v.max()))
if diag:
    import numpy as np
    d = np.array(diag[2:])
    print("tracking calls: %d; per call median wall %.3f ms, kernel %.3f ms, channel-epochs %.0f, epochs %.0f; kernel us per epoch %.2f"
          % (len(d), np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 2]), np.median(d[:, 3]), 1e3 * d[:, 1].sum() / max(d[:, 3].sum(), 1)))
