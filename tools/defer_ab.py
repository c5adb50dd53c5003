"""A/B of gm_acq_set_deferred_decision on the headline geometry: K back-to-back dwells, decision immediate / deferred, interleaved."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import importlib
A = importlib.import_module("gnss_sdr_rs_amd.acquisition")
fs, N, M = 8.0e6, 8000, 10
dop = np.arange(-10000.0, 10001.0, 500.0, dtype=np.float32)
eng = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=dop, prn_ids=list(range(1, 33)), n_integrations=M)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
x = torch.randint(-100, 100, (M * N * 2,), dtype=torch.int8, device="cuda")
met = torch.empty(3 * 32 * dop.size, dtype=torch.int32, device="cuda")
def run(k):
    for _ in range(k):
        eng.search_dev(x.data_ptr(), A.FMT_I8_IQ, met.data_ptr()); eng.decide_dev(met.data_ptr())
    eng.synchronize()
for rep in range(4):
    for on in (False, True):
        eng.set_deferred_decision(on)
        run(20); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(300); torch.cuda.synchronize(); t = time.perf_counter() - t0
        print(f"deferred={on}: {t / 300 * 1e6:.1f} us per dwell", flush=True)
