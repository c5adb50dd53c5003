"""A/B of gm_acq_prepare_dev (stage F of dwell k + 1 beside stage C of dwell k): K back-to-back dwells, plain / prepared, interleaved,
on the reference's geometry (32 PRN x 29 bins x 16368, real int8) and on the headline geometry (32 x 41 x 8000, int8 IQ)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, importlib
A = importlib.import_module("gnss_sdr_rs_amd.acquisition")
def leg(name, fs, N, M, dop, fmt, n_per, prns=32):
    eng = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=dop, prn_ids=list(range(1, prns + 1)), n_integrations=M)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    xs = [torch.randint(-100, 100, (M * N * n_per,), dtype=torch.int8, device="cuda") for _ in range(2)]
    met = torch.empty(3 * prns * dop.size, dtype=torch.int32, device="cuda")
    def run(k, ahead):
        tok = eng.prepare_dev(xs[0].data_ptr(), fmt) if ahead else 0
        for i in range(k):
            if ahead:
                eng.search_prepared_dev(tok, met.data_ptr())
            else:
                eng.search_dev(xs[i & 1].data_ptr(), fmt, met.data_ptr())
            if ahead and i + 1 < k:
                tok = eng.prepare_dev(xs[(i + 1) & 1].data_ptr(), fmt)
            eng.decide_dev(met.data_ptr())
        eng.synchronize()
    for rep in range(3):
        for ahead in (False, True):
            run(10, ahead); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(200, ahead); torch.cuda.synchronize(); t = time.perf_counter() - t0
            print(f"{name}: prepared={ahead}: {t / 200 * 1e6:.1f} us per dwell", flush=True)
    eng.close()
leg("N=16368 32x29", 16.368e6, 16368, 10, np.arange(-7000.0, 7001.0, 500.0, dtype=np.float32), A.FMT_I8_REAL, 1)
leg("N=8000 32x41", 8.0e6, 8000, 10, np.arange(-5000.0, 5001.0, 250.0, dtype=np.float32), A.FMT_I8_IQ, 2)
leg("N=16368 1x29 (configs[0] proper)", 16.368e6, 16368, 10, np.arange(-7000.0, 7001.0, 500.0, dtype=np.float32), A.FMT_I8_REAL, 1, prns=1)
