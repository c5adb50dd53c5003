"""Feasibility probe: does a SMALL-footprint stage-F-like load (the N = 1024 stage-F kernel: 128 lanes, 52 registers, 9 KB of LDS) run
beside the headline's stage C (two 512-lane workgroups per CU, 110 registers, 2 x 66 KB) in the issue slots stage C leaves idle, or does
it displace / slow it like the N = 8000 stage-F workgroups do?  Engine A: the headline dwell loop.  Engine B: prepare_dev() only (stage F on
B's own second stream), `BG` transforms of 1024 points per dwell of A."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, importlib
A = importlib.import_module("gnss_sdr_rs_amd.acquisition")
N, M = 8000, 10
dop = np.arange(-5000.0, 5001.0, 250.0, dtype=np.float32)
eng = A.AcquisitionEngine(8.0e6, 0.0, N, doppler_hz=dop, prn_ids=list(range(1, 33)), n_integrations=M)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
x = torch.randint(-100, 100, (M * N * 2,), dtype=torch.int8, device="cuda")
met = torch.empty(3 * 32 * dop.size, dtype=torch.int32, device="cuda")
NB = int(os.environ.get("BG_N", "1024"))
def bg_engine(n_items):
    d = np.linspace(-5000.0, 5000.0, n_items // 10).astype(np.float32)
    e = A.AcquisitionEngine(NB * 1000.0, 0.0, NB, doppler_hz=d, prn_ids=[1], n_integrations=10)
    xb = torch.randint(-100, 100, (10 * NB * 2,), dtype=torch.int8, device="cuda")
    return e, xb
for items in (0, 410, 1640, 3280):
    b = bg_engine(items) if items else None
    def run(k):
        for _ in range(k):
            eng.search_dev(x.data_ptr(), A.FMT_I8_IQ, met.data_ptr())
            if b: b[0].prepare_dev(b[1].data_ptr(), A.FMT_I8_IQ)
            eng.decide_dev(met.data_ptr())
        eng.synchronize(); torch.cuda.synchronize()
    for rep in range(2):
        run(20)
        t0 = time.perf_counter(); run(200); t = time.perf_counter() - t0
        print(f"background {items} x {NB}-point stage-F transforms per dwell: {t / 200 * 1e6:.1f} us per dwell", flush=True)
    if b: b[0].close()
