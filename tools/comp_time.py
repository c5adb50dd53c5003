"""Diagnostic: forward (sub + post) and correlation kernel times of the composite path at the configs[3] Galileo geometry
(36 codes x 41 bins x N = 32000 = 2 x 16000, 2 periods) from the library's own HIP events."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gnss_sdr_rs_amd import _lib, acquisition as A, synth
_lib.init(0)
N = int(os.environ.get("COMP_N", "32000"))
fs, L, rate, M, P = 8.0e6, 4092, 1.023e6, 2, 36
rng = np.random.default_rng(4)
codes = np.where(rng.integers(0, 2, (P, L)) > 0, 1, -1).astype(np.int8)
dop = np.arange(-5000.0, 5000.1, 250.0, dtype=np.float32)
x = synth.to_i8_iq(synth.make_scene(codes, fs, 0.0, M * N, [dict(prn_row=1, cn0_dbhz=48.0, doppler_hz=300.0, code_start=77)], config_id=44, code_rate=rate))
eng = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=dop, prn_ids=np.arange(1, P + 1), n_integrations=M, codes=codes, code_rate=rate)
d_x = torch.from_numpy(x).cuda()
d_met = torch.zeros(3 * P * dop.size, dtype=torch.int32, device="cuda")
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
eng.set_stream(st.cuda_stream)
for _ in range(3):
    eng.search_dev(d_x.data_ptr(), A.FMT_I8_IQ, d_met.data_ptr()); eng.decide_dev(d_met.data_ptr())
torch.cuda.synchronize()
eng.enable_timing(True)
for _ in range(20):
    eng.search_dev(d_x.data_ptr(), A.FMT_I8_IQ, d_met.data_ptr()); eng.decide_dev(d_met.data_ptr())
torch.cuda.synchronize()
print("N", N, eng.timing_summary())
