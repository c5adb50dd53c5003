"""Diagnostic: stage F / correlation kernel times at the reference's test geometry (32 PRN x 29 bins x N = 16368, 10 ms, real int8)
from the library's HIP events; GM_CORR_SPLIT=10 forces the grid-tail cut that one-workgroup-per-CU plans skip by default."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gnss_sdr_rs_amd import _lib, acquisition as A, synth
_lib.init(0)
cap = json.load(open(os.path.join(ROOT, "tests", "golden", "capture_config.json")))
sc = synth.cfg1_scene(A.ca_code_table(), cap)
x = torch.from_numpy(synth.to_i8_real(sc["x"])).cuda()
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
eng = A.AcquisitionEngine(sc["fs"], sc["f_if"], sc["N"], doppler_hz=sc["doppler_hz"], n_integrations=sc["M"])
eng.set_stream(st.cuda_stream)
for _ in range(3):
    eng.search_dev(x.data_ptr(), A.FMT_I8_REAL); eng.decide_dev()
torch.cuda.synchronize()
eng.enable_timing(True)
for _ in range(30):
    eng.search_dev(x.data_ptr(), A.FMT_I8_REAL); eng.decide_dev()
torch.cuda.synchronize()
print("split env", os.environ.get("GM_CORR_SPLIT"), eng.timing_summary())
res = eng.fetch_results()
print("found", [(r["prn"], r["code_phase_samples"], r["doppler_bin"], round(r["mag_relative"], 1)) for r in res if r])
