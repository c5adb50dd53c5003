import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gnss_sdr_rs_amd import _lib, acquisition as A, synth
_lib.init(0)
fs, L, rate, M, P, N = 8.0e6, 4092, 1.023e6, 2, 36, 32000
rng = np.random.default_rng(4)
codes = np.where(rng.integers(0, 2, (P, L)) > 0, 1, -1).astype(np.int8)
dop = np.arange(-5000.0, 5000.1, 250.0, dtype=np.float32)
x = synth.to_i8_iq(synth.make_scene(codes, fs, 0.0, M * N, [dict(prn_row=1, cn0_dbhz=48.0, doppler_hz=300.0, code_start=77)], config_id=44, code_rate=rate))
eng = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=dop, prn_ids=np.arange(1, P + 1), n_integrations=M, codes=codes, code_rate=rate)
d_x = torch.from_numpy(x).cuda()
d_met = torch.zeros(3 * P * dop.size, dtype=torch.int32, device="cuda")
eng.search_dev(d_x.data_ptr(), A.FMT_I8_IQ, d_met.data_ptr()); torch.cuda.synchronize()
ref = d_met.clone()
t0 = time.time(); bad = 0
for i in range(200):
    for _ in range(100):
        eng.search_dev(d_x.data_ptr(), A.FMT_I8_IQ, d_met.data_ptr())
    torch.cuda.synchronize()
    if not torch.equal(d_met, ref): bad += 1
    if i % 20 == 0: print("block", i, "elapsed %.1f s" % (time.time() - t0), "mismatching blocks", bad, flush=True)
print("20000 launches of the composite dwell, %.1f s, mismatching blocks: %d" % (time.time() - t0, bad))
# the same for the reference geometry (ws31)
import json
cap = json.load(open("/root/repo/tests/golden/capture_config.json"))
sc = synth.cfg1_scene(A.ca_code_table(), cap)
x1 = torch.from_numpy(synth.to_i8_real(sc["x"])).cuda()
e1 = A.AcquisitionEngine(sc["fs"], sc["f_if"], sc["N"], doppler_hz=sc["doppler_hz"], n_integrations=sc["M"])
m1 = torch.zeros(3 * 32 * len(sc["doppler_hz"]), dtype=torch.int32, device="cuda")
e1.search_dev(x1.data_ptr(), A.FMT_I8_REAL, m1.data_ptr()); torch.cuda.synchronize()
r1 = m1.clone(); bad = 0; t0 = time.time()
for i in range(100):
    for _ in range(100):
        e1.search_dev(x1.data_ptr(), A.FMT_I8_REAL, m1.data_ptr())
    torch.cuda.synchronize()
    if not torch.equal(m1, r1): bad += 1
print("10000 launches of the N = 16368 dwell, %.1f s, mismatching blocks: %d" % (time.time() - t0, bad))
