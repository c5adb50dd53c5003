"""Diagnostic: acq_corr_kernel time vs number of workers (grid-tail / quantisation check), configs[1] geometry."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnss_sdr_rs_amd import _lib, acquisition as A, synth
_lib.init(0)
ca = A.ca_code_table(); sc = synth.cfg2_scene(ca)
x = torch.from_numpy(synth.to_i8_iq(sc["x"])).cuda()
for P in (12, 24, 25, 28, 32):
    eng = A.AcquisitionEngine(sc["fs"], sc["f_if"], sc["N"], doppler_hz=sc["doppler_hz"], prn_ids=np.arange(1, P + 1), n_integrations=sc["M"])
    for _ in range(3):
        eng.search_dev(x.data_ptr(), A.FMT_I8_IQ)
    eng.synchronize(); eng.enable_timing(True)
    for _ in range(20):
        eng.search_dev(x.data_ptr(), A.FMT_I8_IQ)
    ts = eng.timing_summary()
    wgs = P * 41
    print(f"P={P:2d} workgroups={wgs:4d} rounds={wgs/512:.2f} corr={ts['avg_corr_ms']*1e3:7.1f} us  per-WG-round={ts['avg_corr_ms']*1e3/ (wgs/512):6.1f} us  us per 512 WGs if linear={ts['avg_corr_ms']*1e3*512/wgs:6.1f}")
    eng.close()
