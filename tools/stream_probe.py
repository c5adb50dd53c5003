"""Diagnostic: is a library launch on HIP stream 0 ordered with torch ops on torch's default stream?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gnss_sdr_rs_amd import _lib, acquisition as A, synth
_lib.init(0)
print("torch default stream handle:", torch.cuda.current_stream().cuda_stream, "hip", torch.version.hip)
sc = synth.cfg2_scene(A.ca_code_table())
x = torch.from_numpy(synth.to_i8_iq(sc["x"])).cuda()
P, D = 32, 41


def trial(stream_obj, label):
    eng = A.AcquisitionEngine(sc["fs"], sc["f_if"], sc["N"], doppler_hz=sc["doppler_hz"], n_integrations=sc["M"])
    h = stream_obj.cuda_stream
    eng.set_stream(h)
    met = torch.zeros(3 * P * D, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    bad = 0
    for i in range(10):
        met.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(stream_obj):
            eng.search_dev(x.data_ptr(), A.FMT_I8_IQ, met.data_ptr())
            snap = met.clone()           # must be ordered behind the search
        torch.cuda.synchronize()
        bad += int((snap != met).sum().item())
    print(label, "handle", h, "stale words over 10 dwells:", bad)
    eng.close()


trial(torch.cuda.current_stream(), "torch default stream")
trial(torch.cuda.Stream(), "torch side stream")
trial(torch.cuda.ExternalStream(torch.cuda.Stream().cuda_stream), "external stream")
