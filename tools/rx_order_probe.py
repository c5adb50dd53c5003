import json, os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path.insert(0, ROOT)
import torch, bench
from gnss_sdr_rs_amd import _lib, acquisition as A, tracking as T, synth
_lib.init(0)
synth.DEFAULT_GENERATOR = "xoshiro"
dev = torch.device("cuda:0")
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ca = A.ca_code_table()
which = sys.argv[1]
if which == "frontend":
    bench.frontend_leg(torch, dev, False)
elif which == "cfg1":
    bench.cfg1_leg(torch, dev, st.cuda_stream, ca, A, synth, False)
elif which == "cfg4":
    bench.cfg4_leg(torch, dev, A, synth)
elif which == "grid":
    bench.cfg4_grid_leg(torch, dev, st.cuda_stream, ca, A, synth, 1, 0, None, False)
keep = []
if which.startswith("eng"):
    import numpy as np
    sc = synth.cfg2_scene(ca)
    for i in range(int(which[3:] or 1)):
        e = A.AcquisitionEngine(sc["fs"], sc["f_if"], sc["N"], doppler_hz=sc["doppler_hz"], n_integrations=sc["M"])
        e.set_stream(st.cuda_stream)
        keep.append(e)
if which.startswith("streams"):
    keep = [torch.cuda.Stream() for _ in range(int(which[7:]))]
    for k in keep:
        with torch.cuda.stream(k): torch.zeros(16, device=dev)
    torch.cuda.synchronize()
r = bench.receiver_leg(ca, A, T, synth, False)
print(which, r["x_real_time"])
