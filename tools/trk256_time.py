"""Diagnostic: bench.py's tracking leg at 256 channels x 25 Msps (three arms): ms per epoch, channel x Msps."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gnss_sdr_rs_amd import _lib, acquisition as A, tracking as T, synth
_lib.init(0)
synth.DEFAULT_GENERATOR = "xoshiro"
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
for C in ((32,) if os.environ.get("TRK_ONLY32") else (256, 32)):
    t = bench.tracking_leg(torch, torch.device("cuda:0"), st.cuda_stream, A.ca_code_table(), T, synth, 1, None, 0.0, C=C)
    print(C, json.dumps({k: t[k] for k in ("value", "ms_per_epoch", "channels_locked")}))
