"""Diagnostic: bench.py's `cfg5_geometry` leg alone (BASELINE configs[4]: 36 channels x 50 Msps, 4092-chip BOC(1,1), five arms,
4 ms code periods) — ms per code period, channel x Msps; CFG5_REPS repeats the leg."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gnss_sdr_rs_amd import _lib, tracking as T
_lib.init(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
for _ in range(int(os.environ.get("CFG5_REPS", "2"))):
    out = bench.cfg5_leg(torch, st.cuda_stream, T)
    print(json.dumps({k: out.get(k) for k in ("ms_per_code_period", "ch_msps", "channels_locked", "error", "roofline")}))
