import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gnss_sdr_rs_amd import _lib, acquisition as A, synth, distributed as Dm
_lib.init(0)
dev = torch.device("cuda", 0)
ca = A.ca_code_table()
stream = torch.cuda.current_stream().cuda_stream
for rep in range(3):
    out = bench.cfg4_grid_leg(torch, dev, stream, ca, A, synth, 1, 0, None, False)
    print(rep, {k: v for k, v in out.items() if k in ("ms_per_dwell", "cells_per_s", "simulated_satellites_found_at_true_phase")})
