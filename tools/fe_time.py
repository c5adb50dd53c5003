"""Diagnostic: the front-end kernel on one block of FE_N (2^19) int8 IQ samples resident in HBM -> c32: Msps and us per block
(GM_DIAGNOSTICS=1 GM_FE_SPEC=0: one workgroup per block; =2: every guess of the speculative form spoiled, i.e. every run repaired)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnss_sdr_rs_amd import frontend as F, _lib
_lib.init(0)
dev = torch.device('cuda', 0)
n = int(os.environ.get("FE_N", str(1 << 19)))
rng = np.random.default_rng(2)
xi = np.clip(np.rint(rng.normal(5.0, 16.0, 2 * n)), -127, 127).astype(np.int8)
d_in = torch.from_numpy(xi).to(dev); d_out = torch.empty(2 * n, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
fe = F.DigitalFrontend(4.1304e6, 16.3676e6, 16.3676e6)
for _ in range(3):
    fe.process_dev(d_in.data_ptr(), _lib.FMT_I8_IQ, d_out.data_ptr(), n, st)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    fe.process_dev(d_in.data_ptr(), _lib.FMT_I8_IQ, d_out.data_ptr(), n, st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print('GM_FE_SPEC=%s: %.1f us per block of %d samples, %.1f Msps; runs repaired in %d blocks: %d' % (os.environ.get("GM_FE_SPEC", "default"), dt * 1e6, n, n / dt / 1e6, K + 3, fe.debug_repairs()), flush=True)
fe.close()
