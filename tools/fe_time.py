import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from gnss_sdr_rs_amd import frontend as F, _lib
_lib.init(0)
dev = torch.device('cuda', 0)
n = 1 << 21
xi = np.random.default_rng(2).integers(-127, 128, 2 * n).astype(np.int8)
d_in = torch.from_numpy(xi).to(dev); d_out = torch.empty(2 * n, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for dbg in (0,):
    fe = F.DigitalFrontend(4.1304e6, 16.3676e6, 16.3676e6)
    fe.process_dev(d_in.data_ptr(), _lib.FMT_I8_IQ, d_out.data_ptr(), n, st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    fe.process_dev(d_in.data_ptr(), _lib.FMT_I8_IQ, d_out.data_ptr(), n, st); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('dbg', dbg, 'us/segment %.2f' % (dt / (n / 2048) * 1e6), 'Msps %.1f' % (n / dt / 1e6), flush=True)
    fe.close()
