import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from gnss_sdr_rs_amd import _lib, acquisition as A, synth
_lib.init(0)
dev = torch.device('cuda', 0)
ca = A.ca_code_table(); sc = synth.cfg2_scene(ca)
P, D, N, M = 32, int(sc["doppler_hz"].size), sc["N"], sc["M"]
xi8 = synth.to_i8_iq(sc["x"])
d_samples = torch.from_numpy(xi8).to(dev)
for n_eng in (1, 2, 3):
    engs, mets, streams = [], [], []
    for k in range(n_eng):
        e = A.AcquisitionEngine(sc["fs"], sc["f_if"], N, doppler_hz=sc["doppler_hz"], n_integrations=M)
        st = torch.cuda.Stream(device=dev)
        e.set_stream(st.cuda_stream)
        engs.append(e); streams.append(st); mets.append(torch.zeros(3 * P * D, dtype=torch.int32, device=dev))
    def step(i):
        e = engs[i % n_eng]
        e.search_dev(d_samples.data_ptr(), A.FMT_I8_IQ, mets[i % n_eng].data_ptr())
        e.decide_dev(mets[i % n_eng].data_ptr())
    for i in range(6): step(i)
    torch.cuda.synchronize()
    for e in engs: e.enable_timing(True)
    t0 = time.perf_counter()
    K = 60
    for i in range(K): step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ts = [e.timing_summary() for e in engs]
    print(n_eng, 'engines: ms/step %.4f' % (dt / K * 1e3), 'cells/s %.3e' % (P * D * N * K / dt), 'corr ms', [round(t["avg_corr_ms"], 4) for t in ts], flush=True)
    res = engs[0].fetch_results(P)
    print('  found', sorted(r["prn"] for r in res if r))
    for e in engs: e.close()
