"""Diagnostic: the composite (N = 32000) path must give a code the same metrics words whatever other codes share the engine,
and the same words launch after launch."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnss_sdr_rs_amd import _lib, acquisition as A, distributed as Dm, synth

_lib.init(0)
ca = A.ca_code_table()
b1i = A.b1i_codes(range(1, 23))
sc = synth.cfg4_grid_scene(ca, b1i)
f = Dm.baseline_grid_families(sc, b1i)[1]
x = synth.to_i8_iq(sc["x"])


def run(first, cnt, reps=3):
    eng = A.AcquisitionEngine(f.fs, f.f_if, f.fft_size, doppler_hz=f.doppler_hz, prn_ids=f.prn_ids[first:first + cnt],
                              n_integrations=f.M, codes=f.codes[first:first + cnt], code_rate=f.code_rate)
    outs = []
    for _ in range(reps):
        eng.search(x)
        outs.append(np.stack([a.view(np.uint32) for a in eng.metrics()]))
    cf = np.stack([eng.code_fft(w) for w in range(cnt)])
    eng.close()
    return outs, cf


full, cf_full = run(0, 36)
for i in (1, 2):
    print("full run", i, "== run 0:", bool((full[i] == full[0]).all()))
for first, cnt in ((0, 3), (3, 11), (14, 11), (25, 11)):
    part, cf_part = run(first, cnt)
    print("codes", first, cnt, "code spectra equal:", bool((cf_part.view(np.uint32) == cf_full[first:first + cnt].view(np.uint32)).all()))
    for i in range(3):
        d = np.argwhere(part[i] != full[0][:, first:first + cnt, :])
        print("  run", i, "differs in", len(d), "words; per plane", [int((d[:, 0] == q).sum()) for q in range(3)])
        for q, p, b in d[:4]:
            a_, b_ = part[i][q, p, b], full[0][q, first + p, b]
            print("     plane", q, "worker", first + p, "bin", b, a_.view(np.float32) if q != 1 else a_, b_.view(np.float32) if q != 1 else b_)
