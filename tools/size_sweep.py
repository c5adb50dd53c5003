"""One-off differential sweep over the transform sizes whose plans changed in round 6 (tools/corr_lab/plan_search.py): per size, SWEEP_SCENES
random scenes x 12 PRNs x 9 Doppler bins x 4 integrations, every plane against the oracle — arg-max indices and decisions exact,
maxima and sums within 1e-5.  GPU box; the oracle runs on the host's cores beside it."""
import os, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O
from gnss_sdr_rs_amd import _lib, acquisition as A, synth
_lib.init(0)
synth.DEFAULT_GENERATOR = "xoshiro"
t = O.ca_code_table()
M, REL = 4, 1e-5
dop = np.array([-1000.0 + 250.0 * i for i in range(9)], np.float32)
sizes = [int(v) for v in os.environ.get("SWEEP_SIZES", "16384,15000,12000,10000,8192,8184,6000,5000,24576,18000,24552").split(",")]
n_scenes = int(os.environ.get("SWEEP_SCENES", "3"))
for N in sizes:
    fs = N * 1000.0
    prns = list(range(1, 13))
    eng = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=dop, prn_ids=prns, n_integrations=M)
    tables = [O.DopplerShiftTable(0.0, float(d), fs, N) for d in dop]
    planes = mism = found = 0
    for seed in range(n_scenes):
        rng = np.random.default_rng(N + seed)
        sats = [dict(prn=int(p), prn_row=int(p) - 1, cn0_dbhz=float(rng.uniform(36.0, 50.0)), doppler_hz=float(rng.uniform(-950, 950)),
                     code_start=int(rng.integers(0, N)), phase=float(rng.uniform(0, 6.28))) for p in rng.choice(prns, size=5, replace=False)]
        x = synth.make_scene(t, fs, 0.0, M * N, sats, config_id=300 + seed)
        xi8, xc = synth.to_i8_iq(x), synth.to_c32(x)
        got = eng.search(xi8, local_tail=seed * 3)
        mx, am, sm = eng.metrics()
        with ThreadPoolExecutor(12) as ex:
            exps = list(ex.map(lambda p: O.AcquisitionWorker(p, N, fs).search_satellite(xc, tables, seed * 3, M, want_planes=True, no_early_exit=True), prns))
        for w, (exp, (bmax, barg, bsum, _)) in enumerate(exps):
            assert np.allclose(mx[w], bmax, rtol=REL, atol=0) and np.allclose(sm[w], bsum, rtol=REL, atol=0), (N, seed, w)
            planes += dop.size
            mism += int((am[w] != barg).sum())
            assert (got[w] is None) == (exp is None), (N, seed, w)
            if exp:
                found += 1
                for k in ("prn", "code_phase_samples", "sample_global_index", "doppler_bin", "carrier_freq"):
                    assert got[w][k] == exp[k], (N, seed, k, got[w], exp)
    eng.close()
    print("N = %6d: %4d planes, %d arg-max mismatches, %d detections equal to the oracle's" % (N, planes, mism, found), flush=True)
    assert mism == 0
print("sweep ok")
