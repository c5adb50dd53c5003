"""Differential sweep: many random scenes at BASELINE configs[1] size, every (PRN, Doppler bin) plane compared with
the oracle — 8 scenes x 32 PRNs x 41 bins = 10 496 planes of 8000 code phases.  Indices (argmax, detections) exact;
max / sum within 1e-5 relative.  Weak satellites near the detection threshold are included on purpose."""
from concurrent.futures import ThreadPoolExecutor

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REL = 1e-5


def test_random_scenes_all_planes(gpu, oracle):
    from gnss_sdr_rs_amd import acquisition as A, synth
    t = oracle.ca_code_table()
    fs, N, M = 8.0e6, 8000, 10
    dop = np.array([-5000.0 + 250.0 * i for i in range(41)], np.float32)
    tables = [oracle.DopplerShiftTable(0.0, float(d), fs, N) for d in dop]
    eng = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=dop, n_integrations=M)
    total_planes = mism = found_total = 0
    n_scenes = int(os.environ.get("GM_SWEEP_SCENES", "8"))      # one-off larger sweeps: GM_SWEEP_SCENES=64
    for seed in range(n_scenes):
        rng = np.random.default_rng(1000 + seed)
        prns = rng.choice(np.arange(1, 33), size=int(rng.integers(3, 10)), replace=False)
        sats = [dict(prn=int(p), prn_row=int(p) - 1, cn0_dbhz=float(rng.uniform(33.0, 50.0)),
                     doppler_hz=float(rng.uniform(-4900, 4900)), code_start=int(rng.integers(0, N)),
                     phase=float(rng.uniform(0, 6.28))) for p in prns]
        x = synth.make_scene(t, fs, 0.0, M * N, sats, config_id=100 + seed, bit_flip_at=(5 * N + 123) if seed % 3 == 0 else None)
        xi8 = synth.to_i8_iq(x)
        xc = synth.to_c32(x)
        got = eng.search(xi8, local_tail=seed * 7)
        mx, am, sm = eng.metrics()

        def ref(p):
            return oracle.AcquisitionWorker(p, N, fs).search_satellite(xc, tables, seed * 7, M, want_planes=True, no_early_exit=True)

        with ThreadPoolExecutor(16) as ex:
            exps = list(ex.map(ref, range(1, 33)))
        for w, (exp, (bmax, barg, bsum, _)) in enumerate(exps):
            assert np.allclose(mx[w], bmax, rtol=REL, atol=0) and np.allclose(sm[w], bsum, rtol=REL, atol=0)
            total_planes += dop.size
            mism += int((am[w] != barg).sum())
            assert (got[w] is None) == (exp is None), (seed, w + 1)
            if exp:
                found_total += 1
                for k in ("prn", "code_phase_samples", "sample_global_index", "doppler_bin", "carrier_freq", "code_phase_chips"):
                    assert got[w][k] == exp[k], (seed, k, got[w], exp)
                assert got[w]["mag_relative"] == pytest.approx(exp["mag_relative"], rel=REL)
    assert total_planes == n_scenes * 32 * 41
    assert mism == 0, f"{mism} of {total_planes} noise-plane argmax indices differ from the oracle"
    assert found_total >= 20 * n_scenes // 8
    eng.close()
