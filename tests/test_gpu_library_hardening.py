"""Round-4 hardening of the C-ABI library:

  * gm_acq_cfg.reference_products (ABI 5): `result_buf[i] *= conj(code_fft[i])` and `norm_sqr()` rounded exactly as
    num-complex rounds them (do_acquisition.rs:184-192) — selectable at run time, parity with the oracle on both sides;
  * the library reads no GM_* environment override unless GM_DIAGNOSTICS=1 is set;
  * one strict_sum_order tracking manager and one default manager in flight on one device (the per-device launch chain
    now covers the strict path as well);
  * the split scratch is sized from the handle's geometry: the reference-style set-up of 32 one-PRN workers fits in
    well under a gigabyte and every cut item still merges to the words of the uncut grid.
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scene(oracle, fs, N, M, config_id, prns=(3, 9), cn0=46.0):
    from gnss_sdr_rs_amd import synth
    t = oracle.ca_code_table()
    sats = [dict(prn_row=p - 1, cn0_dbhz=cn0 + i, doppler_hz=-700.0 + 900.0 * i, code_start=(977 * (i + 1)) % N) for i, p in enumerate(prns)]
    return synth.to_c32(synth.make_scene(t, fs, 0.0, M * N, sats, config_id=config_id)), sats


@pytest.mark.parametrize("fs,N", [(2.048e6, 2048), (8.0e6, 8000), (16.368e6, 16368)])
def test_reference_products_option_matches_the_oracle(gpu, oracle, fs, N):
    from gnss_sdr_rs_amd import acquisition as A
    M = 4
    dop = np.arange(-1000.0, 1000.1, 500.0, dtype=np.float32)
    prn_ids = [3, 9, 17]
    x, sats = _scene(oracle, fs, N, M, config_id=701)
    tables = [oracle.DopplerShiftTable(0.0, float(d), fs, N) for d in dop]
    planes = {}
    for ref in (False, True):
        eng = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=dop, prn_ids=prn_ids, n_integrations=M, reference_products=ref)
        got = eng.search(x)
        mx, am, sm = eng.metrics()
        eng.close()
        planes[ref] = (mx.copy(), am.copy(), sm.copy())
        for i, p in enumerate(prn_ids):
            w = oracle.AcquisitionWorker(p, N, fs)
            exp, (bmax, barg, bsum, _) = w.search_satellite(x, tables, 0, M, want_planes=True, no_early_exit=True)
            exp_early = w.search_satellite(x, tables, 0, M)
            assert (got[i] is None) == (exp_early is None), (ref, p)
            if exp_early is not None:
                for k in ("prn", "doppler_bin", "code_phase_samples", "carrier_freq", "sample_global_index"):
                    assert got[i][k] == exp_early[k], (ref, p, k)
                assert abs(got[i]["mag_relative"] - exp_early["mag_relative"]) <= 1e-5 * exp_early["mag_relative"]   # north_star: 1e-5 relative
            assert (am[i] == barg).all(), (ref, p)                                   # bit-exact indices on every plane
            assert float(np.max(np.abs(mx[i] - bmax) / bmax)) <= 1e-5
            assert float(np.max(np.abs(sm[i] - bsum) / bsum)) <= 1e-5
    # the option changes roundings, nothing else: same argmax everywhere, values within a few ulps of each other
    assert (planes[False][1] == planes[True][1]).all()
    assert float(np.max(np.abs(planes[False][0] - planes[True][0]) / planes[True][0])) <= 2e-6


def test_reference_products_grid_tail_cut_merges_to_the_same_words(gpu, oracle):
    """The parts of a cut item (one integration each) must merge to the very words of the uncut item with the num-complex
    roundings as well: a one-worker handle cuts EVERY item, a 32-worker handle of the same shape cuts only its tail."""
    from gnss_sdr_rs_amd import acquisition as A
    fs, N, M = 8.0e6, 8000, 10
    dop = np.arange(-2000.0, 2000.1, 250.0, dtype=np.float32)
    x, _ = _scene(oracle, fs, N, M, config_id=702, prns=(5,))
    one = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=dop, prn_ids=[5], n_integrations=M, reference_products=True)
    one.search(x)
    m1 = [a.copy() for a in one.metrics()]
    one.close()
    allp = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=dop, n_integrations=M, reference_products=True)
    allp.search(x)
    m32 = allp.metrics()
    allp.close()
    for a, b in zip(m1, m32):
        assert (a[0].view(np.uint32) == b[4].view(np.uint32)).all()


def test_reference_products_rejected_on_the_composite_path(gpu):
    from gnss_sdr_rs_amd import acquisition as A
    with pytest.raises(Exception):
        A.AcquisitionEngine(8.0e6, 0.0, 32000, doppler_hz=np.array([0.0], np.float32), prn_ids=[1], n_integrations=2,
                            reference_products=True)


_PROBE = r"""
import hashlib, json, sys
sys.path.insert(0, %r)
import numpy as np
from oracle import oracle as O
from gnss_sdr_rs_amd import _lib, acquisition as A, synth
_lib.init(0)
t = O.ca_code_table()
fs, N, M = 16.368e6, 16368, 2
x = synth.to_c32(synth.make_scene(t, fs, 0.0, M * N, [dict(prn_row=6, cn0_dbhz=50.0, doppler_hz=120.0, code_start=4242)], config_id=703))
eng = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=np.array([-500.0, 0.0, 500.0], np.float32), prn_ids=[7, 8], n_integrations=M)
got = eng.search(x)
mx, am, sm = eng.metrics()
print(json.dumps({"sha": hashlib.sha256(mx.tobytes() + am.tobytes() + sm.tobytes()).hexdigest(),
                  "found": [None if g is None else [g["prn"], g["code_phase_samples"], g["doppler_bin"]] for g in got]}))
eng.close()
"""


def _probe(extra_env):
    env = {k: v for k, v in os.environ.items() if not k.startswith("GM_")}
    env.update(extra_env)
    r = subprocess.run([sys.executable, "-c", _PROBE % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_environment_overrides_need_the_diagnostics_switch(gpu):
    """GM_COMP_BASE=8184 sends N = 16368 through the composite path (2 x 8184: other roundings, same detections) — but only
    when GM_DIAGNOSTICS=1 says the process wants diagnostics.  Without it the library must not look at the variable."""
    plain = _probe({})
    ignored = _probe({"GM_COMP_BASE": "8184", "GM_CORR_SPLIT": "0", "GM_CORR_MAP": "1"})
    honoured = _probe({"GM_DIAGNOSTICS": "1", "GM_COMP_BASE": "8184"})
    assert plain["found"] == ignored["found"] == honoured["found"] and plain["found"][0][:2] == [7, 4242] and plain["found"][1] is None
    assert ignored["sha"] == plain["sha"]           # every metrics word equal: the overrides were not read
    assert honoured["sha"] != plain["sha"]          # the switch works (another transform decomposition, other low bits)


def test_strict_and_default_tracking_managers_share_a_device(gpu, oracle):
    """a strict_sum_order manager's launches used to bypass the per-device chain of persistent launches."""
    from gnss_sdr_rs_amd import tracking as T, synth
    fs, n, E, C = 8.0e6, 8000, 12, 32
    t = oracle.ca_code_table()
    prns = [2, 5, 9, 13, 17, 22, 26, 30]
    sc = synth.tracking_scene(t, fs, 0.0, prns, E + 2, config_id=63, cn0=50.0)
    ring = T.MulticastRingBuffer(1 << 18)
    ring.write_samples(synth.to_c32(sc["x"])[:(E + 1) * n])

    def fresh(strict):
        m = T.TrackingManager(fs, n_channels=C, code_index_mode=T.CODE_INDEX_FIXED, strict_sum_order=strict)
        for i in range(C):
            s = sc["sats"][i % 8]
            m.channels[i].start(dict(prn=s["prn"], code_phase_samples=0, code_phase_chips=0.0, carrier_freq=s["doppler_hz"] + 10.0 - 0.3 * (i // 8),
                                     fs=fs, mag_relative=10.0, sample_global_index=s["code_start"], doppler_bin=0))
        return m
    want = []
    for strict in (False, True):
        m = fresh(strict)
        m.update_all_dev(ring, E); m.synchronize()
        want.append([m.channels[i].state for i in range(C)])
        m.close()
    for order in ((False, True), (True, False)):
        ms = [fresh(s) for s in order]
        for _ in range(2):          # E epochs in two calls each, interleaved
            for m in ms:
                m.update_all_dev(ring, E // 2)
        for m in ms:
            m.synchronize()          # raises on an exchange time-out
        for m, s in zip(ms, order):
            w = want[1 if s else 0]
            for i in range(C):
                st = m.channels[i].state
                assert st.next_sample_index == w[i].next_sample_index and st.lost_counter == 0
                for k in ("carrier_freq", "carrier_phase", "code_phase", "code_rate", "i_prompt", "q_prompt"):
                    assert getattr(st, k) == getattr(w[i], k), (order, s, i, k)
            m.close()
    ring.close()


def test_thirty_two_one_prn_workers_at_the_reference_geometry(gpu, oracle):
    """The reference's own set-up (do_acquisition.rs:268-271): 32 AcquisitionWorkers, one PRN each, N = 16368.  Each handle's
    split scratch is sized from its geometry (290 planes, not 2560): the 32 handles take ~0.6 GB of scratch instead of 8 GB,
    and the searches (every item cut into its 10 integrations) still find what one 32-PRN handle finds."""
    import ctypes as C
    from gnss_sdr_rs_amd import acquisition as A, synth
    cap = json.load(open(os.path.join(ROOT, "tests", "golden", "capture_config.json")))
    sc = synth.cfg1_scene(oracle.ca_code_table(), cap)
    x = synth.to_i8_real(sc["x"])
    hip = C.CDLL("libamdhip64.so.7")          # the runtime the product library already loaded (no PyTorch in this test)

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value
    free0 = free_bytes()
    workers = [A.AcquisitionEngine(sc["fs"], sc["f_if"], sc["N"], doppler_hz=sc["doppler_hz"], n_integrations=sc["M"], prn_ids=[p])
               for p in range(1, 33)]
    used = free0 - free_bytes()
    assert used < 4.0e9, used        # spectra 38 MB + tables 3.8 MB + scratch 19 MB + ... per handle
    found = {}
    for p, w in zip(range(1, 33), workers):
        r = w.search(x)[0]          # int8, one value per sample: GM_FMT_I8_REAL
        if r:
            found[p] = (r["code_phase_samples"], r["doppler_bin"])
        w.close()
    eng = A.AcquisitionEngine(sc["fs"], sc["f_if"], sc["N"], doppler_hz=sc["doppler_hz"], n_integrations=sc["M"])
    res = eng.search(x)
    eng.close()
    assert found == {r["prn"]: (r["code_phase_samples"], r["doppler_bin"]) for r in res if r}
    assert len(found) >= 8



def test_ws31_kernel_strict_sum_and_cut_items_at_the_reference_geometry(gpu, oracle):
    """N = 16368 runs the wave-specialised stage C (csrc/acq_corr_ws31.h: radix-31 pass on the matrix pipe).  Its
    strict_sum_order path (the plane through LDS, the reference's eight-lane sum, do_acquisition.rs:229-235) against the oracle's
    ordered sums, and its cut items (a one-PRN handle cuts every item into its integrations and merges the planes through HBM)
    against the uncut items of a 32-PRN handle: the same words."""
    from gnss_sdr_rs_amd import acquisition as A
    fs, N, M = 16.368e6, 16368, 4
    dop = np.arange(-1000.0, 1000.1, 500.0, dtype=np.float32)
    x, _ = _scene(oracle, fs, N, M, config_id=720, prns=(7, 21), cn0=47.0)
    tables = [oracle.DopplerShiftTable(0.0, float(d), fs, N) for d in dop]
    for strict in (False, True):
        one = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=dop, prn_ids=[7], n_integrations=M, strict_sum_order=strict)
        r1 = one.search(x)
        m1 = [a.copy() for a in one.metrics()]
        one.close()
        allp = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=dop, n_integrations=M, strict_sum_order=strict)
        r32 = allp.search(x)
        m32 = allp.metrics()
        allp.close()
        assert r1[0] == r32[6] and r1[0] is not None and r32[20] is not None
        if not strict:      # (strict_sum_order keeps the integrations in sequence: no cut, nothing to compare)
            for a, b in zip(m1, m32):
                assert (a[0].view(np.uint32) == b[6].view(np.uint32)).all()
        w = oracle.AcquisitionWorker(7, N, fs)
        exp, (bmax, barg, bsum, _) = w.search_satellite(x, tables, 0, M, want_planes=True, no_early_exit=True)
        mx, am, sm = m32
        assert (am[6] == barg).all()
        assert float(np.max(np.abs(mx[6] - bmax) / bmax)) <= 1e-5
        assert float(np.max(np.abs(sm[6] - bsum) / bsum)) <= (5e-7 if strict else 1e-5)


def test_ws31_edge_shapes_and_the_generic_fallback_read_the_same_layout(gpu, oracle):
    """N = 16368 edge shapes through the wave-specialised kernel — ONE integration, one worker x one bin, a masked worker list —
    against the oracle; and the generic kernel the size falls back to under gm_acq_debug_stamps must read the same stored order
    (row pairs in the radix-33 butterfly's consumption order, PairRows): identical metrics words with the stamps armed."""
    import ctypes as C
    from gnss_sdr_rs_amd import _lib, acquisition as A
    fs, N = 16.368e6, 16368
    dop = np.array([-500.0, 0.0, 500.0], np.float32)
    for M, prns, config in ((1, [7], 730), (3, [7, 8, 21], 731)):
        x, _ = _scene(oracle, fs, N, M, config_id=config, prns=(7, 21), cn0=50.0)
        tables = [oracle.DopplerShiftTable(0.0, float(d), fs, N) for d in dop]
        eng = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=dop, prn_ids=prns, n_integrations=M)
        got = eng.search(x)
        mx, am, sm = [a.copy() for a in eng.metrics()]
        for i, p in enumerate(prns):
            w = oracle.AcquisitionWorker(p, N, fs)
            exp, (bmax, barg, bsum, _) = w.search_satellite(x, tables, 0, M, want_planes=True, no_early_exit=True)
            assert (am[i] == barg).all(), (M, p)
            assert float(np.max(np.abs(mx[i] - bmax) / bmax)) <= 1e-5 and float(np.max(np.abs(sm[i] - bsum) / bsum)) <= 1e-5
            e2 = w.search_satellite(x, tables, 0, M)
            assert (got[i] is None) == (e2 is None) and (e2 is None or got[i]["code_phase_samples"] == e2["code_phase_samples"])
        if len(prns) == 3:
            masked = eng.search(x, prn_mask=0b101)                  # workers 0 and 2 only
            assert masked[1] is None and masked[0] == got[0] and masked[2] == got[2]
            # the generic kernel (diagnostic stamps variant) on the same handle: same stored spectra, same words — in a DIAGNOSTIC
            # build (-DGM_DIAG_STAMPS); the product library does not carry the stamped kernels and says so
            L = _lib.lib()
            rc = L.gm_acq_debug_stamps(eng._h, None)
            if rc == -8:                                           # GM_ERR_UNSUPPORTED: nothing was armed, nothing launched
                assert eng.search(x) == got
                eng.close()
                continue
            _lib.check(rc, "arm")
            again = eng.search(x)
            buf = np.zeros((M, 8, 8), np.int64)
            _lib.check(L.gm_acq_debug_stamps(eng._h, buf.ctypes.data_as(C.c_void_p)), "read")
            mx2, am2, sm2 = eng.metrics()
            assert again == got and (am2 == am).all()
            assert float(np.max(np.abs(mx2 - mx) / mx)) <= 2e-6 and float(np.max(np.abs(sm2 - sm) / sm)) <= 2e-6    # (another summation order)
        eng.close()
