"""Diagnostic: the persistent tracking kernel alone at the BASELINE configs[4] geometry (36 ch x 50 Msps, 4092-chip BOC(1,1), five
arms), timed by the library's own HIP events around the launch (no host launch / synchronise time in the figure): us per code
period, median of CFG5_RUNS launches of CFG5_PERIODS (6) periods.  GM_LIB_PATH selects an A/B build (same box, same call: the only comparison
that resolves a few per cent)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnss_sdr_rs_amd import _lib, tracking as T
_lib.init(0)
fs, L, rate, Cn, periods = 50.0e6, 4092, 1.023e6, int(os.environ.get("TRK_C", "36")), int(os.environ.get("CFG5_PERIODS", "6"))
n = int(round(fs / (rate / L)))
rng = np.random.default_rng(5)
codes = np.where(rng.integers(0, 2, (Cn, L)) > 0, 1, -1).astype(np.int8)
tt = np.arange((periods + 1) * n, dtype=np.float64)
cp = (tt * rate / fs) % L
sub = np.where((cp - np.floor(cp)) < 0.5, 1.0, -1.0).astype(np.float32)
ci = np.floor(cp).astype(np.int64)
x = (rng.standard_normal(tt.size) + 1j * rng.standard_normal(tt.size)).astype(np.complex64) * np.float32(8.0)
dopp = rng.uniform(-2000, 2000, Cn)
for c in range(Cn):
    x += (np.float32(0.6) * codes[c][ci] * sub * np.exp(2j * np.pi * dopp[c] * tt / fs)).astype(np.complex64)
ring = T.MulticastRingBuffer(1 << (21 if periods <= 9 else 23))
ring.write_samples(x)
mgr = T.TrackingManager(fs, n_channels=Cn, n_arms=5, code_index_mode=T.CODE_INDEX_FIXED, early_late_space=0.25, very_early_late_space=0.6,
                        boc11=True, codes=codes, nominal_code_rate=rate)
def restart():
    for j in range(Cn):
        mgr.channels[j].start(dict(prn=j + 1, code_phase_samples=0, code_phase_chips=0.0, carrier_freq=float(dopp[j]) + 10.0, fs=fs, mag_relative=1.0,
                                   sample_global_index=0, doppler_bin=0))
        mgr.channels[j].set_state(code_rate=rate, num_samples_per_code=n, carrier_phase=0.0, code_error=0.0, carrier_error=0.0, lost_counter=0)
mgr.enable_timing(True)
ts = []
for r in range(int(os.environ.get("CFG5_RUNS", "7")) + 1):
    restart()
    mgr.update_all_dev(ring, periods); mgr.synchronize()
    ms, nl = mgr.last_timing()
    if r:
        ts.append(ms * 1e3 / periods)
locked = sum(1 for c in mgr.channels if c.is_active() and c.lost_counter == 0)
st = mgr.channels[0].state
print("%s: us per code period: median %.2f  min %.2f  max %.2f   locked %d/%d   ch0 carrier %.6f code_rate %.4f" % (
    os.path.basename(os.environ.get("GM_LIB_PATH", "libgnss_mi355x.so")), np.median(ts), min(ts), max(ts), locked, Cn, st.carrier_freq, st.code_rate))
