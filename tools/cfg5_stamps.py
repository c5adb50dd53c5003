"""Diagnostic: per-phase stamps of one workgroup of the persistent tracking kernel at the BASELINE configs[4] geometry (36 ch x 50 Msps,
4092-chip BOC(1,1), five arms, 200 000 samples per code period): compute / reduce / publish / poll / totals / epilogue per epoch,
per-wave compute end.  GM_TRK_STAMP_WG (with GM_DIAGNOSTICS=1) picks the workgroup."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnss_sdr_rs_amd import _lib, tracking as T
_lib.init(0)
fs, L, rate, Cn, periods = 50.0e6, 4092, 1.023e6, 36, 6
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
ring = T.MulticastRingBuffer(1 << 21)
ring.write_samples(x)
mgr = T.TrackingManager(fs, n_channels=Cn, n_arms=5, code_index_mode=T.CODE_INDEX_FIXED, early_late_space=0.25, very_early_late_space=0.6,
                        boc11=True, codes=codes, nominal_code_rate=rate)
def restart():
    for j in range(Cn):
        mgr.channels[j].start(dict(prn=j + 1, code_phase_samples=0, code_phase_chips=0.0, carrier_freq=float(dopp[j]) + 10.0, fs=fs, mag_relative=1.0,
                                   sample_global_index=0, doppler_bin=0))
        mgr.channels[j].set_state(code_rate=rate, num_samples_per_code=n, carrier_phase=0.0, code_error=0.0, carrier_error=0.0, lost_counter=0)
restart(); mgr.update_all_dev(ring, periods); mgr.synchronize(); restart()
Lb = _lib.lib()
E = periods
_lib.check(Lb.gm_trk_debug_stamps(mgr._h, E, None), 'arm')
mgr.update_all_dev(ring, E); mgr.synchronize()
buf = np.zeros((E, 48), np.int64)
_lib.check(Lb.gm_trk_debug_stamps(mgr._h, E, buf.ctypes.data_as(C.c_void_p)), 'read')
d = np.diff(buf[:, :8], axis=1)
names = ['compute', 'reduce+barrier', 'wg-partial+publish', 'poll', 'totals', 'epilogue', 'barrier+copy']
print('per-phase 10 ns ticks (s_memrealtime, median over epochs 1..):')
for i, nm in enumerate(names):
    print('  %-20s %8.0f' % (nm, np.median(d[1:, i])))
print('epoch total (stamp0 -> next stamp0):', np.median(np.diff(buf[1:, 0])), ' all epochs:', np.diff(buf[:, 0]))
ce = buf[1:, 8:16] - buf[1:, 0:1]
ba = buf[1:, 24:32] - buf[1:, 0:1]
print('per-wave compute end after the epoch start, median:', np.median(ce, axis=0).astype(int))
print('per-wave barrier arrival, median:', np.median(ba, axis=0).astype(int))
