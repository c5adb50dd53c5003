"""Diagnostic: where the dispatcher puts the workgroups of the persistent tracking kernel at the BASELINE configs[4] geometry
(GM_DIAGNOSTICS=1 GM_TRK_STAMP_WG=-2: every workgroup records XCC_ID and HW_REG_HW_ID): which workgroups share a CU."""
import os, sys, ctypes as C
import numpy as np
os.environ["GM_DIAGNOSTICS"] = "1"; os.environ["GM_TRK_STAMP_WG"] = "-2"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnss_sdr_rs_amd import _lib, tracking as T
_lib.init(0)
fs, L, rate, Cn = 50.0e6, 4092, 1.023e6, int(os.environ.get("TRK_C", "36"))
n = 200000
rng = np.random.default_rng(5)
codes = np.where(rng.integers(0, 2, (Cn, L)) > 0, 1, -1).astype(np.int8)
x = (rng.standard_normal(3 * n) + 1j * rng.standard_normal(3 * n)).astype(np.complex64)
ring = T.MulticastRingBuffer(1 << 20); ring.write_samples(x)
mgr = T.TrackingManager(fs, n_channels=Cn, n_arms=5, code_index_mode=T.CODE_INDEX_FIXED, early_late_space=0.25, very_early_late_space=0.6,
                        boc11=True, codes=codes, nominal_code_rate=rate)
for j in range(Cn):
    mgr.channels[j].start(dict(prn=j + 1, code_phase_samples=0, code_phase_chips=0.0, carrier_freq=100.0, fs=fs, mag_relative=1.0, sample_global_index=0, doppler_bin=0))
    mgr.channels[j].set_state(code_rate=rate, num_samples_per_code=n)
Lb = _lib.lib()
E = 32
_lib.check(Lb.gm_trk_debug_stamps(mgr._h, E, None), 'arm')
mgr.update_all_dev(ring, 2); mgr.synchronize()
buf = np.zeros((E, 48), np.int64)
_lib.check(Lb.gm_trk_debug_stamps(mgr._h, E, buf.ctypes.data_as(C.c_void_p)), 'read')
v = buf.reshape(-1)
slots = ((Cn + 7) // 8) * 8
G = int(os.environ.get("TRK_G", "12"))
PACKED = os.environ.get("TRK_PACKED", "0") == "1"       # the packed layout (trk_kernels.hip): TRK_G = its workgroups per channel (14 at 36 channels)
per_xcd = (Cn * G + 7) // 8
def chan_of(b):
    if PACKED:
        num = (b & 7) * per_xcd + (b >> 3)
        return (num // G, num % G) if (b >> 3) < per_xcd and num < Cn * G else (None, None)
    return ((b >> 3) // G * 8 + (b & 7), (b >> 3) % G)
NB = 8 * per_xcd if PACKED else slots * G
place = {}
for b in range(NB):
    if v[b] == 0:
        continue
    xcc, hw = int(v[b]) >> 32, int(v[b]) & 0xffffffff
    cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 0x7
    ch, g = chan_of(b)
    place.setdefault((xcc, se, sh, cu), []).append((ch, g, b))
t0s = np.array([v[512 + b] for b in range(NB) if v[b]]); t1s = np.array([v[1024 + b] for b in range(NB) if v[b]])
base = t0s.min()
print("start spread (us): min 0, median %.2f, max %.2f; end (us after the first start): min %.2f median %.2f max %.2f" % (
    np.median(t0s - base) / 100, (t0s.max() - base) / 100, (t1s.min() - base) / 100, np.median(t1s - base) / 100, (t1s.max() - base) / 100))
chs = {}
for b in range(NB):
    if v[b]:
        ch = chan_of(b)[0]
        chs.setdefault(ch, []).append(((v[512 + b] - base) / 100, (v[1024 + b] - base) / 100))
for ch in sorted(chs):
    a_ = np.array(chs[ch])
    print("  channel %2d: starts %.1f .. %.1f us, ends %.1f .. %.1f us" % (ch, a_[:, 0].min(), a_[:, 0].max(), a_[:, 1].min(), a_[:, 1].max()))
print("workgroups recorded:", sum(len(p) for p in place.values()), "CUs used:", len(place))
from collections import Counter
print("tenants per CU:", Counter(len(p) for p in place.values()))
for k in sorted(place)[:8]:
    print(k, place[k])
