"""Diagnostic: host wall clock of every call of the asynchronous receiver loop (front-end block -> tracking enqueue -> collect),
per block, to find where the host blocks.  TRK_SYNC=1: the synchronous per-block path for comparison."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnss_sdr_rs_amd import _lib, acquisition as A, tracking as T, frontend as F, synth
_lib.init(0)
fs, f_if, N = 16_367_600.0, 4_130_400.0, 16368
BLK, NB = 1 << int(os.environ.get("LOG2BLK", "19")), int(os.environ.get("NB", "60"))
rng = np.random.default_rng(1)
NBUF = min(NB, 8)                      # the blocks are reused cyclically (a long soak needs no long capture)
xi8 = rng.integers(-60, 60, (NBUF * BLK, 2), dtype=np.int8)
ring = T.MulticastRingBuffer(1 << 23)
fe = F.DigitalFrontend(f_if, fs, fs)
mgr = T.TrackingManager(fs, n_channels=15, code_index_mode=T.CODE_INDEX_FIXED, share_device=os.environ.get("SHARE", "1") == "1")
for ch in range(8):
    mgr.channels[ch].start(dict(prn=ch + 1, code_phase_samples=0, code_phase_chips=0.0, carrier_freq=100.0 * ch, fs=fs, mag_relative=1.0,
                                sample_global_index=100 * ch, doppler_bin=0))
sync = os.environ.get("TRK_SYNC") == "1"
E = 2 * (BLK // N) + 3
tickets, rows = [], []
t_start = time.perf_counter()
for b in range(NB):
    t0 = time.perf_counter()
    fe.write_ring(ring, xi8[(b % NBUF) * BLK:(b % NBUF + 1) * BLK])
    t1 = time.perf_counter()
    if sync:
        mgr.update_all(ring, E)
        t2 = t3 = time.perf_counter()
        nq = 0
    else:
        tickets.append(mgr.update_all_async(ring, E))
        t2 = time.perf_counter()
        nq = 0
        while tickets:
            r = mgr.collect(tickets[0], wait=len(tickets) >= 7)
            nq += 1
            if r is None:
                break
            tickets.pop(0)
        t3 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1, t3 - t2, len(tickets), nq))
ring.flush()
for tk in tickets:
    mgr.collect(tk, wait=True)
wall = time.perf_counter() - t_start
print("blocks %d, wall %.4f s = %.1f us per block (%.1f x real time)" % (NB, wall, wall / NB * 1e6, NB * BLK / fs / wall))
slow = [(i, round(sum(r[:3]) * 1e6)) for i, r in enumerate(rows) if i > 2 and sum(r[:3]) > 2e-3]
print("blocks above 2 ms of host time (index, us):", slow[:20], "of", len(rows))
for i, r in enumerate(rows[:int(os.environ.get("SHOW", "60"))]):
    print("%3d  fe %.0f us  enqueue %.0f us  collect %.0f us  outstanding %d  collect calls %d" % (i, r[0] * 1e6, r[1] * 1e6, r[2] * 1e6, r[3], r[4]))
