import sys, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnss_sdr_rs_amd import _lib, acquisition as A, synth, tracking as T
_lib.init(0)
ca=A.ca_code_table(); fs=25e6; Cn=int(os.environ.get('TRK_C','32')); E=40; n=25000
sc=synth.tracking_scene(ca, fs, 0.0, list(range(1,Cn+1)), E+2, config_id=3, cn0=47.0)
ring=T.MulticastRingBuffer(1<<21); ring.write_samples(synth.to_c32(sc['x']))
mgr=T.TrackingManager(fs, n_channels=Cn, code_index_mode=1)
for i,s in enumerate(sc['sats']):
    mgr.channels[i].start(dict(prn=s['prn'], code_phase_samples=0, code_phase_chips=0.0, carrier_freq=s['doppler_hz']+20.0, fs=fs, mag_relative=1.0, sample_global_index=s['code_start'], doppler_bin=0))
L=_lib.lib()
_lib.check(L.gm_trk_debug_stamps(mgr._h, E, None),'arm')
mgr.update_all_dev(ring, E); mgr.synchronize()
buf=np.zeros((E,48),np.int64)
_lib.check(L.gm_trk_debug_stamps(mgr._h, E, buf.ctypes.data_as(C.c_void_p)),'read')
d=np.diff(buf[:,:8],axis=1)
names=['compute','reduce+barrier','wg-partial+publish','poll','totals','epilogue','barrier+copy']
print('per-phase cycles (median over epochs 5..):')
for i,nm in enumerate(names): print('  %-20s %8.0f'%(nm, np.median(d[5:,i])))
print('epoch total (stamp0->stamp0 next):', np.median(np.diff(buf[5:,0])))

ce=buf[5:,8:24]-buf[5:,0:1]; ba=buf[5:,24:40]-buf[5:,0:1]
print('per-wave compute end (cycles after wave0 epoch start), median:', np.median(ce,axis=0).astype(int))
print('per-wave barrier arrival, median:', np.median(ba,axis=0).astype(int))
w1 = buf[5:, 40:43] - buf[5:, 2:3]      # wave 1 of the serial section, relative to the reduce barrier's release (stamp 2)
print('wave 1 (code half) after the barrier: gathered %d, half done %d, next-epoch constants %d   | wave 0: gathered %d, half done %d, constants+outs %d' % (
    np.median(w1[:, 0]), np.median(w1[:, 1]), np.median(w1[:, 2]),
    np.median(buf[5:, 4] - buf[5:, 2]), np.median(buf[5:, 6] - buf[5:, 2]), np.median(buf[5:, 7] - buf[5:, 2])))
w1b = buf[5:, 43:45] - buf[5:, 2:3]
print('wave 1 detail: totals done %d, bookkeeping done %d' % (np.median(w1b[:, 0]), np.median(w1b[:, 1])))
print('extra poll rounds after the first sweep (median / mean over epochs): wave 0 %.1f / %.2f, wave 1 %.1f / %.2f' % (
    np.median(buf[5:, 45]), np.mean(buf[5:, 45]), np.median(buf[5:, 46]), np.mean(buf[5:, 46])))
