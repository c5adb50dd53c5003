"""Diagnostic: where one workgroup of acq_corr_kernel spends its cycles (per-wave phase stamps, configs[1]).  Needs the DIAGNOSTIC
build of the library (the product library does not carry the stamped kernels):
    GM_EXTRA_FLAGS=-DGM_DIAG_STAMPS GM_LIB_SUFFIX=_diag python gnss-sdr-rs_amd/build.py
    GM_LIB_PATH=gnss-sdr-rs_amd/lib/libgnss_mi355x_diag.so python tools/acq_stamps.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnss_sdr_rs_amd import _lib, acquisition as A, synth
_lib.init(0)
ca = A.ca_code_table(); sc = synth.cfg2_scene(ca); x = synth.to_i8_iq(sc["x"])
eng = A.AcquisitionEngine(sc["fs"], sc["f_if"], sc["N"], doppler_hz=sc["doppler_hz"], n_integrations=sc["M"])
eng.search(x)
L = _lib.lib()
_lib.check(L.gm_acq_debug_stamps(eng._h, None), "arm")
eng.search(x)
buf = np.zeros((sc["M"], 8, 8), np.int64)
_lib.check(L.gm_acq_debug_stamps(eng._h, buf.ctypes.data_as(C.c_void_p)), "read")
names = ["loads+pass0.s1", "barrier1", "pass0.s2(scatter)", "barrier2+pass1.s1", "barrier3", "pass1.s2", "barrier4"]
t0 = buf[:, :, 0].min(axis=1, keepdims=True)
for w in range(8):
    d = np.diff(buf[2:, w, :], axis=1)
    tot = int(np.median(buf[3:, w, 0] - buf[2:-1, w, 0]))
    last = int(np.median(buf[3:, w, 0] - buf[2:-1, w, 7]))
    print("wave", w, " ".join("%s=%d" % (n, np.median(d[:, i])) for i, n in enumerate(names)), "last-pass+acc=%d" % last, "| total", tot)
