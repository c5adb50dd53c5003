"""Host mirror of rf::frontend::DigitalFrontend (src/rf/frontend.rs:6-62) over the C ABI.
Same constructor and `process_block` as the reference; `write_ring` is rf_thread's block step (rf_thread.rs:43-48)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FMT_C32, FMT_I8_IQ, check, lib

LUT_SIZE = 2048   # nco_lut.rs:4


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class DigitalFrontend:
    def __init__(self, f_if, fs_in, fs_out, device=None):
        _lib.init(device if device is not None else (_lib._initialised or 0))
        h = C.c_void_p()
        check(lib().gm_frontend_create(f_if, fs_in, fs_out, C.byref(h)), "DigitalFrontend::new")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().gm_frontend_destroy(self._h)
            self._h = None

    def __del__(self):      # (at interpreter shutdown the module globals close() uses may be gone already)
        try:
            self.close()
        except Exception:
            pass

    def nco(self):
        """(lut_re, lut_im, phase_step) of NcoLut::new (nco_lut.rs:25-42)"""
        re, im, st = np.zeros(LUT_SIZE, np.float32), np.zeros(LUT_SIZE, np.float32), C.c_float(0)
        check(lib().gm_frontend_lut(self._h, _p(re), _p(im), C.byref(st)), "gm_frontend_lut")
        return re, im, st.value

    def state(self):
        ph, br, bi = C.c_float(0), np.zeros(8, np.float32), np.zeros(8, np.float32)
        check(lib().gm_frontend_get_state(self._h, C.byref(ph), _p(br), _p(bi)), "gm_frontend_get_state")
        return ph.value, br, bi

    def set_state(self, phase_accumulator, bias_re, bias_im):
        br, bi = np.ascontiguousarray(bias_re, np.float32), np.ascontiguousarray(bias_im, np.float32)
        check(lib().gm_frontend_set_state(self._h, phase_accumulator, _p(br), _p(bi)), "gm_frontend_set_state")

    def process_block(self, raw_floats):
        """process_block(&mut [f32]) (:33-62): interleaved I/Q float32, in place."""
        assert raw_floats.dtype == np.float32 and raw_floats.flags.c_contiguous
        check(lib().gm_frontend_process_block(self._h, _p(raw_floats), raw_floats.size), "process_block")
        return raw_floats

    def process_dev(self, d_in, fmt, d_out, n_samples, stream=None):
        check(lib().gm_frontend_process_dev(self._h, d_in, fmt, d_out, n_samples, stream), "gm_frontend_process_dev")

    def synchronize(self):
        check(lib().gm_frontend_synchronize(self._h), "gm_frontend_synchronize")

    def debug_repairs(self):
        """runs of the speculative form the verification has had to repeat on this handle (diagnostic)"""
        n = C.c_uint32(0)
        check(lib().gm_frontend_debug_repairs(self._h, C.byref(n)), "gm_frontend_debug_repairs")
        return n.value

    def write_ring(self, ring, samples):
        """samples: complex64 array, or int8 array of interleaved I/Q."""
        s = np.ascontiguousarray(samples)
        if s.dtype == np.int8:
            n, fmt = s.size // 2, FMT_I8_IQ
        else:
            s = np.ascontiguousarray(s, np.complex64)
            n, fmt = s.size, FMT_C32
        check(lib().gm_frontend_write_ring(self._h, ring._h, _p(s), n, fmt), "gm_frontend_write_ring")


def process_dev_batch(frontends, d_in_ptrs, fmt, d_out_ptrs, n_samples, stream=None):
    """n independent streams in one launch (gm_frontend_process_dev_batch): frontends[i]: d_in_ptrs[i] -> d_out_ptrs[i]."""
    n = len(frontends)
    H = (C.c_void_p * n)(*[f._h for f in frontends])
    I = (C.c_void_p * n)(*d_in_ptrs)
    O = (C.c_void_p * n)(*d_out_ptrs)
    check(lib().gm_frontend_process_dev_batch(C.cast(H, C.c_void_p), n, C.cast(I, C.c_void_p), fmt, C.cast(O, C.c_void_p),
                                              n_samples, stream), "gm_frontend_process_dev_batch")
