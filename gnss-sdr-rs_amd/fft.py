"""Mirror of the crate-root FFT<T> / RealFFT<T> wrappers (src/fft.rs:5-56) over the C ABI."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def supported_sizes():
    buf = np.zeros(64, np.uint32)
    n = lib().gm_fft_supported_sizes(_p(buf), buf.size)
    return [int(x) for x in buf[:n]]


class FFT:
    def __init__(self, len):
        self.len = int(len)
        _lib.init(_lib._initialised or 0)

    def execute(self, input, inverse=False):
        """in place on `input` (complex64 array of len, or batch*len) and returns it, like fft.rs:21-25"""
        assert input.dtype == np.complex64 and input.flags.c_contiguous and input.size % self.len == 0
        check(lib().gm_fft_c2c_f32(self.len, 1 if inverse else 0, _p(input), input.size // self.len), "FFT::execute")
        return input

    def power_spectrum(self, input):
        assert input.dtype == np.complex64 and input.size == self.len
        p = np.zeros(self.len, np.float32)
        check(lib().gm_fft_power_spectrum_f32(self.len, _p(input), _p(p)), "FFT::power_spectrum")
        return p


class RealFFT:
    def __init__(self, len):
        self.len = int(len)
        _lib.init(_lib._initialised or 0)

    def execute(self, input):
        a = np.ascontiguousarray(input, np.float32)
        out = np.zeros(self.len // 2 + 1, np.complex64)
        check(lib().gm_rfft_f32(self.len, _p(a), _p(out)), "RealFFT::execute")
        return out

    def power_spectrum(self, input):
        o = self.execute(input)
        return (o.real * o.real + o.imag * o.imag).astype(np.float32)
