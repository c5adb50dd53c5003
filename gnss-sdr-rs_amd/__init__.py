"""gnss-sdr-rs_amd — MI355X-native acquisition + tracking hot path of kewei/gnss-sdr-rs.

The product is the C-ABI shared library built from csrc/ (hand-written HIP for gfx950), declared in
include/gnss_mi355x.h.  This Python package is plumbing around it:
  _lib        ctypes loader (fails loudly when the library is missing; there is no CPU fallback)
  acquisition / tracking / fft   thin mirrors of the reference's Rust API names over the C ABI
  synth       deterministic synthetic IF scenes
  build       hipcc build recipe
"""
from . import _lib  # noqa: F401
from ._lib import GmError, lib, library_path  # noqa: F401
