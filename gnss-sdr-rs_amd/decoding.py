"""Host mirror of the legacy decoding.rs pieces that are well defined: NavSyncStatus + nav_decoding's
per-epoch step up to frame sync (decoding.rs:40-227) and the word parity (:259-352).  Pure host work in the C-ABI
library (no GPU needed)."""
import ctypes as C

import numpy as np

from ._lib import NavStatus, check, lib

NAV_FAITHFUL, NAV_FIXED = 0, 1
GPS_CA_PREAMBLE = (1, -1, -1, -1, 1, -1, 1, 1)   # gps_property_constants.rs:13


class NavSyncStatus:
    def __init__(self, mode=NAV_FAITHFUL):
        h = C.c_void_p()
        check(lib().gm_nav_sync_create(mode, C.byref(h)), "NavSyncStatus::new")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().gm_nav_sync_destroy(self._h)
            self._h = None

    def __del__(self):      # (at interpreter shutdown the module globals close() uses may be gone already)
        try:
            self.close()
        except Exception:
            pass

    def update(self, old_i_prompt, i_prompt, cnt, buff_loc=0):
        """nav_decoding's step (:102-145) -> dict of NavStatus fields"""
        st = NavStatus()
        check(lib().gm_nav_sync_update(self._h, old_i_prompt, i_prompt, int(cnt), int(buff_loc), C.byref(st)), "nav_decoding")
        return {k: getattr(st, k) for k, _ in NavStatus._fields_}

    def update_many(self, old_i_prompt0, i_prompts, cnt0, buff_loc=0):
        """the same step for the consecutive epochs cnt0 .. (gm_nav_sync_update_many) -> (status dict after the last one,
        index of the epoch of this call at which bit sync / frame sync first appeared, or -1)"""
        ip = np.ascontiguousarray(i_prompts, np.float32)
        st = NavStatus()
        fb, ff = C.c_int64(-1), C.c_int64(-1)
        check(lib().gm_nav_sync_update_many(self._h, float(old_i_prompt0), ip.ctypes.data_as(C.c_void_p), 1, ip.size, int(cnt0), int(buff_loc),
                                            C.byref(st), C.byref(fb), C.byref(ff)), "nav_decoding (many)")
        return {k: getattr(st, k) for k, _ in NavStatus._fields_}, fb.value, ff.value

    def frame_bits(self):
        n = C.c_size_t(0)
        check(lib().gm_nav_sync_frame_bits(self._h, None, 0, C.byref(n)), "frame_bits")
        out = np.zeros(max(n.value, 1), np.int8)
        check(lib().gm_nav_sync_frame_bits(self._h, out.ctypes.data_as(C.c_void_p), out.size, C.byref(n)), "frame_bits")
        return out[:n.value]

    def histogram(self):
        h = np.zeros(20, np.uint64)
        check(lib().gm_nav_sync_histogram(self._h, h.ctypes.data_as(C.c_void_p)), "histogram")
        return h


def parity_check(bits32):
    """-> (ok, ref_sum_zero) for 32 symbols in +-1 form [D29*, D30*, d1..d24, D25..D30]"""
    b = np.ascontiguousarray(bits32, np.int8)
    assert b.size == 32
    ok, ref = C.c_int(0), C.c_int(0)
    check(lib().gm_nav_parity_check(b.ctypes.data_as(C.c_void_p), C.byref(ok), C.byref(ref)), "parity_check")
    return bool(ok.value), bool(ref.value)
