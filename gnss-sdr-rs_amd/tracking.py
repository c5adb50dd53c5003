"""Mirror of the reference's tracking API over the C ABI (names follow src/tracking/do_tracking.rs
and src/utilities/multicast_ring_buffer.rs)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (AcqResult, CODE_INDEX_FAITHFUL, CODE_INDEX_FIXED, TrkCfg, TrkOut, TrkState, check, lib)  # noqa: F401

NUM_OF_CHANNELS = 15       # do_tracking.rs:18
LOCK_THRESHOLD = 15.0      # :16
MAX_LOST_EPOCHS = 20       # :17
LOOP_MS = 10               # :29


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class LoopFilter:
    """do_tracking.rs:52-71"""

    def __init__(self, noise_bw, dumping_ratio, gain):
        a, b = C.c_float(0), C.c_float(0)
        check(lib().gm_loop_filter_new(noise_bw, dumping_ratio, gain, C.byref(a), C.byref(b)), "LoopFilter::new")
        self.tau1, self.tau2 = a.value, b.value

    def update(self, d_err, err, dt):
        return float(lib().gm_loop_filter_update(self.tau1, self.tau2, d_err, err, dt))


class MulticastRingBuffer:
    """Device mirror of utilities::multicast_ring_buffer::MulticastRingBuffer (:36-130)."""

    def __init__(self, buf_size, device=None):
        _lib.init(device if device is not None else (_lib._initialised or 0))
        if buf_size <= 0 or buf_size & (buf_size - 1):
            raise AssertionError("Buffer size must be a power of two")
        h = C.c_void_p()
        check(lib().gm_ring_create(buf_size, C.byref(h)), "MulticastRingBuffer::new")
        self._h, self.buf_size = h, buf_size

    def close(self):
        if getattr(self, "_h", None):
            lib().gm_ring_destroy(self._h)
            self._h = None

    def __del__(self):      # (at interpreter shutdown the module globals close() uses may be gone already)
        try:
            self.close()
        except Exception:
            pass

    def write_samples(self, samples):
        s = np.ascontiguousarray(samples, np.complex64)
        check(lib().gm_ring_write_samples(self._h, _p(s), s.size), "write_samples")

    def write_samples_async(self, samples):
        """write_samples without blocking on the H2D copy: head advances once the samples are in HBM."""
        s = np.ascontiguousarray(samples, np.complex64)
        check(lib().gm_ring_write_samples_async(self._h, _p(s), s.size), "write_samples_async")

    def flush(self):
        check(lib().gm_ring_flush(self._h), "flush")

    def wait_head(self, required_idx, timeout_ms=1000):
        """The Condvar wait of do_tracking::run (:392-406).  True if head reached required_idx."""
        r = C.c_int(0)
        check(lib().gm_ring_wait_head(self._h, int(required_idx), int(timeout_ms), C.byref(r)), "wait_head")
        return bool(r.value)

    def get_head(self):
        h = C.c_uint64(0)
        check(lib().gm_ring_get_head(self._h, C.byref(h)), "get_head")
        return h.value

    def get_enqueued_head(self):
        """what the asynchronous writer has enqueued so far (>= get_head()): the head update_all_async's passes are gated on"""
        h = C.c_uint64(0)
        check(lib().gm_ring_get_enqueued_head(self._h, C.byref(h)), "get_enqueued_head")
        return h.value

    def copy_to_slice(self, start, n):
        d = np.zeros(n, np.complex64)
        check(lib().gm_ring_copy_to_slice(self._h, int(start), _p(d), n), "copy_to_slice")
        return d


def _out_tuple(o, arms=3):
    t = (o.ip, o.qp, o.ie, o.qe, o.il, o.ql)
    return t + (o.ive, o.qve, o.ivl, o.qvl) if arms == 5 else t


class TrackingManager:
    """TrackingManager::new (:336-348): n channels in one handle; process_channels' rayon fan-out
    (:364-371) becomes update_all()."""

    def __init__(self, fs, n_channels=NUM_OF_CHANNELS, n_arms=3, code_index_mode=CODE_INDEX_FAITHFUL,
                 early_late_space=0.5, very_early_late_space=1.0, boc11=False, codes=None, nominal_code_rate=0.0,
                 device=None, strict_libm=False, strict_sum_order=False, share_device=False):
        _lib.init(device if device is not None else (_lib._initialised or 0))
        cfg = TrkCfg()
        cfg.strict_libm = int(strict_libm)     # the carrier's cos / sin as glibc's cosf / sinf, bit for bit (gm_trk_cfg.strict_libm)
        cfg.share_device = int(share_device)          # a receiver: leave room for the other stages' kernels beside a tracking launch
        cfg.strict_sum_order = int(strict_sum_order)   # the sums in the reference's sample order (with strict_libm: bit-identical state)
        cfg.fs, cfg.n_channels, cfg.n_arms = fs, n_channels, n_arms
        cfg.early_late_space, cfg.very_early_late_space = early_late_space, very_early_late_space
        cfg.code_index_mode, cfg.boc11 = code_index_mode, int(boc11)
        self._codes = None
        if codes is not None:
            self._codes = np.ascontiguousarray(codes, np.int8)
            cfg.codes, cfg.n_codes, cfg.code_len = self._codes.ctypes.data, self._codes.shape[0], self._codes.shape[1]
        cfg.nominal_code_rate = nominal_code_rate
        h = C.c_void_p()
        check(lib().gm_trk_create(C.byref(cfg), C.byref(h)), "gm_trk_create")
        self._h, self.fs, self.n_channels, self.n_arms = h, float(fs), int(n_channels), int(n_arms)
        self.channels = [TrackingChannel(self, i) for i in range(n_channels)]

    def close(self):
        if getattr(self, "_h", None):
            lib().gm_trk_destroy(self._h)
            self._h = None

    def __del__(self):      # (at interpreter shutdown the module globals close() uses may be gone already)
        try:
            self.close()
        except Exception:
            pass

    def update_all(self, ring, max_epochs=1):
        """-> (outs [E][C][2*arms] f32, processed [E][C] u8, lost [E][C] u8, epochs_done)"""
        E, Cn = int(max_epochs), self.n_channels
        outs = (TrkOut * (E * Cn))()
        proc = np.zeros((E, Cn), np.uint8)
        lost = np.zeros((E, Cn), np.uint8)
        done = C.c_uint32(0)
        check(lib().gm_trk_update_all(self._h, ring._h, E, C.cast(outs, C.c_void_p), _p(proc), _p(lost),
                                      C.byref(done)), "gm_trk_update_all")
        o = np.frombuffer(outs, np.float32).reshape(E, Cn, 10)[:, :, :2 * self.n_arms].copy()
        return o, proc, lost, done.value

    def update_all_async(self, ring, max_epochs=1):
        """update_all without a host wait: ordered on the device behind what the ring's writer has enqueued; -> ticket"""
        tok = C.c_uint64(0)
        check(lib().gm_trk_update_all_async(self._h, ring._h, int(max_epochs), C.byref(tok)), "gm_trk_update_all_async")
        self._ticket_epochs = getattr(self, "_ticket_epochs", {})
        self._ticket_epochs[tok.value] = int(max_epochs)
        return tok.value

    def collect(self, ticket, wait=False, with_states=False):
        """-> None while the call is still running (wait = False), else update_all's (outs, processed, lost, epochs_done)
        (+ the list of channel states as they stood behind that call's passes when with_states)"""
        if ticket not in getattr(self, "_ticket_epochs", {}):      # let the library say so (GM_ERR_INVALID_ARG)
            ready = C.c_int(0)
            check(lib().gm_trk_collect(self._h, int(ticket), int(bool(wait)), None, None, None, None, None, C.byref(ready)), "gm_trk_collect")
            raise KeyError(ticket)
        E, Cn = self._ticket_epochs[ticket], self.n_channels
        outs = (TrkOut * (E * Cn))()
        proc = np.zeros((E, Cn), np.uint8)
        lost = np.zeros((E, Cn), np.uint8)
        done, ready = C.c_uint32(0), C.c_int(0)
        states = (TrkState * Cn)() if with_states else None
        rc = lib().gm_trk_collect(self._h, int(ticket), int(bool(wait)), C.cast(outs, C.c_void_p), _p(proc), _p(lost),
                                  C.cast(states, C.c_void_p) if with_states else None, C.byref(done), C.byref(ready))
        if rc != 0:
            del self._ticket_epochs[ticket]          # a failed collect has consumed the ticket (include/gnss_mi355x.h)
            check(rc, "gm_trk_collect")
        if not ready.value:
            return None
        del self._ticket_epochs[ticket]
        o = np.frombuffer(outs, np.float32).reshape(E, Cn, 10)[:, :, :2 * self.n_arms].copy()
        if with_states:
            return o, proc, lost, done.value, list(states)
        return o, proc, lost, done.value

    def get_states(self):
        """every channel's record in one call (gm_trk_get_states)"""
        states = (TrkState * self.n_channels)()
        check(lib().gm_trk_get_states(self._h, C.cast(states, C.c_void_p)), "gm_trk_get_states")
        return list(states)

    def set_states(self, states, which=None):
        arr = (TrkState * self.n_channels)(*states)
        w = None if which is None else np.ascontiguousarray(which, np.uint8)
        check(lib().gm_trk_set_states(self._h, C.cast(arr, C.c_void_p), _p(w) if w is not None else None), "gm_trk_set_states")

    def update_all_dev(self, ring, epochs):
        check(lib().gm_trk_update_all_dev(self._h, ring._h, int(epochs)), "gm_trk_update_all_dev")

    def synchronize(self):
        check(lib().gm_trk_synchronize(self._h), "gm_trk_synchronize")

    def set_stream(self, stream_ptr):
        check(lib().gm_trk_set_stream(self._h, C.c_void_p(stream_ptr)), "gm_trk_set_stream")

    def enable_timing(self, on=True):
        check(lib().gm_trk_enable_timing(self._h, int(on)), "gm_trk_enable_timing")

    def last_timing(self):
        ms, n = C.c_float(0), C.c_uint32(0)
        check(lib().gm_trk_last_timing(self._h, C.byref(ms), C.byref(n)), "gm_trk_last_timing")
        return ms.value, n.value


class TrackingChannel:
    """TrackingChannel (:88-327): a view of one channel of a TrackingManager handle."""

    def __init__(self, manager, id):
        self._m, self.id = manager, int(id)

    @property
    def state(self):
        s = TrkState()
        check(lib().gm_trk_get_state(self._m._h, self.id, C.byref(s)), "gm_trk_get_state")
        return s

    def set_state(self, **kw):
        s = self.state
        for k, v in kw.items():
            setattr(s, k, v)
        check(lib().gm_trk_set_state(self._m._h, self.id, C.byref(s)), "gm_trk_set_state")

    def __getattr__(self, name):   # carrier_freq, code_phase, next_sample_index, ... like the pub fields
        if name.startswith("_") or name in ("id",):
            raise AttributeError(name)
        s = self.state
        if hasattr(s, name):
            return getattr(s, name)
        raise AttributeError(name)

    def start(self, result):
        r = result if isinstance(result, AcqResult) else AcqResult(**{k: v for k, v in result.items()})
        check(lib().gm_trk_start(self._m._h, self.id, C.byref(r)), "TrackingChannel::start")

    def is_active(self):
        return bool(self.state.active)

    def reset(self):
        check(lib().gm_trk_reset(self._m._h, self.id), "TrackingChannel::reset")

    def get_ca_chip(self, phase):
        v = C.c_float(0)
        st = lib().gm_trk_get_ca_chip(self._m._h, self.id, phase, C.byref(v))
        if st == -5:
            raise IndexError("GPS_CA_CODE_32_PRN row out of bounds (the reference panics)")
        check(st, "get_ca_chip")
        return v.value

    def early_late_correlation(self, data_samples):
        d = np.ascontiguousarray(data_samples, np.complex64)
        o = TrkOut()
        st = lib().gm_trk_correlate(self._m._h, self.id, _p(d), d.size, C.byref(o))
        if st == -5:
            raise IndexError("out of range (the reference panics)")
        check(st, "early_late_correlation")
        return _out_tuple(o, self._m.n_arms)

    def do_work(self, data_samples):
        d = np.ascontiguousarray(data_samples, np.complex64)
        o = TrkOut()
        lost, lprn = C.c_uint8(0), C.c_uint8(0)
        st = lib().gm_trk_do_work(self._m._h, self.id, _p(d), d.size, C.byref(o), C.byref(lost), C.byref(lprn))
        if st == -5:
            raise IndexError("out of range (the reference panics)")
        check(st, "do_work")
        return _out_tuple(o, self._m.n_arms), (("SatelliteLost", lprn.value) if lost.value else None)
