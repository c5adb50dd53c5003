"""ctypes binding of host/receiver_harness.cpp (lib/libgm_receiver.so): the receiver chain of main.rs:182-227 run by the C++ stage
drivers of host/gnss_sdr.hpp — gnss::run_acquisition (do_acquisition.rs:241-327) and gnss::run_tracking (do_tracking.rs:384-415)
on threads of their own, the feeder on the caller's — over the C ABI.  bench.py's `receiver` leg and tests/test_gpu_stage_drivers.py
call it; no fallback: a missing library raises."""
import ctypes as C
import os

import numpy as np

from ._lib import AcqResult, GmError, TrkState, lib as _product_lib

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "lib", "libgm_receiver.so")


class RxCfg(C.Structure):
    _fields_ = [("device", C.c_int32), ("fs", C.c_float), ("f_if", C.c_float), ("freq_search_hz", C.c_float), ("freq_step_hz", C.c_float),
                ("n_integrations", C.c_uint32), ("n_channels", C.c_uint32), ("block_samples", C.c_uint32), ("ring_log2", C.c_uint32),
                ("decision_mode", C.c_int32), ("code_index_mode", C.c_int32), ("nav_mode", C.c_int32), ("fine_doppler", C.c_int32),
                ("async_tickets", C.c_int32), ("first_round_signal_ms", C.c_double), ("pre_samples", C.c_uint32),
                ("warmup_calls", C.c_uint32)]


class RxChannel(C.Structure):
    _fields_ = [("prn", C.c_uint8), ("active", C.c_uint8), ("bit_sync", C.c_uint8), ("frame_sync", C.c_uint8),
                ("lost_counter", C.c_uint32), ("carrier_freq", C.c_float), ("frame_sync_ind", C.c_uint32), ("epochs", C.c_uint64),
                ("n_frame_bits", C.c_uint64), ("start_index", C.c_uint64)]


class RxReport(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("wall_seconds", "signal_seconds", "seconds_frontend", "seconds_acquisition",
                                          "seconds_fine_doppler", "seconds_tracking", "seconds_nav_bits", "fe_block_first_s",
                                          "fe_block_median_s", "fe_block_max_after_first_s", "feeder_held_back_s")] + \
               [("dwells", C.c_uint32), ("channels_started", C.c_uint32), ("blocks", C.c_uint32), ("fe_runs_repaired", C.c_uint32),
                ("channel_epochs", C.c_uint64), ("tracking_passes", C.c_uint64)] + \
               [(k, C.c_double) for k in ("first_handover_signal_ms", "first_handover_wall_s", "first_bit_sync_signal_ms",
                                          "first_bit_sync_wall_s", "first_frame_sync_signal_ms", "first_frame_sync_wall_s")] + \
               [("channels", RxChannel * 32)]


_rx = None


def library_path():
    return _PATH


def lib():
    global _rx
    if _rx is not None:
        return _rx
    _product_lib()                      # the product library first (the harness links it by rpath $ORIGIN)
    if not os.path.exists(_PATH):
        raise ImportError(f"{_PATH} is missing: run `python __graft_entry__.py build`")
    L = C.CDLL(_PATH)
    L.gmrx_last_error.restype = C.c_char_p
    L.gmrx_abi_version.restype = C.c_int
    L.gmrx_receiver_run.restype = C.c_int
    L.gmrx_receiver_run.argtypes = [C.POINTER(RxCfg), C.c_void_p, C.c_size_t, C.POINTER(RxReport)]
    L.gmrx_tracking_ab.restype = C.c_int
    L.gmrx_tracking_ab.argtypes = [C.c_int32, C.c_float, C.c_uint32, C.c_int32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_uint32,
                                   C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _rx = L
    return L


def _check(rc, where):
    if rc != 0:
        raise GmError(rc, where, (lib().gmrx_last_error() or b"").decode())


def receiver_run(iq_i8, fs, f_if, *, freq_search_hz=14e3, freq_step_hz=500.0, n_integrations=10, n_channels=15, block_samples=1 << 19,
                 ring_log2=23, decision_mode=1, code_index_mode=1, nav_mode=1, fine_doppler=True, async_tickets=True,
                 first_round_signal_ms=10.0, pre_samples=64, warmup_calls=48, device=0):
    """Run the whole chain on `iq_i8` (interleaved int8 I/Q, 2 bytes per sample) and return the report as a dict."""
    x = np.ascontiguousarray(iq_i8, np.int8).reshape(-1)
    cfg = RxCfg(device, fs, f_if, freq_search_hz, freq_step_hz, n_integrations, n_channels, block_samples, ring_log2, decision_mode,
                code_index_mode, nav_mode, int(fine_doppler), int(async_tickets), float(first_round_signal_ms), pre_samples, warmup_calls)
    rep = RxReport()
    _check(lib().gmrx_receiver_run(C.byref(cfg), x.ctypes.data, x.size // 2, C.byref(rep)), "gmrx_receiver_run")
    out = {k: getattr(rep, k) for k, _ in RxReport._fields_ if k != "channels"}
    out["channels"] = [{k: getattr(rep.channels[c], k) for k, _ in RxChannel._fields_} for c in range(n_channels)]
    return out


def tracking_ab(samples_c64, results, fs, *, n_channels=15, code_index_mode=1, ring_log2=23, write_block=0, device=0):
    """gnss::run_tracking alone on a pre-loaded ring with the given acquisition results (dicts), synchronous loop and ticket loop:
    -> dict(sync=[TrkState], async_=[TrkState], epochs=(s, a), locked=(s, a), lost=(s, a), seconds=(s, a))."""
    x = np.ascontiguousarray(samples_c64, np.complex64)
    res = (AcqResult * len(results))()
    for i, r in enumerate(results):
        for k, _ in AcqResult._fields_:
            if k in r:
                setattr(res[i], k, r[k])
    s_sync, s_async = (TrkState * n_channels)(), (TrkState * n_channels)()
    epochs, locked, lost, secs = (C.c_uint64 * 2)(), (C.c_uint32 * 2)(), (C.c_uint32 * 2)(), (C.c_double * 2)()
    _check(lib().gmrx_tracking_ab(device, fs, n_channels, code_index_mode, ring_log2, x.ctypes.data, x.size, write_block,
                                  C.cast(res, C.c_void_p), len(results), C.cast(s_sync, C.c_void_p), C.cast(s_async, C.c_void_p),
                                  C.cast(epochs, C.c_void_p), C.cast(locked, C.c_void_p), C.cast(lost, C.c_void_p),
                                  C.cast(secs, C.c_void_p)), "gmrx_tracking_ab")
    return dict(sync=list(s_sync), async_=list(s_async), epochs=tuple(epochs), locked=tuple(locked), lost=tuple(lost),
                seconds=tuple(secs))
