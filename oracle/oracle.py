"""ctypes binding of the CPU ORACLE (oracle/liboracle.so) — test infrastructure, NOT product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Function names follow the reference items they restate (see gnss_oracle.h for file:line).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class AcqResult(C.Structure):
    _fields_ = [("prn", C.c_uint8), ("code_phase_samples", C.c_uint64), ("code_phase_chips", C.c_float),
                ("carrier_freq", C.c_float), ("fs", C.c_float), ("mag_relative", C.c_float),
                ("sample_global_index", C.c_uint64), ("doppler_bin", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class LoopFilter(C.Structure):
    _fields_ = [("tau1", C.c_float), ("tau2", C.c_float)]


class TrkChannel(C.Structure):
    _fields_ = [("id", C.c_uint8), ("prn", C.c_uint8), ("state", C.c_int32), ("state_prn", C.c_uint8),
                ("lost_counter", C.c_uint32), ("fs", C.c_float), ("next_sample_index", C.c_uint64),
                ("num_samples_per_code", C.c_uint64), ("carrier_freq", C.c_float),
                ("carrier_phase", C.c_float), ("carrier_error", C.c_float), ("carrier_nco", C.c_float),
                ("code_phase", C.c_float), ("code_error", C.c_float), ("code_nco", C.c_float),
                ("code_rate", C.c_float), ("i_prompt", C.c_float), ("q_prompt", C.c_float),
                ("pll_filter", LoopFilter), ("dll_filter", LoopFilter), ("code_index_mode", C.c_int32),
                ("n_arms", C.c_int32), ("el_space", C.c_float), ("vel_space", C.c_float), ("boc11", C.c_int32),
                ("custom_codes", C.c_void_p), ("n_codes", C.c_uint32), ("code_len", C.c_uint32)]


class Ring(C.Structure):
    _fields_ = [("buffer", C.c_void_p), ("buf_size", C.c_size_t), ("mask", C.c_size_t), ("head", C.c_uint64)]


CODE_INDEX_FAITHFUL, CODE_INDEX_FIXED = 0, 1


def build(native=False):
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    target = "liboracle_native.so" if native else "liboracle.so"
    subprocess.run(["make", "-C", _HERE, target], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return os.path.join(_HERE, target)


_libs = {}


def lib(native=False):
    key = bool(native)
    if key in _libs:
        return _libs[key]
    path = os.path.join(_HERE, "liboracle_native.so" if native else "liboracle.so")
    src = os.path.join(_HERE, "gnss_oracle.c")
    if (not os.path.exists(path)) or os.path.getmtime(path) < os.path.getmtime(src):
        build(native)
    L = C.CDLL(path)
    vp, f32, u64, sz = C.c_void_p, C.c_float, C.c_uint64, C.c_size_t
    L.orc_ca_code_row.argtypes = [C.c_int, vp]
    L.orc_generate_ca_code_samples.argtypes = [C.c_int, f32, f32, vp, sz]
    L.orc_generate_ca_code_samples.restype = C.c_long
    L.orc_num_samples_per_code.argtypes = [f32, f32]
    L.orc_num_samples_per_code.restype = sz
    L.orc_doppler_table_new.argtypes = [f32, f32, f32, sz, vp]
    L.orc_doppler_table_new.restype = f32
    L.orc_apply_doppler_shift.argtypes = [vp, vp, vp, sz]
    L.orc_fft_plan_create.argtypes = [sz, C.c_int]
    L.orc_fft_plan_create.restype = vp
    L.orc_fft_plan_destroy.argtypes = [vp]
    L.orc_fft_exec.argtypes = [vp, vp]
    L.orc_fft_forward.argtypes = [vp, sz]
    L.orc_fft_power_spectrum.argtypes = [vp, sz, vp]
    L.orc_rfft_forward.argtypes = [vp, sz, vp]
    L.orc_acq_worker_new.argtypes = [C.c_uint8, sz, f32]
    L.orc_acq_worker_new.restype = vp
    L.orc_acq_worker_new_custom.argtypes = [C.c_uint8, sz, f32, vp, sz, f32]
    L.orc_acq_worker_new_custom.restype = vp
    L.orc_acq_worker_free.argtypes = [vp]
    L.orc_acq_worker_code_fft.argtypes = [vp]
    L.orc_acq_worker_code_fft.restype = vp
    L.orc_search_satellite.argtypes = [vp, vp, sz, vp, vp, sz, u64, sz, C.POINTER(AcqResult), vp, vp, vp, vp, C.c_int]
    L.orc_is_good_satellite.argtypes = [vp, sz, f32, C.POINTER(f32)]
    L.orc_decide_from_metrics.argtypes = [vp, vp, vp, vp, sz, sz, C.c_uint8, f32, u64, f32, C.POINTER(AcqResult)]
    L.orc_acq_search_all.argtypes = [vp, sz, u64, vp, sz, vp, vp, sz, u64, sz, C.c_int, C.c_int, vp, vp, C.POINTER(u64)]
    L.orc_acq_mode_for.argtypes = [sz]
    L.orc_acq_pacing_and_list.argtypes = [C.c_int, C.c_uint32, C.POINTER(u64), C.POINTER(C.c_uint32)]
    L.orc_loop_filter_new.argtypes = [f32, f32, f32]
    L.orc_loop_filter_new.restype = LoopFilter
    L.orc_loop_filter_update.argtypes = [C.POINTER(LoopFilter), f32, f32, f32]
    L.orc_loop_filter_update.restype = f32
    TP = C.POINTER(TrkChannel)
    L.orc_trk_new.argtypes = [TP, C.c_uint8, f32]
    L.orc_trk_start.argtypes = [TP, C.POINTER(AcqResult)]
    L.orc_trk_is_active.argtypes = [TP]
    L.orc_trk_reset.argtypes = [TP]
    L.orc_trk_get_ca_chip.argtypes = [TP, f32, C.POINTER(f32)]
    L.orc_trk_early_late_correlation.argtypes = [TP, vp, vp, vp]
    L.orc_trk_early_late_correlation_ex.argtypes = [TP, vp, vp, vp]
    L.orc_trk_run_loop_filters.argtypes = [TP, f32, f32, f32, f32, f32, f32]
    L.orc_trk_do_work.argtypes = [TP, vp, vp, C.POINTER(C.c_uint8)]
    RP = C.POINTER(Ring)
    L.orc_ring_new.argtypes = [RP, sz]
    L.orc_ring_free.argtypes = [RP]
    L.orc_ring_write_samples.argtypes = [RP, vp, sz]
    L.orc_ring_get_head.argtypes = [RP]
    L.orc_ring_get_head.restype = u64
    L.orc_ring_copy_to_slice.argtypes = [RP, u64, vp, sz]
    L.orc_trk_update.argtypes = [TP, RP, vp, vp, C.POINTER(C.c_uint8)]
    L.orc_trk_update_ex.argtypes = [TP, RP, vp, vp, C.POINTER(C.c_uint8)]
    L.orc_trk_update_forced.argtypes = [TP, RP, vp, vp, vp, vp, C.POINTER(C.c_uint8)]
    L.orc_trk_process_channels.argtypes = [vp, C.c_int, RP, vp, C.c_size_t, C.c_int, C.c_int]
    L.orc_trk_process_channels.restype = C.c_int64
    _libs[key] = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.complex64)


# ---------------------------------------------------------------- C/A code
def ca_code_row(row):
    out = np.zeros(1023, np.int8)
    if lib().orc_ca_code_row(int(row), _p(out)):
        raise IndexError("row out of range (the reference panics)")
    return out


def ca_code_table():
    return np.stack([ca_code_row(r) for r in range(32)])


def generate_ca_code_samples(prn, code_rate, fs):
    n = lib().orc_num_samples_per_code(code_rate, fs)
    out = np.zeros(max(n, 1), np.int8)
    r = lib().orc_generate_ca_code_samples(int(prn), code_rate, fs, _p(out), out.size)
    if r < 0:
        raise IndexError("generate_ca_code_samples: the reference would panic (prn or chip index out of bounds)")
    return out[:r]


def num_samples_per_code(code_rate, fs):
    return int(lib().orc_num_samples_per_code(code_rate, fs))


# ---------------------------------------------------------------- Doppler
class DopplerShiftTable:
    """acquisition::doppler_shift::DopplerShiftTable (doppler_shift.rs:5-22)"""

    def __init__(self, f_if, doppler_freq_hz, fs, num_samples):
        self.table = np.zeros(num_samples, np.complex64)
        self.doppler_freq_hz = float(lib().orc_doppler_table_new(f_if, doppler_freq_hz, fs, num_samples, _p(self.table)))


def apply_doppler_shift(samples, table, output):
    s = _c64(samples)
    t = table.table if isinstance(table, DopplerShiftTable) else _c64(table)
    assert output.dtype == np.complex64 and output.flags.c_contiguous
    lib().orc_apply_doppler_shift(_p(s), _p(t), _p(output), s.size)
    return output


# ---------------------------------------------------------------- FFT
def fft(x, inverse=False):
    d = _c64(x).copy()
    plan = lib().orc_fft_plan_create(d.size, 1 if inverse else 0)
    lib().orc_fft_exec(plan, _p(d))
    lib().orc_fft_plan_destroy(plan)
    return d


def rfft(x):
    a = np.ascontiguousarray(x, np.float32)
    out = np.zeros(a.size // 2 + 1, np.complex64)
    lib().orc_rfft_forward(_p(a), a.size, _p(out))
    return out


# ---------------------------------------------------------------- Acquisition
class AcquisitionWorker:
    """acquisition::do_acquisition::AcquisitionWorker (do_acquisition.rs:118-239)"""

    def __init__(self, prn, fft_size, freq_sampling_hz, code=None, code_rate=1.023e6, native=False):
        self._L = lib(native)
        self.prn, self.fft_size, self.fs = int(prn), int(fft_size), float(freq_sampling_hz)
        if code is None:
            self._h = self._L.orc_acq_worker_new(self.prn, self.fft_size, self.fs)
        else:
            c = np.ascontiguousarray(code, np.int8)
            self._h = self._L.orc_acq_worker_new_custom(self.prn, self.fft_size, self.fs, _p(c), c.size, code_rate)
        if not self._h:
            raise ValueError("AcquisitionWorker::new would panic (code length != fft_size or bad prn)")

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_acq_worker_free(self._h)
            self._h = None

    @property
    def ca_code_samples_fft(self):
        ptr = self._L.orc_acq_worker_code_fft(self._h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), (self.fft_size * 2,)).view(np.complex64).copy()

    def search_satellite(self, samples_chunk, doppler_table, local_tail, num_integrations,
                         want_planes=False, no_early_exit=False):
        """Returns AcqResult dict or None; with want_planes also (bin_max, bin_argmax, bin_sum, bins_done)."""
        s = _c64(samples_chunk)
        D = len(doppler_table)
        ptrs = (C.c_void_p * D)(*[t.table.ctypes.data for t in doppler_table])
        freqs = np.array([t.doppler_freq_hz for t in doppler_table], np.float32)
        res = AcqResult()
        bmax = np.zeros(D, np.float32)
        barg = np.zeros(D, np.uint32)
        bsum = np.zeros(D, np.float32)
        done = C.c_uint32(0)
        rc = self._L.orc_search_satellite(self._h, _p(s), s.size, C.cast(ptrs, C.c_void_p), _p(freqs), D,
                                          int(local_tail), int(num_integrations), C.byref(res),
                                          _p(bmax), _p(barg), _p(bsum), C.addressof(done), int(no_early_exit))
        if rc < 0:
            raise IndexError("search_satellite: slice out of range (the reference panics)")
        out = res.as_dict() if rc == 1 else None
        if want_planes:
            return out, (bmax, barg, bsum, done.value)
        return out


def search_all(workers, mask, samples_chunk, doppler_table, local_tail, num_integrations, n_threads=1,
               no_early_exit=False, native=False):
    """run()'s rayon fan-out (do_acquisition.rs:302-313).  Returns (results list[dict|None], cells)."""
    L = lib(native)
    s = _c64(samples_chunk)
    D = len(doppler_table)
    ptrs = (C.c_void_p * D)(*[t.table.ctypes.data for t in doppler_table])
    freqs = np.array([t.doppler_freq_hz for t in doppler_table], np.float32)
    W = len(workers)
    wp = (C.c_void_p * W)(*[w._h for w in workers])
    res = (AcqResult * W)()
    found = np.zeros(W, np.uint8)
    cells = C.c_uint64(0)
    rc = L.orc_acq_search_all(C.cast(wp, C.c_void_p), W, int(mask), _p(s), s.size, C.cast(ptrs, C.c_void_p),
                              _p(freqs), D, int(local_tail), int(num_integrations), int(n_threads),
                              int(no_early_exit), C.cast(res, C.c_void_p), _p(found), C.byref(cells))
    if rc:
        raise IndexError("search_satellite: slice out of range")
    return [res[i].as_dict() if found[i] else None for i in range(W)], cells.value


def is_good_satellite(power, max_val):
    p = np.ascontiguousarray(power, np.float32)
    s = C.c_float(0)
    ok = lib().orc_is_good_satellite(_p(p), p.size, max_val, C.byref(s))
    return bool(ok), s.value


def decide_from_metrics(bin_max, bin_argmax, bin_sum, table_freq, fft_size, prn, fs, local_tail, threshold=7.0):
    res = AcqResult()
    a = np.ascontiguousarray(bin_max, np.float32)
    b = np.ascontiguousarray(bin_argmax, np.uint32)
    c = np.ascontiguousarray(bin_sum, np.float32)
    f = np.ascontiguousarray(table_freq, np.float32)
    rc = lib().orc_decide_from_metrics(_p(a), _p(b), _p(c), _p(f), a.size, fft_size, prn, fs, local_tail,
                                       threshold, C.byref(res))
    return res.as_dict() if rc else None


class AcquisitionManager:
    """do_acquisition.rs:39-74"""
    COLD, WARM, STEADY = 0, 1, 2

    def __init__(self):
        self.mode = self.COLD

    def update_mode(self, tracked_count):
        self.mode = lib().orc_acq_mode_for(int(tracked_count))

    def get_pacing_and_list(self, active_prns):
        am = 0
        for p in active_prns:
            am |= 1 << (p - 1)
        iv, m = C.c_uint64(0), C.c_uint32(0)
        lib().orc_acq_pacing_and_list(self.mode, am, C.byref(iv), C.byref(m))
        return iv.value, m.value


# ---------------------------------------------------------------- Tracking
def loop_filter_new(bw, zeta, gain):
    return lib().orc_loop_filter_new(bw, zeta, gain)


class TrackingChannel:
    """tracking::do_tracking::TrackingChannel (do_tracking.rs:88-327)"""

    def __init__(self, id, fs, code_index_mode=CODE_INDEX_FAITHFUL, n_arms=3, el_space=0.0, vel_space=0.0, boc11=False,
                 codes=None, code_rate=None):
        self.c = TrkChannel()
        lib().orc_trk_new(C.byref(self.c), id, fs)
        self.c.code_index_mode = code_index_mode
        self.c.n_arms, self.c.el_space, self.c.vel_space, self.c.boc11 = n_arms, el_space, vel_space, int(boc11)
        self._codes = None
        if codes is not None:   # generalisation (no reference code): custom chip table
            self._codes = np.ascontiguousarray(codes, np.int8)
            self.c.custom_codes, self.c.n_codes, self.c.code_len = self._codes.ctypes.data, self._codes.shape[0], self._codes.shape[1]
            if code_rate:
                self.c.code_rate = code_rate
            self.c.num_samples_per_code = int(np.round(np.float32(fs) / (np.float32(self.c.code_rate) / np.float32(self._codes.shape[1]))))

    def start(self, result):
        r = result if isinstance(result, AcqResult) else AcqResult(**result)
        lib().orc_trk_start(C.byref(self.c), C.byref(r))

    def is_active(self):
        return bool(lib().orc_trk_is_active(C.byref(self.c)))

    def reset(self):
        lib().orc_trk_reset(C.byref(self.c))

    def get_ca_chip(self, phase):
        v = C.c_float(0)
        if lib().orc_trk_get_ca_chip(C.byref(self.c), phase, C.byref(v)):
            raise IndexError("get_ca_chip: GPS_CA_CODE_32_PRN row out of bounds (the reference panics)")
        return v.value

    def early_late_correlation(self, data_samples, want_f64=False):
        d = _c64(data_samples).copy()
        assert d.size >= self.c.num_samples_per_code
        out = np.zeros(6, np.float32)
        acc = np.zeros(6, np.float64)
        rc = lib().orc_trk_early_late_correlation(C.byref(self.c), _p(d), _p(out), _p(acc) if want_f64 else None)
        if rc:
            raise IndexError("get_ca_chip out of bounds")
        return (out, acc) if want_f64 else out

    def early_late_correlation_ex(self, data_samples):
        """generalised (arms / BOC / custom code): returns (out10 f32, acc10 f64)"""
        d = _c64(data_samples).copy()
        assert d.size >= self.c.num_samples_per_code
        out = np.zeros(10, np.float32)
        acc = np.zeros(10, np.float64)
        if lib().orc_trk_early_late_correlation_ex(C.byref(self.c), _p(d), _p(out), _p(acc)):
            raise IndexError("code table row out of bounds")
        return out, acc

    def run_loop_filters(self, i_p, q_p, i_e, q_e, i_l, q_l):
        lib().orc_trk_run_loop_filters(C.byref(self.c), i_p, q_p, i_e, q_e, i_l, q_l)

    def do_work(self, data_samples):
        d = _c64(data_samples).copy()
        out = np.zeros(6, np.float32)
        mp = C.c_uint8(255)
        rc = lib().orc_trk_do_work(C.byref(self.c), _p(d), _p(out), C.byref(mp))
        if rc < 0:
            raise IndexError("get_ca_chip out of bounds")
        return out, (("SatelliteLost", mp.value) if rc == 1 else None)

    def update(self, ring):
        scratch = np.zeros(max(int(self.c.num_samples_per_code) * 2, 16), np.complex64)
        out = np.zeros(6, np.float32)
        mp = C.c_uint8(255)
        rc = lib().orc_trk_update(C.byref(self.c), C.byref(ring.r), _p(scratch), _p(out), C.byref(mp))
        if rc < 0:
            raise IndexError("update: the reference would panic")
        return rc, out, (("SatelliteLost", mp.value) if rc == 2 else None)


    def update_ex(self, ring):
        """update() generalised to the channel's code length / arm count / BOC flag -> (rc, out10, msg)"""
        scratch = np.zeros(max(int(self.c.num_samples_per_code) * 2, 16), np.complex64)
        out = np.zeros(10, np.float32)
        mp = C.c_uint8(255)
        rc = lib().orc_trk_update_ex(C.byref(self.c), C.byref(ring.r), _p(scratch), _p(out), C.byref(mp))
        if rc < 0:
            raise IndexError("update_ex: the reference would panic")
        return rc, out, (("SatelliteLost", mp.value) if rc == 2 else None)


    def update_forced(self, ring, forced):
        """Teacher-forced update (see orc_trk_update_forced): -> (rc, computed10 f32, computed10 f64-accumulated, msg);
        state advanced with `forced`."""
        scratch = np.zeros(max(int(self.c.num_samples_per_code) * 2, 16), np.complex64)
        out = np.zeros(10, np.float32)
        f = np.zeros(10, np.float32)
        f[:len(forced)] = forced
        mp = C.c_uint8(255)
        acc = np.zeros(10, np.float64)
        rc = lib().orc_trk_update_forced(C.byref(self.c), C.byref(ring.r), _p(scratch), _p(out), _p(acc), _p(f), C.byref(mp))
        if rc < 0:
            raise IndexError("update_forced: the reference would panic")
        return rc, out, acc, (("SatelliteLost", mp.value) if rc == 2 else None)


def process_channels(channels, ring, max_passes, n_threads=1, native=False):
    """TrackingManager::process_channels (do_tracking.rs:351-371) looped like run() (:384-415): one task per channel.
    channels: list of TrackingChannel; their state is updated in place.  Returns channel-epochs processed."""
    arr = (TrkChannel * len(channels))(*[ch.c for ch in channels])
    stride = max(int(max(ch.c.num_samples_per_code for ch in channels)) * 2, 16)
    stride = max(stride, int(channels[0].c.fs / 1000.0) * 2)
    scratch = np.zeros(stride * len(channels), np.complex64)
    done = lib(native).orc_trk_process_channels(C.cast(arr, C.c_void_p), len(channels), C.byref(ring.r), _p(scratch), stride,
                                                max_passes, n_threads)
    if done < 0:
        raise IndexError("process_channels: the reference would panic")
    for ch, c in zip(channels, arr):
        C.memmove(C.byref(ch.c), C.byref(c), C.sizeof(TrkChannel))
    return int(done)


def finer_doppler(samples, code_phase, chips, fs, size_signal_use, code_rate=1.023e6, native=False):
    """finer_doppler (acquisition_bk.rs:215-302, legacy) on a c32 snapshot.  Returns dict(peak_index, peak_mag, freq_hz,
    upper_half, fft_size)."""
    L = lib(native)
    L.orc_finer_doppler.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_float, C.c_float,
                                    C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                    C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    L.orc_finer_doppler.restype = C.c_int
    s = _c64(samples)
    ch = np.ascontiguousarray(chips, np.int8)
    idx, mag, f, up, n = C.c_uint64(0), C.c_float(0), C.c_float(0), C.c_int(0), C.c_size_t(0)
    rc = L.orc_finer_doppler(_p(s), s.size, int(code_phase), _p(ch), ch.size, code_rate, fs, int(size_signal_use),
                             C.byref(idx), C.byref(mag), C.byref(f), C.byref(up), C.byref(n))
    if rc:
        raise IndexError("finer_doppler: slice out of range (the legacy would panic)")
    return dict(peak_index=idx.value, peak_mag=mag.value, freq_hz=f.value, upper_half=bool(up.value), fft_size=n.value)


class NavSyncState(C.Structure):
    _fields_ = [("fixed", C.c_int), ("flag_bit_sync", C.c_int), ("flag_frame_sync", C.c_int), ("sync_sw", C.c_int),
                ("loop_sw", C.c_int), ("biti", C.c_uint64), ("frame_sync_ind", C.c_uint64), ("bit_code_cnt", C.c_uint64),
                ("sf_buffer_loc", C.c_uint64), ("sf_cnt", C.c_uint64), ("sf_start_biti", C.c_uint64),
                ("tow_expected_ind", C.c_uint64), ("bit_sync_buff", C.c_uint64 * 20), ("i_p", C.c_float),
                ("polarity", C.c_int8), ("frame_bits", C.POINTER(C.c_int8)), ("n_frame_bits", C.c_size_t),
                ("cap_frame_bits", C.c_size_t), ("buff_preamble", C.c_int8 * 8), ("n_preamble", C.c_size_t),
                ("last_bit", C.c_int8)]


class NavSyncStatus:
    """decoding.rs NavSyncStatus + nav_decoding's per-epoch step (:40-227, legacy)"""

    def __init__(self, fixed=False):
        self.s = NavSyncState()
        L = lib()
        L.orc_nav_sync_new.argtypes = [C.c_void_p, C.c_int]
        L.orc_nav_sync_new.restype = None
        L.orc_nav_sync_free.argtypes = [C.c_void_p]
        L.orc_nav_sync_free.restype = None
        L.orc_nav_sync_update.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_uint64, C.c_uint64]
        L.orc_nav_sync_update.restype = C.c_int
        L.orc_nav_sync_new(C.byref(self.s), int(fixed))

    def __del__(self):
        lib().orc_nav_sync_free(C.byref(self.s))

    def update(self, old_i_prompt, i_prompt, cnt, buff_loc=0):
        return lib().orc_nav_sync_update(C.byref(self.s), old_i_prompt, i_prompt, int(cnt), int(buff_loc))

    def frame_bits(self):
        return np.array([self.s.frame_bits[i] for i in range(self.s.n_frame_bits)], np.int8)


def nav_parity_check(bits32):
    L = lib()
    L.orc_nav_parity_check.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.orc_nav_parity_check.restype = C.c_int
    b = np.ascontiguousarray(bits32, np.int8)
    ref = C.c_int(0)
    ok = L.orc_nav_parity_check(_p(b), C.byref(ref))
    return bool(ok), bool(ref.value)


class FrontendState(C.Structure):
    _fields_ = [("lut_re", C.c_float * 2048), ("lut_im", C.c_float * 2048), ("phase_accumulator", C.c_float),
                ("phase_step", C.c_float), ("bias_re", C.c_float * 8), ("bias_im", C.c_float * 8), ("alpha", C.c_float),
                ("con", C.c_float)]


class DigitalFrontend:
    """rf::frontend::DigitalFrontend (frontend.rs:6-62) with NcoLut (nco_lut.rs:17-42) and DcRemoverSimd (dc_remove.rs)."""

    def __init__(self, f_if, fs_in, fs_out, native=False):
        self.s = FrontendState()
        self._native = native
        L = lib(native)
        L.orc_frontend_new.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.orc_frontend_new.restype = None
        L.orc_frontend_process_block.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_frontend_process_block.restype = None
        L.orc_frontend_new(C.byref(self.s), f_if, fs_in, fs_out)

    def process_block(self, raw_floats):
        """in place on a float32 array of interleaved I/Q"""
        assert raw_floats.dtype == np.float32 and raw_floats.flags.c_contiguous
        lib(self._native).orc_frontend_process_block(C.byref(self.s), _p(raw_floats), raw_floats.size)
        return raw_floats


class MulticastRingBuffer:
    """utilities::multicast_ring_buffer::MulticastRingBuffer (multicast_ring_buffer.rs:36-130)"""

    def __init__(self, buf_size):
        self.r = Ring()
        if lib().orc_ring_new(C.byref(self.r), buf_size):
            raise AssertionError("Buffer size must be a power of two")

    def __del__(self):
        lib().orc_ring_free(C.byref(self.r))

    def write_samples(self, samples):
        s = _c64(samples)
        lib().orc_ring_write_samples(C.byref(self.r), _p(s), s.size)

    def get_head(self):
        return int(lib().orc_ring_get_head(C.byref(self.r)))

    def copy_to_slice(self, start, n):
        d = np.zeros(n, np.complex64)
        lib().orc_ring_copy_to_slice(C.byref(self.r), start, _p(d), n)
        return d

    def raw(self):
        return np.ctypeslib.as_array(C.cast(self.r.buffer, C.POINTER(C.c_float)), (self.r.buf_size * 2,)).view(np.complex64)
