"""Mirror of the reference's acquisition API over the C ABI (names follow
src/acquisition/do_acquisition.rs and src/acquisition/doppler_shift.rs)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import AcqCfg, AcqResult, FMT_C32, FMT_I8_IQ, FMT_I8_REAL, check, lib

PRN_SEARCH_ACQUISITION_TOTAL = 32      # do_acquisition.rs:22
FREQ_SEARCH_ACQUISITION_HZ = 14e3      # :20
FREQ_SEARCH_STEP_HZ = 500              # :21
LONG_SAMPLES_LENGTH = 10               # :23


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def generate_ca_code_samples(prn, code_rate, f_sampling):
    """utilities::ca_code::generate_ca_code_samples (ca_code.rs:12-27)"""
    n = C.c_size_t(0)
    st = lib().gm_generate_ca_code_samples(prn, code_rate, f_sampling, None, 0, C.byref(n))
    check(st, "generate_ca_code_samples")
    out = np.zeros(n.value, np.int8)
    check(lib().gm_generate_ca_code_samples(prn, code_rate, f_sampling, _p(out), out.size, C.byref(n)),
          "generate_ca_code_samples")
    return out


def ca_code_table():
    t = np.zeros((32, 1023), np.int8)
    for r in range(32):
        check(lib().gm_ca_code_row(r, _p(t[r])), "gm_ca_code_row")
    return t


class DopplerShiftTable:
    """doppler_shift.rs:5-22: pub doppler_freq_hz (= IF + Doppler), pub table"""

    def __init__(self, f_if, doppler_freq_hz, fs, num_samples):
        self.table = np.zeros(num_samples, np.complex64)
        f = C.c_float(0)
        check(lib().gm_doppler_table_new(f_if, doppler_freq_hz, fs, num_samples, C.byref(f), _p(self.table)),
              "DopplerShiftTable::new")
        self.doppler_freq_hz = f.value


def apply_doppler_shift(samples, doppler_table, output):
    """doppler_shift.rs:25-40 (GPU)"""
    s = np.ascontiguousarray(samples, np.complex64)
    t = doppler_table.table if isinstance(doppler_table, DopplerShiftTable) else np.ascontiguousarray(doppler_table, np.complex64)
    assert output.dtype == np.complex64 and output.flags.c_contiguous and t.size >= s.size
    _lib.init(_lib._initialised or 0)
    check(lib().gm_apply_doppler_shift(_p(s), _p(t), _p(output), s.size), "apply_doppler_shift")
    return output


def doppler_grid(span_hz=FREQ_SEARCH_ACQUISITION_HZ, step_hz=FREQ_SEARCH_STEP_HZ):
    """run()'s bin list: -span/2 + i*step, i = 0..span/step (do_acquisition.rs:248,253-255)"""
    capacity = int(span_hz) // int(step_hz) + 1
    return np.array([np.float32(-span_hz / 2.0) + np.float32(i) * np.float32(step_hz) for i in range(capacity)], np.float32)


def b1i_codes(prns=range(1, 38), n_chips=2046):
    """BeiDou B1I ranging codes (BDS-SIS-ICD-B1I 11-stage Gold codes) as an int8 [len(prns)][n_chips] table of +-1."""
    prns = list(prns)
    t = np.zeros((len(prns), n_chips), np.int8)
    for i, p in enumerate(prns):
        check(lib().gm_b1i_code(int(p), _p(t[i]), n_chips), "gm_b1i_code")
    return t


DECIDE_REFERENCE, DECIDE_BEST_BIN = 0, 1   # gm_decision_mode


class AcquisitionEngine:
    """The batched replacement of `workers.par_iter_mut()` (do_acquisition.rs:268-271, 302-313):
    all AcquisitionWorkers of one stage in one handle."""

    def __init__(self, fs, f_if, fft_size, doppler_hz=None, prn_ids=None, n_integrations=LONG_SAMPLES_LENGTH,
                 tables=None, codes=None, code_rate=1.023e6, threshold=7.0, decision_mode=0, strict_sum_order=False, reference_products=False,
                 device=None):
        _lib.init(device if device is not None else (_lib._initialised or 0))
        self.fs, self.f_if, self.fft_size, self.M = float(fs), float(f_if), int(fft_size), int(n_integrations)
        self.prn_ids = np.ascontiguousarray(prn_ids if prn_ids is not None else np.arange(1, 33), np.uint8)
        cfg = AcqCfg()
        cfg.fs, cfg.f_if, cfg.fft_size, cfg.n_integrations = self.fs, self.f_if, self.fft_size, self.M
        keep = []
        if tables is not None:   # &[DopplerShiftTable] built by the caller
            tb = np.ascontiguousarray(np.stack([t.table for t in tables]), np.complex64)
            tf = np.array([t.doppler_freq_hz for t in tables], np.float32)
            cfg.tables, cfg.table_freq, cfg.n_bins = tb.ctypes.data, tf.ctypes.data, len(tables)
            keep += [tb, tf]
        else:
            dh = np.ascontiguousarray(doppler_hz if doppler_hz is not None else doppler_grid(), np.float32)
            cfg.doppler_hz, cfg.n_bins = dh.ctypes.data, dh.size
            keep.append(dh)
        self.D = int(cfg.n_bins)
        cfg.n_prn, cfg.prn_ids = self.prn_ids.size, self.prn_ids.ctypes.data
        if codes is not None:
            cd = np.ascontiguousarray(codes, np.int8)
            assert cd.ndim == 2 and cd.shape[0] == self.prn_ids.size
            cfg.codes, cfg.code_len, cfg.code_rate = cd.ctypes.data, cd.shape[1], code_rate
            keep.append(cd)
        cfg.threshold = threshold
        cfg.decision_mode = decision_mode
        cfg.strict_sum_order = int(bool(strict_sum_order))
        cfg.reference_products = int(bool(reference_products))
        self.P = int(cfg.n_prn)
        h = C.c_void_p()
        check(lib().gm_acq_create(C.byref(cfg), C.byref(h)), "gm_acq_create")
        self._h = h
        self.table_freq = np.zeros(self.D, np.float32)
        check(lib().gm_acq_tables(self._h, None, _p(self.table_freq)), "gm_acq_tables")

    def close(self):
        if getattr(self, "_h", None):
            lib().gm_acq_destroy(self._h)
            self._h = None

    def __del__(self):      # (at interpreter shutdown the module globals close() uses may be gone already)
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _fmt(samples):
        a = np.asarray(samples)
        if a.dtype == np.int8:
            if a.ndim == 2 and a.shape[1] == 2:
                return np.ascontiguousarray(a), FMT_I8_IQ, a.shape[0]
            return np.ascontiguousarray(a), FMT_I8_REAL, a.size
        a = np.ascontiguousarray(a, np.complex64)
        return a, FMT_C32, a.size

    def search(self, samples_chunk, local_tail=0, prn_mask=0xFFFFFFFFFFFFFFFF):
        """-> list (one per worker) of AcquisitionResult dict or None"""
        a, fmt, n = self._fmt(samples_chunk)
        res = (AcqResult * self.P)()
        found = np.zeros(self.P, np.uint8)
        check(lib().gm_acq_search(self._h, _p(a), n, fmt, int(local_tail), int(prn_mask) & (2**64 - 1),
                                  C.cast(res, C.c_void_p), _p(found)), "gm_acq_search")
        return [res[i].as_dict() if found[i] else None for i in range(self.P)]

    def search_ring(self, ring, prn_mask=0xFFFFFFFFFFFFFFFF):
        """run()'s snapshot + fan-out against the device ring (do_acquisition.rs:297-313): -> (results, local_tail),
        or (None, None) while the ring holds fewer than M*N samples (:299)."""
        res = (AcqResult * self.P)()
        found = np.zeros(self.P, np.uint8)
        tail = C.c_uint64(0)
        st = lib().gm_acq_search_ring(self._h, ring._h, int(prn_mask) & (2**64 - 1), C.cast(res, C.c_void_p), _p(found),
                                      C.byref(tail))
        if st == -5:
            return None, None
        check(st, "gm_acq_search_ring")
        return [res[i].as_dict() if found[i] else None for i in range(self.P)], tail.value

    def finer_doppler(self, results):
        """Fine-Doppler refinement (finer_doppler, acquisition_bk.rs:215-302) of the found results of the LAST search,
        on that search's snapshot.  -> list (per worker) of dict(freq_hz, peak_index, peak_mag, fft_size) or None."""
        n = len(results)
        res = (AcqResult * n)()
        found = np.zeros(n, np.uint8)
        for i, r in enumerate(results):
            if r:
                found[i] = 1
                for k, _ in AcqResult._fields_:
                    setattr(res[i], k, r[k])
        f = np.zeros(n, np.float32)
        idx = np.zeros(n, np.uint64)
        mag = np.zeros(n, np.float32)
        size = C.c_uint64(0)
        check(lib().gm_acq_finer_doppler(self._h, C.cast(res, C.c_void_p), _p(found), n, _p(f), _p(idx), _p(mag),
                                         C.byref(size)), "gm_acq_finer_doppler")
        return [dict(freq_hz=float(f[i]), peak_index=int(idx[i]), peak_mag=float(mag[i]), fft_size=size.value)
                if found[i] else None for i in range(n)]

    def metrics(self):
        mx = np.zeros((self.P, self.D), np.float32)
        am = np.zeros((self.P, self.D), np.uint32)
        sm = np.zeros((self.P, self.D), np.float32)
        check(lib().gm_acq_metrics(self._h, _p(mx), _p(am), _p(sm)), "gm_acq_metrics")
        return mx, am, sm

    def code_fft(self, worker):
        out = np.zeros(self.fft_size, np.complex64)
        check(lib().gm_acq_code_fft(self._h, worker, _p(out)), "gm_acq_code_fft")
        return out

    def tables(self):
        t = np.zeros((self.D, self.fft_size), np.complex64)
        check(lib().gm_acq_tables(self._h, _p(t), None), "gm_acq_tables")
        return t

    # ---- device-resident / asynchronous forms (bench, multi-GPU)
    def set_stream(self, stream_ptr):
        check(lib().gm_acq_set_stream(self._h, C.c_void_p(stream_ptr)), "gm_acq_set_stream")

    def set_prn_mask(self, mask):
        check(lib().gm_acq_set_prn_mask(self._h, int(mask) & (2**64 - 1)), "gm_acq_set_prn_mask")

    def search_dev(self, d_samples_ptr, fmt, d_metrics_ptr=None):
        check(lib().gm_acq_search_dev(self._h, C.c_void_p(d_samples_ptr), fmt,
                                      C.c_void_p(d_metrics_ptr) if d_metrics_ptr else None), "gm_acq_search_dev")

    def decide_dev(self, d_metrics_ptr=None, n_prn=None, prn_ids=None, local_tail=0):
        n = self.P if n_prn is None else int(n_prn)
        ids = None
        if prn_ids is not None:
            ids = np.ascontiguousarray(prn_ids, np.uint8)
        check(lib().gm_acq_decide_dev(self._h, C.c_void_p(d_metrics_ptr) if d_metrics_ptr else None, n,
                                      _p(ids) if ids is not None else None, int(local_tail)), "gm_acq_decide_dev")

    def fetch_results(self, n_prn=None):
        n = self.P if n_prn is None else int(n_prn)
        res = (AcqResult * n)()
        found = np.zeros(n, np.uint8)
        check(lib().gm_acq_fetch_results(self._h, n, C.cast(res, C.c_void_p), _p(found)), "gm_acq_fetch_results")
        return [res[i].as_dict() if found[i] else None for i in range(n)]

    def synchronize(self):
        check(lib().gm_acq_synchronize(self._h), "gm_acq_synchronize")

    def prepare_dev(self, d_samples_ptr, fmt, ready_stream=None):
        """Stage F of the next dwell ahead of time, beside the current dwell's stage C (include/gnss_mi355x.h): a SNAPSHOT of the
        samples, named by the token this returns; search_prepared_dev(token) launches stage C on it.  ready_stream: the HIP stream
        whose queued work produces the samples (None: they are there already).  The samples must stay unchanged until the search
        that consumes the token has been synchronised."""
        tok = C.c_uint64(0)
        check(lib().gm_acq_prepare_dev(self._h, C.c_void_p(d_samples_ptr), fmt, C.c_void_p(ready_stream) if ready_stream else None,
                                       C.byref(tok)), "gm_acq_prepare_dev")
        return tok.value

    def search_prepared_dev(self, token, d_metrics_ptr=None):
        """Stage C on the spectra prepared under `token` (GmError INVALID_ARG when the token is stale, consumed, replaced or dropped)."""
        check(lib().gm_acq_search_prepared_dev(self._h, int(token), C.c_void_p(d_metrics_ptr) if d_metrics_ptr else None),
              "gm_acq_search_prepared_dev")

    def drop_prepared(self):
        check(lib().gm_acq_drop_prepared(self._h), "gm_acq_drop_prepared")

    def set_deferred_decision(self, on=True):
        """Back-to-back dwells: let decide_dev() ride with the next search_dev()'s first kernel (include/gnss_mi355x.h);
        synchronize() / fetch_results() run whatever is still pending."""
        check(lib().gm_acq_set_deferred_decision(self._h, int(on)), "gm_acq_set_deferred_decision")

    def enable_timing(self, on=True):
        check(lib().gm_acq_enable_timing(self._h, int(on)), "gm_acq_enable_timing")

    def last_timing(self):
        a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
        check(lib().gm_acq_last_timing(self._h, C.byref(a), C.byref(b), C.byref(c)), "gm_acq_last_timing")
        return {"mix_fft_ms": a.value, "corr_ms": b.value, "decide_ms": c.value}


    def timing_summary(self):
        n, a, b = C.c_uint32(0), C.c_float(0), C.c_float(0)
        check(lib().gm_acq_timing_summary(self._h, C.byref(n), C.byref(a), C.byref(b)), "gm_acq_timing_summary")
        return {"launches": n.value, "avg_mix_fft_ms": a.value, "avg_corr_ms": b.value}


class AcquisitionWorker:
    """AcquisitionWorker::new(prn, fft_size, freq_sampling_hz) + search_satellite(...)
    (do_acquisition.rs:130-226), one PRN per handle like the reference; the Doppler tables arrive
    with each call exactly as in the reference and are cached on the device by identity."""

    def __init__(self, prn, fft_size, freq_sampling_hz):
        self.prn, self.fft_size, self.freq_sampling_hz = int(prn), int(fft_size), float(freq_sampling_hz)
        if not 1 <= self.prn <= 32:
            raise IndexError("GPS_CA_CODE_32_PRN[prn - 1] out of bounds")
        self._eng = None
        self._key = None

    def search_satellite(self, samples_chunk, doppler_table, local_tail, num_integrations):
        key = (tuple(id(t) for t in doppler_table), int(num_integrations))
        if self._eng is None or self._key != key:
            if self._eng is not None:
                self._eng.close()
            self._eng = AcquisitionEngine(self.freq_sampling_hz, 0.0, self.fft_size, prn_ids=[self.prn],
                                          n_integrations=num_integrations, tables=doppler_table)
            self._key = key
        return self._eng.search(samples_chunk, local_tail)[0]

    @property
    def ca_code_samples_fft(self):
        if self._eng is None:
            tmp = AcquisitionEngine(self.freq_sampling_hz, 0.0, self.fft_size, doppler_hz=[0.0], prn_ids=[self.prn],
                                    n_integrations=1)
            out = tmp.code_fft(0)
            tmp.close()
            return out
        return self._eng.code_fft(0)


class SearchMode:
    ColdStart, WarmStart, SteadyState = 0, 1, 2


class AcquisitionManager:
    """do_acquisition.rs:39-74"""

    def __init__(self):
        self.mode = SearchMode.ColdStart

    def update_mode(self, trked_acount):
        self.mode = lib().gm_acq_manager_mode_for(int(trked_acount))

    def get_pacing_and_list(self, active_prns):
        am = 0
        for p in active_prns:
            am |= 1 << (p - 1)
        iv, m = C.c_uint64(0), C.c_uint32(0)
        check(lib().gm_acq_manager_pacing_and_list(self.mode, am, C.byref(iv), C.byref(m)), "get_pacing_and_list")
        return iv.value, m.value
