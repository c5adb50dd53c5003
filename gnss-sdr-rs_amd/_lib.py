"""ctypes binding of include/gnss_mi355x.h.  No fallback: a missing library is an ImportError-class failure."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("GM_LIB_PATH") or os.path.join(_HERE, "lib", "libgnss_mi355x.so")   # GM_LIB_PATH: an A/B build (build.py GM_LIB_SUFFIX), diagnostics only


class GmError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__(f"{where}: gm_status {status} ({detail})")


class c32(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


class AcqResult(C.Structure):
    _fields_ = [("prn", C.c_uint8), ("code_phase_samples", C.c_uint64), ("code_phase_chips", C.c_float),
                ("carrier_freq", C.c_float), ("fs", C.c_float), ("mag_relative", C.c_float),
                ("sample_global_index", C.c_uint64), ("doppler_bin", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class AcqCfg(C.Structure):
    _fields_ = [("fs", C.c_float), ("f_if", C.c_float), ("fft_size", C.c_uint32), ("n_integrations", C.c_uint32),
                ("n_bins", C.c_uint32), ("doppler_hz", C.c_void_p), ("tables", C.c_void_p),
                ("table_freq", C.c_void_p), ("n_prn", C.c_uint32), ("prn_ids", C.c_void_p), ("codes", C.c_void_p),
                ("code_len", C.c_uint32), ("code_rate", C.c_float), ("threshold", C.c_float),
                ("decision_mode", C.c_int32), ("strict_sum_order", C.c_int32), ("reference_products", C.c_int32)]


class TrkState(C.Structure):
    _fields_ = [("prn", C.c_uint8), ("active", C.c_uint8), ("reserved", C.c_uint8 * 2), ("lost_counter", C.c_uint32),
                ("next_sample_index", C.c_uint64), ("num_samples_per_code", C.c_uint64),
                ("carrier_freq", C.c_float), ("carrier_phase", C.c_float), ("carrier_error", C.c_float),
                ("carrier_nco", C.c_float), ("code_phase", C.c_float), ("code_error", C.c_float),
                ("code_nco", C.c_float), ("code_rate", C.c_float), ("i_prompt", C.c_float), ("q_prompt", C.c_float)]


class TrkOut(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("ip", "qp", "ie", "qe", "il", "ql", "ive", "qve", "ivl", "qvl")]


class NavStatus(C.Structure):
    _fields_ = [("flag_bit_sync", C.c_uint8), ("flag_frame_sync", C.c_uint8), ("sync_sw", C.c_uint8), ("bit", C.c_int8),
                ("polarity", C.c_int8), ("frame_sync_ind", C.c_uint32), ("n_frame_bits", C.c_uint64), ("i_p", C.c_float),
                ("sf_cnt", C.c_uint64), ("sf_start_biti", C.c_uint64), ("tow_expected_ind", C.c_uint64)]


class TrkCfg(C.Structure):
    _fields_ = [("fs", C.c_float), ("n_channels", C.c_uint32), ("n_arms", C.c_uint32),
                ("early_late_space", C.c_float), ("very_early_late_space", C.c_float),
                ("code_index_mode", C.c_int32), ("boc11", C.c_int32), ("codes", C.c_void_p),
                ("n_codes", C.c_uint32), ("code_len", C.c_uint32), ("nominal_code_rate", C.c_float),
                ("pll_bw", C.c_float), ("pll_zeta", C.c_float), ("pll_gain", C.c_float), ("dll_bw", C.c_float),
                ("dll_zeta", C.c_float), ("dll_gain", C.c_float), ("pll_dt", C.c_float), ("dll_dt", C.c_float),
                ("lock_threshold", C.c_float), ("max_lost_epochs", C.c_uint32), ("strict_libm", C.c_int32), ("strict_sum_order", C.c_int32),
                ("share_device", C.c_int32)]


FMT_C32, FMT_I8_IQ, FMT_I8_REAL = 0, 1, 2
CODE_INDEX_FAITHFUL, CODE_INDEX_FIXED = 0, 1

# every symbol include/gnss_mi355x.h declares: (name, restype, argtypes)
_vp, _f, _u64, _u32, _sz, _i = C.c_void_p, C.c_float, C.c_uint64, C.c_uint32, C.c_size_t, C.c_int
SIGNATURES = {
    "gm_abi_version": (_i, []),
    "gm_init": (_i, [_i]),
    "gm_device_count": (_i, [C.POINTER(_i)]),
    "gm_last_error": (C.c_char_p, []),
    "gm_status_string": (C.c_char_p, [_i]),
    "gm_ca_code_row": (_i, [_i, _vp]),
    "gm_b1i_code": (_i, [C.c_uint32, _vp, C.c_uint32]),
    "gm_generate_ca_code_samples": (_i, [C.c_uint8, _f, _f, _vp, _sz, C.POINTER(_sz)]),
    "gm_doppler_table_new": (_i, [_f, _f, _f, _sz, C.POINTER(_f), _vp]),
    "gm_apply_doppler_shift": (_i, [_vp, _vp, _vp, _sz]),
    "gm_fft_c2c_f32": (_i, [_sz, _i, _vp, _sz]),
    "gm_fft_power_spectrum_f32": (_i, [_sz, _vp, _vp]),
    "gm_rfft_f32": (_i, [_sz, _vp, _vp]),
    "gm_fft_supported_sizes": (_i, [_vp, _i]),
    "gm_acq_create": (_i, [C.POINTER(AcqCfg), C.POINTER(_vp)]),
    "gm_acq_destroy": (_i, [_vp]),
    "gm_acq_search": (_i, [_vp, _vp, _sz, _i, _u64, _u64, _vp, _vp]),
    "gm_acq_search_c32": (_i, [_vp, _vp, _sz, _u64, _u64, _vp, _vp]),
    "gm_acq_search_i8": (_i, [_vp, _vp, _sz, _u64, _u64, _vp, _vp]),
    "gm_acq_search_ring": (_i, [_vp, _vp, _u64, _vp, _vp, C.POINTER(_u64)]),
    "gm_acq_search_dev": (_i, [_vp, _vp, _i, _vp]),
    "gm_acq_set_prn_mask": (_i, [_vp, _u64]),
    "gm_acq_decide_dev": (_i, [_vp, _vp, _u32, _vp, _u64]),
    "gm_acq_finer_doppler": (_i, [_vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "gm_comm_get_unique_id": (_i, [_vp]),
    "gm_comm_init": (_i, [_i, _i, _vp, _vp]),
    "gm_comm_destroy": (_i, [_vp]),
    "gm_comm_info": (_i, [_vp, _vp, _vp]),
    "gm_acq_allgather_metrics": (_i, [_vp, _vp, _vp, _vp]),
    "gm_acq_allgather_metrics_async": (_i, [_vp, _vp, _vp, _vp]),
    "gm_comm_wait": (_i, [_vp, _vp]),
    "gm_comm_allgather_words": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "gm_grid_assemble_dev": (_i, [_vp, _u32, _u32, _u32, _vp, _u32, _vp, _vp]),
    "gm_acq_decide_planes_dev": (_i, [_vp, _vp, _vp, _u32, _u32, _vp, _vp, _u32, _f, _f, _f, _i, _u64, _vp, _vp, _vp]),
    "gm_acq_fetch_results": (_i, [_vp, _u32, _vp, _vp]),
    "gm_acq_decide_host": (_i, [_vp, _vp, _vp, _vp, _u32, _u32, _vp, _u32, _f, _f, _f, _u64, _vp, _vp]),
    "gm_acq_synchronize": (_i, [_vp]),
    "gm_acq_set_deferred_decision": (_i, [_vp, _i]),
    "gm_acq_prepare_dev": (_i, [_vp, _vp, _i, _vp, C.POINTER(C.c_uint64)]),
    "gm_acq_search_prepared_dev": (_i, [_vp, C.c_uint64, _vp]),
    "gm_acq_drop_prepared": (_i, [_vp]),
    "gm_acq_set_stream": (_i, [_vp, _vp]),
    "gm_acq_metrics": (_i, [_vp, _vp, _vp, _vp]),
    "gm_acq_code_fft": (_i, [_vp, _u32, _vp]),
    "gm_acq_tables": (_i, [_vp, _vp, _vp]),
    "gm_acq_enable_timing": (_i, [_vp, _i]),
    "gm_acq_last_timing": (_i, [_vp, C.POINTER(_f), C.POINTER(_f), C.POINTER(_f)]),
    "gm_acq_timing_summary": (_i, [_vp, C.POINTER(_u32), C.POINTER(_f), C.POINTER(_f)]),
    "gm_acq_debug_stamps": (_i, [_vp, _vp]),
    "gm_acq_manager_mode_for": (_i, [_sz]),
    "gm_acq_manager_pacing_and_list": (_i, [_i, _u32, C.POINTER(_u64), C.POINTER(_u32)]),
    "gm_ring_create": (_i, [_sz, C.POINTER(_vp)]),
    "gm_ring_destroy": (_i, [_vp]),
    "gm_ring_write_samples": (_i, [_vp, _vp, _sz]),
    "gm_ring_get_head": (_i, [_vp, C.POINTER(_u64)]),
    "gm_ring_copy_to_slice": (_i, [_vp, _u64, _vp, _sz]),
    "gm_ring_write_samples_async": (_i, [_vp, _vp, _sz]),
    "gm_ring_flush": (_i, [_vp]),
    "gm_ring_get_enqueued_head": (_i, [_vp, C.POINTER(_u64)]),
    "gm_ring_wait_head": (_i, [_vp, _u64, _u32, _vp]),
    "gm_nav_sync_create": (_i, [_i, _vp]),
    "gm_nav_sync_destroy": (_i, [_vp]),
    "gm_nav_sync_update": (_i, [_vp, _f, _f, _u64, _u64, _vp]),
    "gm_nav_sync_update_many": (_i, [_vp, _f, _vp, _sz, _sz, _u64, _u64, _vp, _vp, _vp]),
    "gm_nav_sync_frame_bits": (_i, [_vp, _vp, _sz, _vp]),
    "gm_nav_sync_histogram": (_i, [_vp, _vp]),
    "gm_nav_parity_check": (_i, [_vp, _vp, _vp]),
    "gm_frontend_create": (_i, [_f, _f, _f, _vp]),
    "gm_frontend_destroy": (_i, [_vp]),
    "gm_frontend_lut": (_i, [_vp, _vp, _vp, _vp]),
    "gm_frontend_get_state": (_i, [_vp, _vp, _vp, _vp]),
    "gm_frontend_set_state": (_i, [_vp, _f, _vp, _vp]),
    "gm_frontend_process_block": (_i, [_vp, _vp, _sz]),
    "gm_frontend_process_dev": (_i, [_vp, _vp, _i, _vp, _sz, _vp]),
    "gm_frontend_process_dev_batch": (_i, [_vp, _u32, _vp, _i, _vp, _sz, _vp]),
    "gm_frontend_synchronize": (_i, [_vp]),
    "gm_frontend_write_ring": (_i, [_vp, _vp, _vp, _sz, _i]),
    "gm_frontend_debug_repairs": (_i, [_vp, C.POINTER(_u32)]),
    "gm_trk_create": (_i, [C.POINTER(TrkCfg), C.POINTER(_vp)]),
    "gm_trk_destroy": (_i, [_vp]),
    "gm_trk_start": (_i, [_vp, _u32, C.POINTER(AcqResult)]),
    "gm_trk_reset": (_i, [_vp, _u32]),
    "gm_trk_get_state": (_i, [_vp, _u32, C.POINTER(TrkState)]),
    "gm_trk_set_state": (_i, [_vp, _u32, C.POINTER(TrkState)]),
    "gm_trk_get_ca_chip": (_i, [_vp, _u32, _f, C.POINTER(_f)]),
    "gm_loop_filter_new": (_i, [_f, _f, _f, C.POINTER(_f), C.POINTER(_f)]),
    "gm_loop_filter_update": (_f, [_f, _f, _f, _f, _f]),
    "gm_trk_correlate": (_i, [_vp, _u32, _vp, _sz, C.POINTER(TrkOut)]),
    "gm_trk_do_work": (_i, [_vp, _u32, _vp, _sz, C.POINTER(TrkOut), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]),
    "gm_trk_update_all": (_i, [_vp, _vp, _u32, _vp, _vp, _vp, C.POINTER(_u32)]),
    "gm_trk_update_all_dev": (_i, [_vp, _vp, _u32]),
    "gm_trk_update_all_async": (_i, [_vp, _vp, _u32, C.POINTER(C.c_uint64)]),
    "gm_trk_collect": (_i, [_vp, C.c_uint64, _i, _vp, _vp, _vp, _vp, C.POINTER(_u32), C.POINTER(_i)]),
    "gm_trk_get_states": (_i, [_vp, _vp]),
    "gm_trk_set_states": (_i, [_vp, _vp, _vp]),
    "gm_trk_synchronize": (_i, [_vp]),
    "gm_trk_set_stream": (_i, [_vp, _vp]),
    "gm_trk_debug_stamps": (_i, [_vp, _u32, _vp]),
    "gm_trk_enable_timing": (_i, [_vp, _i]),
    "gm_trk_last_timing": (_i, [_vp, C.POINTER(_f), C.POINTER(_u32)]),
}

_lib = None


def library_path():
    return _LIB_PATH


def lib():
    """Load libgnss_mi355x.so (built in-tree by build.py).  Raises if it is missing: no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError(f"{_LIB_PATH} is missing: run `python __graft_entry__.py build` (hipcc, gfx950). "
                          "There is no CPU fallback for the HIP path.")
    L = C.CDLL(_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)          # AttributeError if the header and the library drift apart
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def check(status, where):
    if status != 0:
        L = lib()
        raise GmError(status, where, (L.gm_status_string(status) or b"").decode() + "; " +
                      (L.gm_last_error() or b"").decode())


_initialised = None


def init(device=0):
    """gm_init(device): one process per GPU."""
    global _initialised
    if _initialised != device:
        check(lib().gm_init(device), "gm_init")
        _initialised = device
