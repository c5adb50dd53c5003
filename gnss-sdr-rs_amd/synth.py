"""Deterministic synthetic IF scenes.  Two generators behind one interface: SURVEY §8 d2's C++ generator
(synthgen/synth_xoshiro.cpp: splitmix64-seeded xoshiro256**, Box-Muller, the signal model in C++ — generator="xoshiro", what
bench.py uses) and numpy Philox streams keyed by the same seed (the scenes the parity tests were written on; same image, same
numpy on both boxes).

Signal model (complex baseband at IF, or real when real_only):
  x[n] = sum_s A_s * c_s(floor(((n - k_s) mod N_code_samples) * code_rate / fs) mod L) * d_s
               * exp(+j 2 pi (f_if + f_s) n / fs + j phi_s)  +  w[n],   w ~ CN(0, 2 sigma^2)
  A_s = sigma * sqrt(2 * 10^(CN0/10) / fs)  (amplitude giving C/N0 against the complex noise density)
k_s = sample offset of the code start inside each code period = what the reference's argmax reports.
"""
import ctypes as C
import math
import os

import numpy as np

SEED_BASE = 0x6E5553445200
# Which generator the scene functions use when the caller does not say: "philox" (numpy's counter-based generator: the scenes
# the parity tests were written on, some of them bisected to a threshold) or "xoshiro" (SURVEY §8 d2's generator in C++,
# synthgen/synth_xoshiro.cpp: splitmix64-seeded xoshiro256**, Box-Muller normals, the whole signal model evaluated sample by
# sample with the C library — bytes that depend on the seed and libm alone; what bench.py feeds the kernels).
DEFAULT_GENERATOR = os.environ.get("GM_SYNTH_GENERATOR", "philox")
_M64 = (1 << 64) - 1


def _rng(config_id, stream=0, generator=None):
    g = generator or DEFAULT_GENERATOR
    if g == "xoshiro":
        return XoshiroRng(SEED_BASE + config_id, stream)
    if g == "xoshiro-twin":
        return XoshiroTwin(SEED_BASE + config_id, stream)
    return np.random.Generator(np.random.Philox(key=SEED_BASE + config_id, counter=[0, 0, 0, stream]))


class _GsSat(C.Structure):
    _fields_ = [("prn_row", C.c_int32), ("has_bits", C.c_int32), ("cn0_dbhz", C.c_double), ("doppler_hz", C.c_double),
                ("code_start", C.c_double), ("phase", C.c_double), ("data_bits", C.c_void_p), ("n_bits", C.c_int64),
                ("bit_edge_ms", C.c_int64)]


_gs = None


def _gslib():
    """gnss-sdr-rs_amd/lib/libgm_synth.so (build.py builds it with g++ next to the HIP library)."""
    global _gs
    if _gs is None:
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.path.join(here, "lib", "libgm_synth.so")
        if not os.path.exists(path):
            raise RuntimeError("libgm_synth.so is missing: run `python __graft_entry__.py build` (gnss-sdr-rs_amd/build.py)")
        L = C.CDLL(path)
        u64p = C.POINTER(C.c_uint64)
        L.gs_splitmix64_next.restype, L.gs_splitmix64_next.argtypes = C.c_uint64, [u64p]
        L.gs_xoshiro_next.restype, L.gs_xoshiro_next.argtypes = C.c_uint64, [u64p]
        L.gs_stream_seed.restype, L.gs_stream_seed.argtypes = None, [C.c_uint64, C.c_uint64, u64p]
        L.gs_uniform.restype, L.gs_uniform.argtypes = C.c_double, [u64p]
        L.gs_integer.restype, L.gs_integer.argtypes = C.c_int64, [u64p, C.c_int64, C.c_int64]
        L.gs_fill_normal.restype, L.gs_fill_normal.argtypes = None, [u64p, C.c_void_p, C.c_size_t]
        L.gs_make_scene.restype = C.c_int
        L.gs_make_scene.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_size_t, C.c_void_p, C.c_int32,
                                    C.c_double, C.c_uint64, C.c_uint64, C.c_int32, C.c_double, C.c_int32, C.c_int64, C.c_void_p,
                                    C.c_void_p]
        _gs = L
    return _gs


class XoshiroRng:
    """The C++ generator behind the three numpy.random.Generator methods the scene functions call."""

    def __init__(self, seed, stream=0):
        self.s = (C.c_uint64 * 4)()
        _gslib().gs_stream_seed(seed & _M64, stream, self.s)

    def uniform(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * _gslib().gs_uniform(self.s)

    def integers(self, lo, hi, size=None):
        if size is None:
            return int(_gslib().gs_integer(self.s, lo, hi))
        n = int(np.prod(size))
        return np.array([_gslib().gs_integer(self.s, lo, hi) for _ in range(n)], np.int64).reshape(size)

    def standard_normal(self, n):
        out = np.empty(int(n), np.float64)
        _gslib().gs_fill_normal(self.s, out.ctypes.data, out.size)
        return out


class XoshiroTwin:
    """The same streams in pure Python (integers and `math`, i.e. the same libm): the twin the C++ generator is checked
    against on small counts (tests/test_synth_generator.py).  Slow by design."""

    def __init__(self, seed, stream=0):
        sm = (seed + stream * 0x9E3779B97F4A7C15) & _M64
        self.s = []
        for _ in range(4):
            sm = (sm + 0x9E3779B97F4A7C15) & _M64
            z = sm
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
            self.s.append(z ^ (z >> 31))

    @staticmethod
    def _rotl(x, k):
        return ((x << k) | (x >> (64 - k))) & _M64

    def next(self):
        s = self.s
        result = (self._rotl((s[1] * 5) & _M64, 7) * 9) & _M64
        t = (s[1] << 17) & _M64
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]
        s[2] ^= t
        s[3] = self._rotl(s[3], 45)
        return result

    def uniform(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * ((self.next() >> 11) * 2.0 ** -53)

    def integers(self, lo, hi, size=None):
        def one():
            return lo + ((((self.next() >> 32) * (hi - lo)) & _M64) >> 32)
        if size is None:
            return one()
        return np.array([one() for _ in range(int(np.prod(size)))], np.int64).reshape(size)

    def standard_normal(self, n):
        out = np.empty(int(n), np.float64)
        for i in range(0, int(n), 2):
            u1 = ((self.next() >> 11) + 1) * 2.0 ** -53
            u2 = (self.next() >> 11) * 2.0 ** -53
            r, th = math.sqrt(-2.0 * math.log(u1)), 6.283185307179586476925286766559 * u2
            out[i] = r * math.cos(th)
            if i + 1 < n:
                out[i + 1] = r * math.sin(th)
        return out


def _make_scene_native(code_table, fs, f_if, n_samples, sats, sigma, config_id, real_only, code_rate, quantize, bit_flip_at):
    """make_scene evaluated by synthgen/synth_xoshiro.cpp (noise stream 0 of the config's seed)."""
    tab = np.ascontiguousarray(code_table, np.int8)
    arr = (_GsSat * max(len(sats), 1))()
    keep = []
    for i, s in enumerate(sats):
        a = arr[i]
        a.prn_row, a.cn0_dbhz, a.doppler_hz = int(s["prn_row"]), float(s["cn0_dbhz"]), float(s["doppler_hz"])
        a.code_start, a.phase = float(s["code_start"]), float(s.get("phase", 0.0))
        if s.get("data_bits") is not None:
            b = np.ascontiguousarray(s["data_bits"], np.float64)
            keep.append(b)
            a.has_bits, a.data_bits, a.n_bits, a.bit_edge_ms = 1, b.ctypes.data, b.size, int(s.get("bit_edge_ms", 0))
    re, im = np.empty(int(n_samples), np.float64), np.empty(int(n_samples), np.float64)
    rc = _gslib().gs_make_scene(tab.ctypes.data, tab.shape[0], tab.shape[1], float(fs), float(f_if), int(n_samples), C.addressof(arr),
                                len(sats), float(sigma), (SEED_BASE + config_id) & _M64, 0, int(bool(real_only)), float(code_rate),
                                int(bool(quantize)), -1 if bit_flip_at is None else int(bit_flip_at), re.ctypes.data, im.ctypes.data)
    if rc:
        raise ValueError("gs_make_scene: %d" % rc)
    return re + 1j * im


def make_scene(code_table, fs, f_if, n_samples, sats, sigma=16.0, config_id=0, real_only=False, code_rate=1.023e6,
               quantize=True, bit_flip_at=None, generator=None):
    """sats: list of dict(prn_row, cn0_dbhz, doppler_hz, code_start, phase=0.0).  Returns complex128 array
    (quantized to integers in [-127,127] when quantize).  generator: "philox" (numpy), "xoshiro" (the whole scene by the C++
    generator) or "xoshiro-twin" (its streams in pure Python under numpy's signal arithmetic: the cross-check)."""
    if (generator or DEFAULT_GENERATOR) == "xoshiro":
        return _make_scene_native(code_table, fs, f_if, n_samples, sats, sigma, config_id, real_only, code_rate, quantize, bit_flip_at)
    n = np.arange(n_samples, dtype=np.float64)
    L = code_table.shape[1]
    rng = _rng(config_id, 0, generator)
    if real_only:
        x = sigma * rng.standard_normal(n_samples)
    elif (generator or DEFAULT_GENERATOR) == "xoshiro-twin":     # the C++ generator draws one Box-Muller pair per complex sample
        w = rng.standard_normal(2 * n_samples)
        x = sigma * (w[0::2] + 1j * w[1::2])
    else:
        x = sigma * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))
    x = x.astype(np.complex128)
    for s in sats:
        amp = sigma * np.sqrt(2.0 * 10.0 ** (s["cn0_dbhz"] / 10.0) / fs)
        # code Doppler ignored over the short scenes used here; chips from the sample offset
        chip = np.floor((n - s["code_start"]) * code_rate / fs).astype(np.int64) % L
        c = code_table[s["prn_row"]][chip].astype(np.float64)
        if bit_flip_at is not None:
            c = np.where(n >= bit_flip_at, -c, c)
        if s.get("data_bits") is not None:   # 50 bit/s navigation data: bit k covers code periods edge + 20k .. edge + 20k + 19
            period = np.floor((n - s["code_start"]) * (code_rate / L) / fs).astype(np.int64)
            k = (period - int(s.get("bit_edge_ms", 0))) // 20
            c = c * np.asarray(s["data_bits"], np.float64)[k % len(s["data_bits"])]
        ph = 2.0 * np.pi * (f_if + s["doppler_hz"]) * n / fs + s.get("phase", 0.0)
        if real_only:
            x += amp * np.sqrt(2.0) * c * np.cos(ph)
        else:
            x += amp * c * np.exp(1j * ph)
    if quantize:
        x = np.clip(np.round(x.real), -127, 127) + 1j * np.clip(np.round(x.imag), -127, 127)
    return x


def to_i8_iq(x):
    out = np.empty((x.size, 2), np.int8)
    out[:, 0] = x.real.astype(np.int8)
    out[:, 1] = x.imag.astype(np.int8)
    return out


def to_i8_real(x):
    return x.real.astype(np.int8)


def to_c32(x):
    return x.astype(np.complex64)


# ---- the named scenes of BASELINE.json's configs -----------------------------------------------
VISIBLE_CFG2 = [(2, 50.0), (3, 47.0), (6, 45.0), (9, 44.0), (11, 42.0), (14, 41.0), (18, 40.0), (19, 38.0)]


def cfg2_scene(code_table, n_ms=10, config_id=2, bit_flip_at=None, generator=None):
    """8 Msps complex int8, IF 0, N = 8000, +-5 kHz / 250 Hz grid (41 bins), 8 visible PRNs, Doppler
    off-bin-centre by up to +-0.4 bin, integer code starts."""
    fs, f_if, N = 8.0e6, 0.0, 8000
    rng = _rng(config_id, 1, generator)
    sats = []
    for prn, cn0 in VISIBLE_CFG2:
        b = int(rng.integers(2, 39))
        off = float(rng.uniform(-0.4, 0.4))
        sats.append(dict(prn=prn, prn_row=prn - 1, cn0_dbhz=cn0, doppler_hz=-5000.0 + 250.0 * (b + off),
                         code_start=int(rng.integers(0, N)), phase=float(rng.uniform(0, 2 * np.pi))))
    x = make_scene(code_table, fs, f_if, n_ms * N, sats, config_id=config_id, bit_flip_at=bit_flip_at, generator=generator)
    doppler_hz = np.array([-5000.0 + 250.0 * i for i in range(41)], np.float32)
    return dict(fs=fs, f_if=f_if, N=N, M=n_ms, doppler_hz=doppler_hz, sats=sats, x=x)


def cfg1_scene(code_table, capture_cfg, n_ms=10, config_id=1, generator=None):
    """Stand-in for the missing gioveAandB_short.bin (src/test_data/GPS_recordings/config.txt:1-19):
    real int8, fs 16.3676 MHz, IF 4.1304 MHz, N = 16368, PRNs / carriers / code phases from config.txt."""
    fs, f_if, N = capture_cfg["fs_hz"], capture_cfg["if_hz"], capture_cfg["fft_size"]
    cn0 = [50.0, 48.0, 47.0, 46.0, 45.0, 44.0, 43.0, 43.0, 40.0, 39.0]
    sats = []
    for row, c in zip(capture_cfg["signals"], cn0):
        sats.append(dict(prn=row["prn"], prn_row=row["prn"] - 1, cn0_dbhz=c,
                         doppler_hz=row["carrier_mhz"] * 1e6 - f_if, code_start=row["code_phase_samples"], phase=0.3 * row["prn"]))
    x = make_scene(code_table, fs, f_if, n_ms * N, sats, config_id=config_id, real_only=True, generator=generator)
    return dict(fs=fs, f_if=f_if, N=N, M=n_ms, doppler_hz=np.array(capture_cfg["doppler_hz"], np.float32), sats=sats, x=x)


def tracking_scene(code_table, fs, f_if, prns, n_ms, config_id=3, cn0=47.0, sigma=16.0, quantize=True, code_rows=None,
                   generator=None):
    """Continuous stream holding `prns` for n_ms ms; returns the stream and per-PRN truth."""
    N = int(round(fs / 1000.0))
    rng = _rng(config_id, 2, generator)
    sats = []
    for i, prn in enumerate(prns):
        sats.append(dict(prn=prn, prn_row=(code_rows[i] if code_rows is not None else prn - 1), cn0_dbhz=cn0,
                         doppler_hz=float(rng.uniform(-3000, 3000)), code_start=int(rng.integers(0, N)),
                         phase=float(rng.uniform(0, 2 * np.pi))))
    x = make_scene(code_table, fs, f_if, n_ms * N, sats, sigma=sigma, config_id=config_id, quantize=quantize, generator=generator)
    return dict(fs=fs, f_if=f_if, N=N, sats=sats, x=x)


def cfg4_grid_scene(ca_table, b1i_codes, config_id=4, generator=None):
    """BASELINE configs[3]: ONE 10 ms snapshot at 8 Msps complex int8 holding two satellites of each of the grid's three
    families — 32 GPS L1 C/A codes (N = 8000, 10 x 1 ms), 36 codes of Galileo-E1 GEOMETRY (4092 chips at 1.023 Mcps, N = 32000,
    2 x 4 ms; stand-in random codes: the ICD's memory codes are hex tables that cannot be derived offline) and 22 BeiDou B1I
    codes (2046 chips at 2.046 Mcps, N = 8000; `b1i_codes` = the ICD generator's output, gm_b1i_code).  Same bytes on every
    rank (fixed Philox keys).  -> dict(x complex128 quantised, e1 codes, truth {family: {prn: code start}}, fs, D, doppler_hz)."""
    fs, D, n = 8.0e6, 41, 80000
    dop = np.array([-5000.0 + 250.0 * i for i in range(D)], np.float32)
    e1 = np.where(_rng(config_id, 7, generator).integers(0, 2, (36, 4092)) > 0, 1, -1).astype(np.int8)
    x = make_scene(ca_table, fs, 0.0, n, [dict(prn_row=4, cn0_dbhz=50.0, doppler_hz=1130.0, code_start=4321),
                                          dict(prn_row=20, cn0_dbhz=47.0, doppler_hz=-2210.0, code_start=77)],
                   config_id=config_id, quantize=False, generator=generator)

    def clean(codes, sats, rate):      # the other families are added noise-free on top of the first scene's noise
        t = np.arange(n, dtype=np.float64)
        y = np.zeros(n, np.complex128)
        for s_ in sats:
            amp = 16.0 * np.sqrt(2.0 * 10.0 ** (s_["cn0_dbhz"] / 10.0) / fs)
            chip = np.floor((t - s_["code_start"]) * rate / fs).astype(np.int64) % codes.shape[1]
            y += amp * codes[s_["prn_row"]][chip] * np.exp(2j * np.pi * s_["doppler_hz"] * t / fs)
        return y
    x = x + clean(e1, [dict(prn_row=6, cn0_dbhz=49.0, doppler_hz=620.0, code_start=20001),
                       dict(prn_row=30, cn0_dbhz=47.0, doppler_hz=-3300.0, code_start=555)], 1.023e6)
    x = x + clean(b1i_codes, [dict(prn_row=2, cn0_dbhz=50.0, doppler_hz=-870.0, code_start=3000),
                              dict(prn_row=14, cn0_dbhz=48.0, doppler_hz=2950.0, code_start=6100)], 2.046e6)
    xq = np.clip(np.round(x.real), -127, 127) + 1j * np.clip(np.round(x.imag), -127, 127)
    truth = {"gps": {5: 4321, 21: 77}, "e1": {7: 20001, 31: 555}, "b1i": {3: 3000, 15: 6100}}
    truth_doppler = {"gps": {5: 1130.0, 21: -2210.0}, "e1": {7: 620.0, 31: -3300.0}, "b1i": {3: -870.0, 15: 2950.0}}
    return dict(fs=fs, D=D, doppler_hz=dop, x=xq, e1=e1, truth=truth, truth_doppler=truth_doppler)
