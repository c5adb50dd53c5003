"""Deterministic synthetic IF scenes.  numpy Philox streams keyed by a fixed seed, so
the build container and the GPU box regenerate identical bytes (same image, same numpy).

Signal model (complex baseband at IF, or real when real_only):
  x[n] = sum_s A_s * c_s(floor(((n - k_s) mod N_code_samples) * code_rate / fs) mod L) * d_s
               * exp(+j 2 pi (f_if + f_s) n / fs + j phi_s)  +  w[n],   w ~ CN(0, 2 sigma^2)
  A_s = sigma * sqrt(2 * 10^(CN0/10) / fs)  (amplitude giving C/N0 against the complex noise density)
k_s = sample offset of the code start inside each code period = what the reference's argmax reports.
"""
import numpy as np

SEED_BASE = 0x6E5553445200


def _rng(config_id, stream=0):
    return np.random.Generator(np.random.Philox(key=SEED_BASE + config_id, counter=[0, 0, 0, stream]))


def make_scene(code_table, fs, f_if, n_samples, sats, sigma=16.0, config_id=0, real_only=False, code_rate=1.023e6,
               quantize=True, bit_flip_at=None):
    """sats: list of dict(prn_row, cn0_dbhz, doppler_hz, code_start, phase=0.0).  Returns complex128 array
    (quantized to integers in [-127,127] when quantize)."""
    n = np.arange(n_samples, dtype=np.float64)
    L = code_table.shape[1]
    rng = _rng(config_id)
    if real_only:
        x = sigma * rng.standard_normal(n_samples)
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


def cfg2_scene(code_table, n_ms=10, config_id=2, bit_flip_at=None):
    """8 Msps complex int8, IF 0, N = 8000, +-5 kHz / 250 Hz grid (41 bins), 8 visible PRNs, Doppler
    off-bin-centre by up to +-0.4 bin, integer code starts."""
    fs, f_if, N = 8.0e6, 0.0, 8000
    rng = _rng(config_id, 1)
    sats = []
    for prn, cn0 in VISIBLE_CFG2:
        b = int(rng.integers(2, 39))
        off = float(rng.uniform(-0.4, 0.4))
        sats.append(dict(prn=prn, prn_row=prn - 1, cn0_dbhz=cn0, doppler_hz=-5000.0 + 250.0 * (b + off),
                         code_start=int(rng.integers(0, N)), phase=float(rng.uniform(0, 2 * np.pi))))
    x = make_scene(code_table, fs, f_if, n_ms * N, sats, config_id=config_id, bit_flip_at=bit_flip_at)
    doppler_hz = np.array([-5000.0 + 250.0 * i for i in range(41)], np.float32)
    return dict(fs=fs, f_if=f_if, N=N, M=n_ms, doppler_hz=doppler_hz, sats=sats, x=x)


def cfg1_scene(code_table, capture_cfg, n_ms=10, config_id=1):
    """Stand-in for the missing gioveAandB_short.bin (src/test_data/GPS_recordings/config.txt:1-19):
    real int8, fs 16.3676 MHz, IF 4.1304 MHz, N = 16368, PRNs / carriers / code phases from config.txt."""
    fs, f_if, N = capture_cfg["fs_hz"], capture_cfg["if_hz"], capture_cfg["fft_size"]
    cn0 = [50.0, 48.0, 47.0, 46.0, 45.0, 44.0, 43.0, 43.0, 40.0, 39.0]
    sats = []
    for row, c in zip(capture_cfg["signals"], cn0):
        sats.append(dict(prn=row["prn"], prn_row=row["prn"] - 1, cn0_dbhz=c,
                         doppler_hz=row["carrier_mhz"] * 1e6 - f_if, code_start=row["code_phase_samples"], phase=0.3 * row["prn"]))
    x = make_scene(code_table, fs, f_if, n_ms * N, sats, config_id=config_id, real_only=True)
    return dict(fs=fs, f_if=f_if, N=N, M=n_ms, doppler_hz=np.array(capture_cfg["doppler_hz"], np.float32), sats=sats, x=x)


def tracking_scene(code_table, fs, f_if, prns, n_ms, config_id=3, cn0=47.0, sigma=16.0, quantize=True, code_rows=None):
    """Continuous stream holding `prns` for n_ms ms; returns the stream and per-PRN truth."""
    N = int(round(fs / 1000.0))
    rng = _rng(config_id, 2)
    sats = []
    for i, prn in enumerate(prns):
        sats.append(dict(prn=prn, prn_row=(code_rows[i] if code_rows is not None else prn - 1), cn0_dbhz=cn0,
                         doppler_hz=float(rng.uniform(-3000, 3000)), code_start=int(rng.integers(0, N)),
                         phase=float(rng.uniform(0, 2 * np.pi))))
    x = make_scene(code_table, fs, f_if, n_ms * N, sats, sigma=sigma, config_id=config_id, quantize=quantize)
    return dict(fs=fs, f_if=f_if, N=N, sats=sats, x=x)


def cfg4_grid_scene(ca_table, b1i_codes, config_id=4):
    """BASELINE configs[3]: ONE 10 ms snapshot at 8 Msps complex int8 holding two satellites of each of the grid's three
    families — 32 GPS L1 C/A codes (N = 8000, 10 x 1 ms), 36 codes of Galileo-E1 GEOMETRY (4092 chips at 1.023 Mcps, N = 32000,
    2 x 4 ms; stand-in random codes: the ICD's memory codes are hex tables that cannot be derived offline) and 22 BeiDou B1I
    codes (2046 chips at 2.046 Mcps, N = 8000; `b1i_codes` = the ICD generator's output, gm_b1i_code).  Same bytes on every
    rank (fixed Philox keys).  -> dict(x complex128 quantised, e1 codes, truth {family: {prn: code start}}, fs, D, doppler_hz)."""
    fs, D, n = 8.0e6, 41, 80000
    dop = np.array([-5000.0 + 250.0 * i for i in range(D)], np.float32)
    e1 = np.where(_rng(config_id, 7).integers(0, 2, (36, 4092)) > 0, 1, -1).astype(np.int8)
    x = make_scene(ca_table, fs, 0.0, n, [dict(prn_row=4, cn0_dbhz=50.0, doppler_hz=1130.0, code_start=4321),
                                          dict(prn_row=20, cn0_dbhz=47.0, doppler_hz=-2210.0, code_start=77)],
                   config_id=config_id, quantize=False)

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
