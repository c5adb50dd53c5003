"""Restates the reference's synthetic tracking fixtures as data (run here, in the build container; commits
tests/golden/tracking_synthetic.npz).

  generate_synthetic_signal (src/tracking/do_tracking.rs:434-462): one ms of "GPS L1" signal, noise-free:
      samples_per_ms   = (f_sampling / 1000.0) as usize
      code_phase_step  = 1.023e6 / f_sampling                                    (f32)
      carrier_phase_i  = starting_carrier_phase + (2.0 * PI * doppler / f_sampling * i as f32)     (:445-446)
      code_phase_i     = starting_code_phase + (code_phase_step * i as f32)                        (:449)
      chip_idx         = (code_phase_i.floor() as usize) % 1023                                    (:450)
      code_val         = ca_code[chip_idx] as f32                                                  (:451)
      sample_i         = (code_val * carrier_phase_i.cos(), code_val * carrier_phase_i.sin())      (:454-457)
  where the tests pass `ca_code = generate_ca_code_samples(prn, 1.023e6, f_sampling)` (:468-469, :576-577): the
  RESAMPLED code (4096 entries) indexed by a CHIP index, i.e. not a physical C/A signal — restated as
  written, because these are the reference's own known-answer inputs for the tracking channel.

f32 arithmetic is numpy float32 scalar arithmetic in the reference's operation order; cos/sin are the host libm's
cosf/sinf (what Rust's f32::cos/sin call on linux-gnu), reached through ctypes so that numpy's own SIMD kernels play no
part.  The chip table comes from the oracle's regenerated GPS_CA_CODE_32_PRN (pinned by test_oracle_golden.py).

Also stored: what the oracle (FAITHFUL mode = the reference's arithmetic as written, with update()'s buffer sized: SURVEY
§4) gives for the three update() calls of test_pll_frequency_pull_in (:464-570) and test_dll_code_phase_tracking
(:572-655) — regression vectors of the restatement, not outputs of the reference (rustc is not available here).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_libm = C.CDLL("libm.so.6")
_libm.cosf.argtypes = _libm.sinf.argtypes = [C.c_float]
_libm.cosf.restype = _libm.sinf.restype = C.c_float
F = np.float32
PI = F(3.14159265358979323846)       # std::f32::consts::PI


def generate_synthetic_signal(ca_code, doppler, starting_carrier_phase, starting_code_phase, f_sampling):
    f_sampling, doppler = F(f_sampling), F(doppler)
    samples_per_ms = int(f_sampling / F(1000.0))
    step = F(1.023e6) / f_sampling
    w = F(2.0) * PI * doppler / f_sampling                       # ((2.0 * PI) * doppler) / f_sampling, left to right
    out = np.zeros(samples_per_ms, np.complex64)
    for i in range(samples_per_ms):
        carrier_phase = F(starting_carrier_phase) + w * F(i)
        code_phase = F(starting_code_phase) + step * F(i)
        fl = np.floor(code_phase)
        chip_idx = (int(fl) if fl > 0 else 0) % 1023              # `as usize` saturates
        code_val = F(ca_code[chip_idx])
        out[i] = complex(code_val * F(_libm.cosf(float(carrier_phase))), code_val * F(_libm.sinf(float(carrier_phase))))
    return out


def scenario(O, name):
    """The two reference tests' set-ups: (prn, fs, signal, ring size in samples, channel id, start carrier)."""
    fs = 4_096_000.0
    if name == "pll":      # test_pll_frequency_pull_in :464-570
        prn, dop, cph, ring_mult, ch_id, start_freq = 2, 3000.0, 0.0, 8, 0, 2950.0
    else:                  # test_dll_code_phase_tracking :572-655
        prn, dop, cph, ring_mult, ch_id, start_freq = 3, 0.0, 0.25, 2, 3, 0.0
    mock = O.generate_ca_code_samples(prn, 1.023e6, fs)
    sig = generate_synthetic_signal(mock, dop, 0.0, cph, fs)
    return dict(prn=prn, fs=fs, signal=sig, ring=ring_mult * sig.size, ch_id=ch_id, start_freq=start_freq)


STATE_WORDS = ("carrier_freq", "carrier_phase", "carrier_error", "carrier_nco", "code_phase", "code_error", "code_nco", "code_rate",
               "i_prompt", "q_prompt")


def replay_oracle(O, sc):
    """The reference test's sequence of writes and update() calls through the oracle; returns per-update records."""
    ring = O.MulticastRingBuffer(sc["ring"])
    ch = O.TrackingChannel(sc["ch_id"], sc["fs"], code_index_mode=O.CODE_INDEX_FAITHFUL)
    ch.start(dict(prn=sc["prn"], code_phase_samples=0, code_phase_chips=0.0, carrier_freq=sc["start_freq"], fs=sc["fs"],
                  mag_relative=10.0, sample_global_index=0))
    recs = []
    for writes in (1, 2, 1):
        for _ in range(writes):
            ring.write_samples(sc["signal"])
        rc, out, msg = ch.update(ring)
        assert rc == 1 and msg is None
        recs.append(dict(out=out.copy(), head=ring.get_head(), next_sample_index=int(ch.c.next_sample_index),
                         num_samples_per_code=int(ch.c.num_samples_per_code),
                         **{k: np.float32(getattr(ch.c, k)) for k in STATE_WORDS}))
    return recs


def main():
    from oracle import oracle as O
    blob = {}
    for name in ("pll", "dll"):
        sc = scenario(O, name)
        blob[name + "_signal"] = sc["signal"]
        recs = replay_oracle(O, sc)
        blob[name + "_out"] = np.stack([r["out"] for r in recs])
        blob[name + "_index"] = np.array([[r["head"], r["next_sample_index"], r["num_samples_per_code"]] for r in recs], np.uint64)
        blob[name + "_state"] = np.array([[r[k] for k in STATE_WORDS] for r in recs], np.float32)
    np.savez_compressed(os.path.join(HERE, "tracking_synthetic.npz"), **blob)
    for k, v in blob.items():
        print(k, v.shape, v.dtype)
    print(blob["pll_state"][:, :4], blob["pll_out"])


if __name__ == "__main__":
    main()
