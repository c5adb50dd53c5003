#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference checkout.

Run ONLY in the build container (needs /root/reference, read as text/data — nothing is
imported or executed from it; the reference is Rust and cannot run here).  The GPU box
never runs this script: it only reads the committed fixtures.

Fixtures written (data only: inputs + expected outputs of the reference's own tests/constants):
  ca_code_known_answers.json
      - prn1_chips          : the 1023-chip PRN-1 vector asserted by the reference's
                              test_prn_code (src/bk/gps_ca_prn.rs:72-124)
      - table_sha256        : SHA-256 of GPS_CA_CODE_32_PRN as 32*1023 int8 bytes, row-major
                              (src/constants/gps_ca_constants.rs:1-1346)
      - table_bits_hex      : the same table packed 1 bit/chip (+1 -> 1), 128 B per row, hex
      - first10_octal       : first 10 chips of every row in octal (IS-GPS-200 Table 3-Ia column,
                              derived here from the table; PRN1 = 1440)
  manager_known_answers.json  : (interval, mask) pairs asserted at do_acquisition.rs:364-395
  ring_buffer_vectors.json    : the write/read sequence asserted at multicast_ring_buffer.rs:147-209
  loop_filter_constants.json  : tau1/tau2 implied by do_tracking.rs:16-28,60-64 (computed in f32 here)
  capture_config.json         : PRN / carrier / code-phase table of the (missing) IF capture,
                              src/test_data/GPS_recordings/config.txt:1-19, and the accepted PRN list
                              of do_acquisition.rs:438
"""
import hashlib
import json
import os
import re

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def parse_int_list(text):
    return [int(x) for x in re.findall(r"-?\d+", text)]


def main():
    # ---- PRN-1 known answer from the reference's own test
    src = open(f"{REF}/src/bk/gps_ca_prn.rs").read()
    m = re.search(r"fn test_prn_code\(\).*?vec!\[(.*?)\]", src, re.S)
    prn1 = parse_int_list(m.group(1))
    assert len(prn1) == 1023 and set(prn1) == {1, -1}

    # ---- the constant table (data) -> digest + packed bits
    tsrc = open(f"{REF}/src/constants/gps_ca_constants.rs").read()
    body = tsrc[tsrc.index("= [") + 2:]
    vals = parse_int_list(body)
    assert len(vals) == 32 * 1023, len(vals)
    table = np.array(vals, np.int8).reshape(32, 1023)
    assert set(np.unique(table)) == {-1, 1}
    assert (table[0] == np.array(prn1, np.int8)).all()
    bits = (table > 0).astype(np.uint8)
    packed = np.packbits(np.pad(bits, ((0, 0), (0, 1))), axis=1)  # 1024 bits -> 128 B per row
    first10 = []
    for r in range(32):
        v = 0
        for b in bits[r, :10]:
            v = (v << 1) | int(b)
        first10.append(oct(v)[2:])
    json.dump({
        "source": "src/bk/gps_ca_prn.rs:72-124; src/constants/gps_ca_constants.rs:1-1346",
        "prn1_chips": prn1,
        "table_sha256": hashlib.sha256(table.tobytes()).hexdigest(),
        "table_bits_hex": packed.tobytes().hex(),
        "first10_octal": first10,
    }, open(f"{OUT}/ca_code_known_answers.json", "w"))

    # ---- acquisition manager known answers (do_acquisition.rs:339-395)
    json.dump({
        "source": "src/acquisition/do_acquisition.rs:339-395",
        "initial_mode": "ColdStart",
        "mode_for_tracked": {"3": "WarmStart", "5": "SteadyState", "0": "ColdStart"},
        "cold_start": {"active": [], "interval_ms": 500, "mask": 0xFFFFFFFF},
        "warm_start": {"update_mode": 3, "active": [1, 2, 3], "interval_ms": 1000, "mask": 2040},
    }, open(f"{OUT}/manager_known_answers.json", "w"), indent=1)

    # ---- ring buffer vectors (multicast_ring_buffer.rs:147-209)
    json.dump({
        "source": "src/utilities/multicast_ring_buffer.rs:147-209",
        "buf_size": 1024,
        "steps": [
            {"write": [0, 500], "head": 500},
            {"write": [500, 1030], "head": 1030, "buffer_1020_1024": [1020, 1024], "buffer_0_6": [1024, 1030],
             "copy_to_slice": {"start": 1020, "n": 10, "expect": [1020, 1030]}},
            {"write": [1030, 1050], "head": 1050, "buffer_1020_1024": [1020, 1024], "buffer_0_6": [1024, 1030],
             "buffer_6_16": [1030, 1040]},
        ],
    }, open(f"{OUT}/ring_buffer_vectors.json", "w"), indent=1)

    # ---- loop filter constants (do_tracking.rs:16-28, 60-64), evaluated in float32
    f = np.float32

    def lf(bw, z, g):
        w = f(f(f(bw) * f(8.0)) * f(z)) / f(f(f(4.0) * f(f(z) * f(z))) + f(1.0))
        return float(f(g) / f(w * w)), float(f(f(2.0) * f(z)) / w)

    pll, dll = lf(25.0, 0.7, 0.25), lf(2.0, 0.7, 1.0)
    json.dump({"source": "src/tracking/do_tracking.rs:16-28,60-64",
               "pll": {"bw": 25.0, "zeta": 0.7, "gain": 0.25, "tau1": pll[0], "tau2": pll[1]},
               "dll": {"bw": 2.0, "zeta": 0.7, "gain": 1.0, "tau1": dll[0], "tau2": dll[1]},
               "lock_threshold": 15.0, "max_lost_epochs": 20, "num_channels": 15,
               "early_late_space": 0.5},
              open(f"{OUT}/loop_filter_constants.json", "w"), indent=1)

    # ---- capture documentation (config.txt) + the acquisition test's accepted list
    cfg = open(f"{REF}/src/test_data/GPS_recordings/config.txt").read()
    rows = []
    for line in cfg.splitlines():
        mm = re.match(r"\s*(\d+)(\[\d\])?\s+(\d+\.\d+)\s+(\d+)\s*$", line)
        if mm:
            rows.append({"prn": int(mm.group(1)), "carrier_mhz": float(mm.group(3)),
                         "code_phase_samples": int(mm.group(4)), "note": mm.group(2) or ""})
    assert len(rows) == 10, rows
    json.dump({"source": "src/test_data/GPS_recordings/config.txt:1-19; src/acquisition/do_acquisition.rs:400-438",
               "fs_hz": 16367600.0, "if_hz": 4130400.0, "sample_format": "int8 real", "fft_size": 16368,
               "doppler_hz": [-7000.0 + 500.0 * i for i in range(29)], "num_integrations": 10,
               "signals": rows, "test_accepts_prns": [3, 6, 9, 11, 14, 18, 19, 22, 28, 32]},
              open(f"{OUT}/capture_config.json", "w"), indent=1)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
