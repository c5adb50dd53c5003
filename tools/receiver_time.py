"""Diagnostic: bench.py's `receiver` leg alone (the whole chain of SURVEY §8 f1: feeder -> front-end -> ring -> acquisition + fine
Doppler -> tracking -> nav bits); RX_MS = milliseconds of signal (default 2200), RX_CPU=1 adds the CPU oracle chain."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from gnss_sdr_rs_amd import _lib, acquisition as A, tracking as T, synth
_lib.init(0)
synth.DEFAULT_GENERATOR = "xoshiro"
out = bench.receiver_leg(A.ca_code_table(), A, T, synth, os.environ.get("RX_CPU") == "1", n_ms=int(os.environ.get("RX_MS", "2200")))
print(json.dumps(out, indent=1))
