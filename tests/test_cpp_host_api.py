"""The C++ host mirror (gnss-sdr-rs_amd/host/gnss_sdr.hpp) compiles against the C ABI and passes the
reference-shaped tests in tests/cpp/test_host_api.cpp (CPU: manager tests only; GPU: all)."""
import os
import subprocess

import pytest

from conftest import ROOT


def _build(gm):
    exe = os.path.join(ROOT, "gnss-sdr-rs_amd", "build", "test_host_api")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    libdir = os.path.dirname(gm.library_path())
    cmd = ["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I", os.path.join(ROOT, "gnss-sdr-rs_amd", "host"),
           os.path.join(ROOT, "tests", "cpp", "test_host_api.cpp"), "-o", exe, "-L", libdir, "-lgnss_mi355x",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-pthread"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


def test_cpp_host_api_builds_and_runs_cpu_part(gm):
    exe = _build(gm)
    r = subprocess.run([exe, "--cpu-only"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "test_acquisition_manager ok" in r.stdout and "test_ticket_loop_planning ok" in r.stdout, r.stdout


@pytest.mark.gpu
def test_cpp_host_api_gpu(gm):
    exe = _build(gm)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout
    for name in ("test_multicast_ring_buffer", "test_pll_frequency_pull_in", "test_acquisition_with_synthetic_data",
                 "test_receiver_threads", "test_frontend_refinement_navsync"):
        assert name + " ok" in r.stdout, r.stdout
