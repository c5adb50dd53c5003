"""Build the C-ABI shared library (hipcc, gfx950 only) in-tree: gnss-sdr-rs_amd/lib/libgnss_mi355x.so.

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the reference-faithful
products (carrier mix, x conj(code), per-sample tracking arithmetic) must round like rustc's
(no implicit FMA); FFT butterflies call __builtin_fmaf explicitly.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgnss_mi355x.so")
SOURCES = ["acq_kernels.hip", "acq_composite.hip", "trk_kernels.hip", "fe_kernels.hip", "nav_host.hip", "gm_api.hip"]
HEADERS = ["fft_core.h", "fft_plans.h", "acq_corr_plans.h", "gm_internal.h", "gm_libm.h", "acq_device.h", "acq_corr_ws31.h", "ws31_core.h", "acq_comp_ws.h", os.path.join("..", "..", "include", "gnss_mi355x.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC",
         "-Wall", "-Wno-unused-function"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """GM_EXTRA_FLAGS / GM_LIB_SUFFIX (diagnostics): extra hipcc flags and a suffix for the library / object directory, so that an
    A/B variant (e.g. -DGM_NO_HYBRID_PLANS) can sit beside the product library; load it with GM_LIB_PATH (see _lib.py)."""
    global LIB
    extra = os.environ.get("GM_EXTRA_FLAGS", "").split()
    suffix = os.environ.get("GM_LIB_SUFFIX", "")
    if suffix:
        LIB = os.path.join(LIBDIR, "libgnss_mi355x%s.so" % suffix)
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build" + suffix)
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src + ".o")
        if force or _stale(o, [s] + hdrs):
            jobs.append(["hipcc", *FLAGS, *extra, "-c", s, "-o", o])
    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout)
        return r.stdout
    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        # exported symbols: exactly the gm_* entry points of include/gnss_mi355x.h
        run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs,
             "-Wl,--version-script=" + os.path.join(CSRC, "exports.map")])
    build_synth(force)
    build_receiver_harness(force)
    return LIB


def build_receiver_harness(force=False):
    """host/receiver_harness.cpp: the receiver chain built from the C++ stage drivers of host/gnss_sdr.hpp behind an extern "C"
    surface (g++, host code only, links the product library through the C ABI) — what bench.py's `receiver` leg and the
    stage-driver tests call.  Linked against the plain product library even in a suffixed diagnostic build."""
    src = os.path.join(HERE, "host", "receiver_harness.cpp")
    lib = os.path.join(LIBDIR, "libgm_receiver.so")
    deps = [src, os.path.join(HERE, "host", "gnss_sdr.hpp"), os.path.join(HERE, "..", "include", "gnss_mi355x.h")]
    product = os.path.join(LIBDIR, "libgnss_mi355x.so")
    if force or _stale(lib, deps) or (os.path.exists(product) and os.path.getmtime(product) > os.path.getmtime(lib)):
        r = subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-o", lib, src,
                            "-L", LIBDIR, "-lgnss_mi355x", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", "-pthread"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode:
            raise RuntimeError("g++ failed:\n" + r.stdout)
    return lib


def build_synth(force=False):
    """The C++ scene generator of SURVEY §8 d2 (synthgen/synth_xoshiro.cpp, host code, g++): inputs for bench.py and tests.
    -fno-builtin: g++ otherwise merges cos(th) and sin(th) into one sincos() call, whose results differ from the separate
    functions' in the last bit now and then (1 of 4001 normals against the Python twin)."""
    src = os.path.join(HERE, "synthgen", "synth_xoshiro.cpp")
    lib = os.path.join(LIBDIR, "libgm_synth.so")
    if force or _stale(lib, [src]):
        r = subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-fno-builtin", "-o", lib, src],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode:
            raise RuntimeError("g++ failed:\n" + r.stdout)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
