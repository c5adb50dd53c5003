"""CPU: the C-ABI library loads, exports every symbol the header declares, and its host-side (no-GPU) entry
points agree with the golden fixtures and the oracle.  No compute entry is called here."""
import ctypes as C
import hashlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden


def test_header_symbols_all_exported(gm):
    hdr = open(os.path.join(ROOT, "include", "gnss_mi355x.h")).read()
    declared = set(re.findall(r"\b(gm_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"gm_status", "gm_c32"}
    from gnss_sdr_rs_amd import _lib
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    nm = subprocess.run(["nm", "-D", "--defined-only", gm.library_path()], stdout=subprocess.PIPE, text=True).stdout
    exported = set(re.findall(r" T (gm_[a-z0-9_]+)", nm))
    assert declared <= exported, declared - exported
    assert not [s for s in re.findall(r" [TDB] (\S+)", nm) if not s.startswith("gm_")]   # nothing else leaks
    import __graft_entry__ as entry
    assert gm.lib().gm_abi_version() == entry.header_abi_version() >= 6


def test_graft_entry_build_runs_and_imports_the_package():
    """The driver's "does it build" call: `__graft_entry__.build()` in a process of its own (hipcc cross-compiles without a GPU;
    an up-to-date tree is an incremental no-op) must exit 0, having loaded the library and compared its ABI number with the
    header's (a stale literal made this the one entry point that failed while every test was green)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "__graft_entry__.py"), "build"], cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:]
    assert "built " in out.stdout and "libgnss_mi355x.so" in out.stdout, out.stdout[-1000:]


def test_library_links_no_oracle_and_no_torch(gm):
    out = subprocess.run(["ldd", gm.library_path()], stdout=subprocess.PIPE, text=True).stdout
    assert "liboracle" not in out and "torch" not in out and "libamdhip64" in out


def test_library_reads_no_environment_variable_by_name(gm):
    """a receiver that links the library must not get other kernels because its environment
    carries a GM_* name.  The library imports neither getenv nor secure_getenv; its diagnostic overrides sit behind ONE gate
    (gm::diag_int, gm_api.hip: a walk of `environ` for the literal GM_DIAGNOSTICS=1), and no kernel source calls getenv."""
    out = subprocess.run(["nm", "-D", "--undefined-only", gm.library_path()], stdout=subprocess.PIPE, text=True).stdout
    assert "getenv" not in out, [l for l in out.splitlines() if "getenv" in l]
    csrc = os.path.join(ROOT, "gnss-sdr-rs_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        text = open(os.path.join(csrc, f)).read()
        assert "getenv(" not in text, f
        assert "GM_LAB_" not in text, f           # item 7b: no timing-ablation branches inside the shipped kernels


def test_compute_entry_fails_loudly_without_device(gm):
    """No CPU fallback: on a box without a GPU the compute entries return GM_ERR_NO_DEVICE."""
    code = ("import sys; sys.path.insert(0, %r); import numpy as np, ctypes as C; import gnss_sdr_rs_amd as g; L = g.lib();"
            "n = C.c_int(0); L.gm_device_count(C.byref(n));"
            "x = np.zeros(1024, np.complex64);"
            "rc = L.gm_fft_c2c_f32(1024, 0, x.ctypes.data_as(C.c_void_p), 1);"
            "print(n.value, rc)") % ROOT
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    ndev, rc = out.stdout.split()[-2:]
    if int(ndev) == 0:
        assert int(rc) == -3, out.stdout + out.stderr


def test_null_handles_are_refused_not_dereferenced(gm):
    """Argument checks that need no device: every handle-taking entry of the back-to-back-dwell additions returns GM_ERR_INVALID_ARG
    (-1) on a null handle, and gm_last_error says why."""
    import ctypes as C
    L = gm.lib()
    assert L.gm_acq_set_deferred_decision(None, 1) == -1
    assert b"null handle" in L.gm_last_error()
    assert L.gm_acq_prepare_dev(None, C.c_void_p(16), 0, None, C.byref(C.c_uint64(0))) == -1
    assert L.gm_acq_search_prepared_dev(None, 1, None) == -1 and L.gm_acq_drop_prepared(None) == -1
    assert L.gm_acq_synchronize(None) == -1
    # the asynchronous tracking entries (ABI 6): null handle / ring / ticket pointer, ticket 0, null `ready`
    tok, ready = C.c_uint64(7), C.c_int(5)
    assert L.gm_trk_update_all_async(None, None, 4, C.byref(tok)) == -1 and tok.value == 7
    assert L.gm_trk_collect(None, 1, 1, None, None, None, None, None, C.byref(ready)) == -1 and ready.value == 5
    assert b"bad argument" in L.gm_last_error()
    # ABI 7: the bulk state entries, the enqueued head, the front-end's repair count
    st, h, runs = (C.c_uint8 * 64)(), C.c_uint64(9), C.c_uint32(9)
    assert L.gm_trk_get_states(None, C.cast(st, C.c_void_p)) == -1 and L.gm_trk_set_states(None, C.cast(st, C.c_void_p), None) == -1
    assert L.gm_ring_get_enqueued_head(None, C.byref(h)) == -1 and h.value == 9
    assert L.gm_frontend_debug_repairs(None, C.byref(runs)) == -1 and runs.value == 9


def test_ca_table_and_resampler_host(gm, oracle):
    from gnss_sdr_rs_amd import acquisition as A
    g = golden("ca_code_known_answers.json")
    t = A.ca_code_table()
    assert hashlib.sha256(t.tobytes()).hexdigest() == g["table_sha256"]
    assert (t[0] == np.array(g["prn1_chips"], np.int8)).all()
    for fs in (16_367_600.0, 8.0e6, 4_096_000.0):
        assert (A.generate_ca_code_samples(19, 1.023e6, fs) == oracle.generate_ca_code_samples(19, 1.023e6, fs)).all()
    from gnss_sdr_rs_amd import GmError
    with pytest.raises(GmError):
        A.generate_ca_code_samples(0, 1.023e6, 8e6)
    # bk/gps_ca_prn.rs:65-70 `should_panic`: generate_ca_code(40) — no such PRN: the reference panics (index out of bounds),
    # the boundary reports GM_ERR_OUT_OF_RANGE; and the table ends at row 31 (GPS_CA_CODE_32_PRN[32] is what FAITHFUL
    # tracking of PRN 32 would index, do_tracking.rs:276)
    with pytest.raises(GmError) as ei:
        A.generate_ca_code_samples(40, 1.023e6, 8e6)
    assert ei.value.status == -5
    import ctypes as C
    row = np.zeros(1023, np.int8)
    assert gm.lib().gm_ca_code_row(32, row.ctypes.data_as(C.c_void_p)) == -5 and gm.lib().gm_ca_code_row(31, row.ctypes.data_as(C.c_void_p)) == 0


def test_doppler_table_host_bit_exact_with_oracle(gm, oracle):
    from gnss_sdr_rs_amd import acquisition as A
    for f_if, dop, fs, n in ((4_130_400.0, -7000.0, 16_367_600.0, 16368), (0.0, 250.0, 8.0e6, 8000)):
        a, b = A.DopplerShiftTable(f_if, dop, fs, n), oracle.DopplerShiftTable(f_if, dop, fs, n)
        assert a.doppler_freq_hz == b.doppler_freq_hz
        assert (a.table.view(np.uint32) == b.table.view(np.uint32)).all()
    assert (A.doppler_grid() == np.array(golden("capture_config.json")["doppler_hz"], np.float32)).all()


def test_manager_and_loop_filter_host(gm):
    from gnss_sdr_rs_amd import acquisition as A, tracking as T
    g = golden("manager_known_answers.json")
    m = A.AcquisitionManager()
    assert m.mode == A.SearchMode.ColdStart
    assert m.get_pacing_and_list(set()) == (500, 0xFFFFFFFF)
    m.update_mode(3)
    assert m.mode == A.SearchMode.WarmStart and m.get_pacing_and_list({1, 2, 3}) == (1000, g["warm_start"]["mask"])
    m.update_mode(5)
    assert m.mode == A.SearchMode.SteadyState
    m.update_mode(0)
    assert m.mode == A.SearchMode.ColdStart
    lf = golden("loop_filter_constants.json")
    for k in ("pll", "dll"):
        f = T.LoopFilter(lf[k]["bw"], lf[k]["zeta"], lf[k]["gain"])
        assert f.tau1 == lf[k]["tau1"] and f.tau2 == lf[k]["tau2"]
    f = T.LoopFilter(25.0, 0.7, 0.25)
    d, e, dt = np.float32(0.01), np.float32(0.004), np.float32(0.001)
    exp = np.float32(d * np.float32(dt / np.float32(f.tau1))) + np.float32(np.float32(d - e) * np.float32(np.float32(f.tau2) / np.float32(f.tau1)))
    assert f.update(float(d), float(e), float(dt)) == float(np.float32(exp))


def test_decide_host_equals_oracle_decision(gm, oracle):
    from gnss_sdr_rs_amd import distributed as Dm
    rng = np.random.default_rng(4)
    P, D, N = 6, 9, 4096
    mx = (rng.random((P, D)) * 1e6 + 1e6).astype(np.float32)
    sm = (mx * rng.uniform(300, 900, (P, D))).astype(np.float32)     # ratio max/avg ~ 4.5 .. 13.6
    am = rng.integers(0, N, (P, D)).astype(np.uint32)
    tf = np.linspace(-2000, 2000, D).astype(np.float32)
    prns = [3, 4, 9, 12, 20, 31]
    got = Dm.decide_host(Dm.pack_metrics(mx, am, sm), prns, tf, N, 4.096e6, local_tail=77)
    n_found = 0
    for i, p in enumerate(prns):
        exp = oracle.decide_from_metrics(mx[i], am[i], sm[i], tf, N, p, 4.096e6, 77)
        assert got[i] == exp
        n_found += exp is not None
    assert 0 < n_found < P


def test_synth_scene_is_deterministic(gm):
    from gnss_sdr_rs_amd import acquisition as A, synth
    t = A.ca_code_table()
    a, b = synth.cfg2_scene(t), synth.cfg2_scene(t)
    assert (a["x"] == b["x"]).all() and a["sats"] == b["sats"]
    i8 = synth.to_i8_iq(a["x"])
    assert i8.shape == (80000, 2) and np.abs(i8).max() <= 127
    assert hashlib.sha256(i8.tobytes()).hexdigest() == hashlib.sha256(synth.to_i8_iq(b["x"]).tobytes()).hexdigest()


def test_fft_core_cpu_emulation_of_every_plan():
    """tests/cpu/test_fft_core.cpp: the in-LDS FFT header (csrc/fft_core.h) is host/device portable; g++ runs every
    shipped plan lane by lane, barrier phase by barrier phase, against a float64 DFT (threshold 7e-7 relative L2)."""
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(tempfile.mkdtemp(prefix="gm_fftcore_"), "test_fft_core")
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I", os.path.join(root, "gnss-sdr-rs_amd", "csrc"),
                    os.path.join(root, "tests", "cpu", "test_fft_core.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], stdout=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "Plan8000" in r.stdout and "Plan256" in r.stdout and "worst" in r.stdout
    assert "ws31<Plan16368>" in r.stdout          # the wave-specialised N = 16368 transform incl. its matrix-product radix-31 pass (csrc/ws31_core.h)
    assert "comp_ws assumptions on Hybrid16000: 0 violations" in r.stdout      # the stored row order and the one-wrap slot rule of csrc/acq_comp_ws.h


def test_device_atanf_restatement_matches_host_libm_bit_for_bit():
    """tests/cpu/test_libm.cpp: gm::atanf_glibc (csrc/gm_libm.h, what the tracking epilogue runs on the device for
    f32::atan of do_tracking.rs:280) equals this host's atanf on 8.7e7 arguments, bit for bit; gm::sincosf_glibc (what
    gm_trk_cfg.strict_libm runs for `phase.cos()` / `phase.sin()`, :234-235) equals this host's sinf AND cosf on 1.0e8
    arguments (every 127th bit pattern, the carrier's operating range, the range boundaries), bit for bit."""
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(tempfile.mkdtemp(prefix="gm_libm_"), "test_libm")
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I", os.path.join(root, "gnss-sdr-rs_amd", "csrc"),
                    os.path.join(root, "tests", "cpu", "test_libm.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], stdout=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:]
    assert "sincosf_glibc:" in r.stdout and "atanf_glibc:" in r.stdout
    for line in r.stdout.splitlines():
        if "mismatches" in line:
            assert " 0 mismatches" in line, line


def test_rust_binding_source_matches_the_abi(gm):
    """rust/src/*.rs (the reference-side binding shipped as source) cannot be compiled here, so it is
    checked structurally: every #[repr(C)] struct lists the same fields in the same order as the ctypes mirror the test
    suite exercises; every extern fn exists in include/gnss_mi355x.h; every gm_* call made by a wrapper file is declared
    in mi355x.rs; and every wrapper carries the reference's own signature (do_acquisition.rs:130-226,
    doppler_shift.rs:5-40, do_tracking.rs:88-382, fft.rs:5-56)."""
    from gnss_sdr_rs_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the layout under rust/src IS the destination layout in the crate: src/mi355x.rs + src/mi355x/{...}.rs
    rs = {n: open(os.path.join(root, "rust", "src", "mi355x.rs" if n == "mi355x.rs" else os.path.join("mi355x", n))).read()
          for n in ("mi355x.rs", "doppler_shift.rs", "do_acquisition.rs", "do_tracking.rs", "fft.rs")}
    src = rs["mi355x.rs"]
    hdr = open(os.path.join(root, "include", "gnss_mi355x.h")).read()

    def rust_fields(text, name):
        body = re.search(r"pub struct %s\s*\{([^}]*)\}" % name, text, re.S).group(1)
        body = re.sub(r"//[^\n]*", "", body)
        return re.findall(r"pub\s+(\w+)\s*:", body)

    pairs = {"GmAcqResult": _lib.AcqResult, "GmAcqCfg": _lib.AcqCfg, "GmTrkState": _lib.TrkState, "GmTrkOut": _lib.TrkOut,
             "GmTrkCfg": _lib.TrkCfg}
    for rname, ct in pairs.items():
        assert rust_fields(src, rname) == [f[0] for f in ct._fields_], rname
    fns = set(re.findall(r"pub fn (gm_\w+)\s*\(", src))
    assert len(fns) >= 30
    declared = set(re.findall(r"\b(gm_[a-z0-9_]+)\s*\(", hdr))
    assert fns <= declared, sorted(fns - declared)
    # every library call of a wrapper file is bound in mi355x.rs
    for name, text in rs.items():
        if name == "mi355x.rs":
            continue
        code = re.sub(r"//[^\n]*", "", text)
        used = set(re.findall(r"\b(gm_[a-z0-9_]+)\s*\(", code))
        assert used and used <= fns, (name, sorted(used - fns))

    def flat(t):
        return re.sub(r"\s+", " ", re.sub(r"//[^\n]*", "", t))
    sigs = {
        "doppler_shift.rs": ["pub struct DopplerShiftTable { pub doppler_freq_hz: f32, pub table: Vec<Complex32>, }",
                             "pub fn new(f_if: f32, doppler_freq_hz: f32, fs: f32, num_samples: usize) -> Self",
                             "pub fn apply_doppler_shift(samples: &[Complex32], doppler_table: &DopplerShiftTable, output: &mut [Complex32])"],
        "do_acquisition.rs": ["pub fn new(prn: u8, fft_size: usize, freq_sampling_hz: f32) -> Self",
                              "pub fn search_satellite( &mut self, samples_chunk: &[Complex32], doppler_table: &[DopplerShiftTable], "
                              "local_tail: usize, num_integrations: usize, ) -> Option<AcquisitionResult>"],
        "do_tracking.rs": ["pub fn new(id: u8, fs: f32) -> Self", "pub fn start(&mut self, result: AcquisitionResult)",
                           "pub fn is_active(&self) -> bool",
                           "pub fn update(&mut self, buff: Arc<MulticastRingBuffer>) -> Option<TrackingMessage>",
                           "pub fn early_late_correlation(&mut self) -> (f32, f32, f32, f32, f32, f32)",
                           "pub fn get_ca_chip(&self, phase: f32) -> f32",
                           "pub fn run_loop_filters(&mut self, i_p: f32, q_p: f32, i_e: f32, q_e: f32, i_l: f32, q_l: f32)",
                           "pub fn reset(&mut self)",
                           "pub fn new( acq_to_trk: Receiver<AcquisitionResult>, trk_to_acq: Sender<TrackingMessage>, fs: f32, ) -> Self",
                           "pub fn process_channels(&mut self, multi_ring_buf: Arc<MulticastRingBuffer>)"],
        "fft.rs": ["pub fn new(len: usize) -> Self", "pub fn execute(&self, input: &mut [Complex<T>]) -> Vec<Complex<T>>",
                   "pub fn power_spectrum(&self, input: &mut [Complex<T>]) -> Vec<T>",
                   "pub fn execute(&self, input: &mut [T]) -> Vec<Complex<T>>", "pub fn power_spectrum(&self, input: &mut [T]) -> Vec<T>"],
    }
    for name, want in sigs.items():
        f = flat(rs[name])
        for w in want:
            assert flat(w) in f, (name, w)
    # against the reference's own API surface (tests/golden/reference_api_signatures.json, extracted from the reference's
    # sources by tests/golden/make_api_signatures.py): a name both files import comes from the SAME path (e.g. the channel
    # ends are crossbeam_channel's, do_tracking.rs:8, not std::sync::mpsc's), and every pub fn that exists in the reference
    # under the same type has the reference's signature, parameter types included
    from rs_api import imports, signatures, strip_comments
    ref = golden("reference_api_signatures.json")
    n_sig = 0
    for name in ("doppler_shift.rs", "do_acquisition.rs", "do_tracking.rs", "fft.rs"):
        text = strip_comments(rs[name])
        mine, theirs = imports(text), ref[name]["imports"]
        for sym, path in mine.items():
            if sym != "*" and sym in theirs:
                if (name, sym, path) == ("fft.rs", "Complex", "num_complex"):
                    assert theirs[sym] == "rustfft::num_complex"      # rustfft's re-export of the same type
                    continue
                assert path == theirs[sym], (name, sym, path, theirs[sym])
        for owner, fns in signatures(text).items():
            for fn, sig in fns.items():
                want = ref[name]["pub_fn"].get(owner, {}).get(fn)
                if want is not None:
                    assert sig == want, (name, owner, fn, sig, want)
                    n_sig += 1
    assert n_sig >= 20, n_sig
    assert imports(strip_comments(rs["do_tracking.rs"]))["Receiver"] == "crossbeam_channel"
    # TrackingChannel keeps the reference's 22 pub fields, in order (do_tracking.rs:88-116)
    assert rust_fields(rs["do_tracking.rs"], "TrackingChannel") == [
        "id", "prn", "state", "lost_counter", "fs", "next_sample_index", "num_samples_per_code", "ca_code_samples", "data_samples",
        "carrier_freq", "carrier_phase", "carrier_error", "carrier_nco", "code_phase", "code_error", "code_nco", "code_rate",
        "cos_p", "sin_p", "i_prompt", "q_prompt", "pll_filter", "dll_filter"]
    assert rust_fields(rs["do_tracking.rs"], "TrackingManager") == ["channels", "acq_to_trk", "trk_to_acq"]
    # the stage drivers keep the reference's signatures too (do_acquisition.rs:241-247, do_tracking.rs:384-389)
    for name in ("do_acquisition.rs", "do_tracking.rs"):
        assert signatures(strip_comments(rs[name]))[""]["run"] == ref[name]["pub_fn"][""]["run"], name

    # the manager's passes go through the bulk state entries (a dirty test, no per-channel FFI call, no Vec
    # per call) and run() drives the ticket path (enqueue behind the mirror's copies, collect later)
    trk = strip_comments(rs["do_tracking.rs"])
    mgr = trk[trk.index("impl TrackingManager"):trk.index("impl Drop for TrackingManager")]
    assert "gm_trk_set_states(" in mgr and "gm_trk_get_states(" in mgr and "gm_trk_update_all_async(" in mgr and "gm_trk_collect(" in mgr
    assert not re.search(r"gm_trk_[gs]et_state\(", mgr) and "vec![" not in mgr[mgr.index("fn take_acquisitions"):]
    run_body = trk[trk.index("pub fn run("):]
    assert "process_channels_async(" in run_body and "collect_ready(" in run_body and "condvar.wait(" in run_body

    # ---- where the files go, and that every path they name resolves ----
    # module tree of the crate as the reference has it (names only, tests/golden/make_api_signatures.py) + the shipped modules
    from rs_api import mod_decls, pub_items
    tree = {k: (set(v) if v is not None else None) for k, v in ref["_module_tree"].items()}
    shipped = {"crate::mi355x": set(pub_items(strip_comments(rs["mi355x.rs"])))}
    assert mod_decls(strip_comments(rs["mi355x.rs"])) == ["doppler_shift", "do_acquisition", "do_tracking", "fft"]
    for sub in mod_decls(strip_comments(rs["mi355x.rs"])):
        assert os.path.exists(os.path.join(root, "rust", "src", "mi355x", sub + ".rs")), sub
        shipped["crate::mi355x::" + sub] = set(pub_items(strip_comments(rs[sub + ".rs"])))
    assert "mi355x" not in ref["_lib_rs_mods"] and "crate::mi355x" not in tree          # a NEW module: nothing is replaced
    n_paths = 0
    for name, text in rs.items():
        for sym, path in imports(strip_comments(text)).items():
            if not path.startswith("crate"):
                continue
            n_paths += 1
            where = shipped if path.startswith("crate::mi355x") else tree
            assert path in where and where[path] is not None, (name, path, "no such module")
            if sym != "*":
                assert sym in where[path], (name, path, sym, "no such pub item in that module")
    assert n_paths >= 14, n_paths
    # none of the wrappers imports from a module it claims to replace: they are siblings, and say so
    for name in ("doppler_shift.rs", "do_acquisition.rs", "do_tracking.rs", "fft.rs"):
        assert "DESTINATION: src/mi355x/%s" % name in rs[name], name
    assert "DESTINATION: src/mi355x.rs" in rs["mi355x.rs"]
    # the three patches: lib.rs gains `pub mod mi355x;` behind its last module; main.rs switches exactly the two stage modules,
    # and the lines it removes are the reference's own imports
    lib_diff = open(os.path.join(root, "rust", "patches", "lib_rs.diff")).read()
    # (the reference's last line `pub mod constants;` has no trailing newline, so `diff -u` re-states it: tests/test_rust_patches.py)
    assert "+pub mod mi355x;" in lib_diff and "+pub mod %s;\n+pub mod mi355x;" % ref["_lib_rs_mods"][-1] in lib_diff
    main_diff = open(os.path.join(root, "rust", "patches", "main_rs.diff")).read()
    removed = [l[1:].strip() for l in main_diff.splitlines() if l.startswith("-") and not l.startswith("---")]
    added = [l[1:].strip() for l in main_diff.splitlines() if l.startswith("+") and not l.startswith("+++")]
    assert removed == ["use gnss_sdr_rs::acquisition::do_acquisition;", "use gnss_sdr_rs::tracking::do_tracking;"]
    assert added == ["use gnss_sdr_rs::mi355x::do_acquisition;", "use gnss_sdr_rs::mi355x::do_tracking;"]
    mi = ref["_main_rs_imports"]
    assert mi["do_acquisition"] == "gnss_sdr_rs::acquisition" and mi["do_tracking"] == "gnss_sdr_rs::tracking"
    assert mi["AcquisitionResult"] == "gnss_sdr_rs::acquisition::do_acquisition" and mi["TrackingMessage"] == "gnss_sdr_rs::tracking::do_tracking"
    # and INTEGRATION.md says the same
    integ = open(os.path.join(root, "INTEGRATION.md")).read()
    for needle in ("src/mi355x.rs", "src/mi355x/do_acquisition.rs", "src/mi355x/do_tracking.rs", "src/mi355x/doppler_shift.rs",
                   "src/mi355x/fft.rs", "pub mod mi355x;", "use gnss_sdr_rs::mi355x::do_acquisition;", "use gnss_sdr_rs::mi355x::do_tracking;"):
        assert needle in integ, needle


def test_beidou_b1i_codes_known_properties(gm):
    """gm_b1i_code (BDS-SIS-ICD-B1I 11-stage Gold codes; not in the reference): equal to an independent restatement of the
    generator, and with the published properties of a Gold family of degree 11 — both LFSRs maximal (period 2047), periodic
    auto- and cross-correlations of the untruncated codes three-valued {-1, -65, 63}, the ICD's code = the first 2046 chips."""
    from gnss_sdr_rs_amd import acquisition as A
    full = A.b1i_codes(range(1, 38), 2047).astype(np.int64)
    code = A.b1i_codes()
    assert code.shape == (37, 2046) and (full[:, :2046] == code).all() and set(np.unique(code)) == {-1, 1}

    def lfsr(taps):
        reg = [k & 1 for k in range(11)]                    # 0 1 0 1 0 1 0 1 0 1 0, stage k at index k-1
        out = []
        for _ in range(2047):
            out.append(list(reg))
            fb = 0
            for t in taps:
                fb ^= reg[t - 1]
            reg = [fb] + reg[:10]
        return np.array(out, np.uint8)
    g1, g2 = lfsr([1, 7, 8, 9, 10, 11]), lfsr([1, 2, 3, 4, 5, 8, 9, 11])
    assert len({tuple(r) for r in g1}) == 2047 and len({tuple(r) for r in g2}) == 2047          # m-sequences
    phases = [(1, 3), (1, 4), (1, 5), (1, 6), (1, 8), (1, 9), (1, 10), (1, 11), (2, 7), (3, 4), (3, 5), (3, 6), (3, 8), (3, 9), (3, 10),
              (3, 11), (4, 5), (4, 6), (4, 8), (4, 9), (4, 10), (4, 11), (5, 6), (5, 8), (5, 9), (5, 10), (5, 11), (6, 8), (6, 9), (6, 10),
              (6, 11), (8, 9), (8, 10), (8, 11), (9, 10), (9, 11), (10, 11)]
    for p, (a, b) in enumerate(phases):
        bits = g1[:, 10] ^ g2[:, a - 1] ^ g2[:, b - 1]
        assert (np.where(bits > 0, -1, 1) == full[p]).all(), p + 1
    F = np.fft.fft(full, axis=1)
    vals = set()
    for i in range(37):
        for j in (i, (i + 5) % 37, (i + 11) % 37):
            cc = np.round(np.fft.ifft(F[i] * np.conj(F[j])).real).astype(int)
            vals |= set(np.unique(cc[1:] if i == j else cc).tolist())
            if i == j:
                assert cc[0] == 2047
    assert vals == {-65, -1, 63}


def test_receiver_harness_loads_and_fails_loudly_without_a_device(gm):
    """host/receiver_harness.cpp (the C++ stage drivers behind an extern "C" surface, what bench.py's receiver leg times): the
    library is built by build(), exports its four entries, links the product library and nothing of the oracle, and — no CPU
    fallback — returns GM_ERR_NO_DEVICE on a box without a GPU."""
    from gnss_sdr_rs_amd import receiver as R
    L = R.lib()
    assert L.gmrx_abi_version() == 1
    nm = subprocess.run(["nm", "-D", "--defined-only", R.library_path()], stdout=subprocess.PIPE, text=True).stdout
    assert set(re.findall(r" T (gmrx_[a-z0-9_]+)", nm)) == {"gmrx_abi_version", "gmrx_last_error", "gmrx_receiver_run", "gmrx_tracking_ab"}
    out = subprocess.run(["ldd", R.library_path()], stdout=subprocess.PIPE, text=True).stdout
    assert "libgnss_mi355x.so" in out and "liboracle" not in out and "torch" not in out
    code = ("import sys; sys.path.insert(0, %r); import numpy as np, ctypes as C; import gnss_sdr_rs_amd as g;"
            "from gnss_sdr_rs_amd import receiver as R; L = g.lib(); n = C.c_int(0); L.gm_device_count(C.byref(n));\n"
            "try:\n    R.receiver_run(np.zeros(2 * 4096, np.int8), 4.096e6, 0.0, warmup_calls=0); print(n.value, 0)\n"
            "except g.GmError as e:\n    print(n.value, e.status)") % ROOT
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    ndev, rc = r.stdout.split()[-2:]
    if int(ndev) == 0:
        assert int(rc) == -3, r.stdout + r.stderr
