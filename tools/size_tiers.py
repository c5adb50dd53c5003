"""README's table of transform sizes: which stage-C kernel serves each size the acquisition handle accepts, how many scratch_*
instructions that kernel carries (tools/so_kernel_stats.py, no GPU), and — with a GPU — its measured time at one common geometry
(32 codes x 9 bins x 2 integrations, complex int8 samples resident), as ns per transform-sample = t_corr / (P D M N).
    python tools/size_tiers.py --measure out.json      (GPU box)      python tools/size_tiers.py --table out.json   (here: markdown)"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

IN_LDS = [8000, 16368, 16384, 16000, 15000, 12000, 10000, 8192, 8184, 6000, 5000, 4096, 4000, 2048, 2000, 1024, 512, 256]
BASES, QS = [16384, 16368, 16000, 8192, 8184, 8000, 6000, 5000, 4000], (2, 3, 4, 5, 6, 8)


def _composite():
    """every size Q x base the handle accepts, with the base it takes: the largest one that divides the size (acq_composite.hip)"""
    out = {}
    for b in BASES:
        for q in QS:
            n = q * b
            if n in IN_LDS:
                continue
            best = max(bb for bb in BASES if n % bb == 0 and n // bb in QS)
            out[n] = (n // best, best)
    return [out[n] for n in sorted(out)]


COMPOSITE = _composite()
P, D, M = (int(v) for v in os.environ.get("TIERS_PDM", "32,9,2").split(","))     # TIERS_PDM=32,41,4 TIERS_ONLY=6000,18000: an A/B at another grid


def measure(path):
    import numpy as np, torch
    from gnss_sdr_rs_amd import _lib, acquisition as A
    _lib.init(0)
    rng = np.random.default_rng(1)
    out = {}
    only = [int(v) for v in os.environ.get("TIERS_ONLY", "").split(",") if v]
    for q, base in [(1, n) for n in IN_LDS] + COMPOSITE:
        N = q * base
        if only and N not in only:
            continue
        fs = N * 1000.0
        dop = (np.arange(D, dtype=np.float32) - D // 2) * 250.0
        x = rng.integers(-60, 60, 2 * M * N, dtype=np.int8)
        try:
            eng = A.AcquisitionEngine(fs, 0.0, N, doppler_hz=dop, n_integrations=M)
        except Exception as e:
            out[str(N)] = {"error": repr(e)}
            continue
        d_x = torch.from_numpy(x).cuda()
        d_met = torch.zeros(3 * P * D, dtype=torch.int32, device="cuda")
        st = torch.cuda.Stream(); torch.cuda.set_stream(st)
        eng.set_stream(st.cuda_stream)
        for _ in range(3):
            eng.search_dev(d_x.data_ptr(), A.FMT_I8_IQ, d_met.data_ptr())
        torch.cuda.synchronize()
        eng.enable_timing(True)
        for _ in range(10):
            eng.search_dev(d_x.data_ptr(), A.FMT_I8_IQ, d_met.data_ptr())
        torch.cuda.synchronize()
        ts = eng.timing_summary()
        out[str(N)] = {"q": q, "base": base, "corr_ms": ts["avg_corr_ms"], "mix_fft_ms": ts["avg_mix_fft_ms"],
                       "ns_per_transform_sample": ts["avg_corr_ms"] * 1e6 / (P * D * M * N)}
        eng.close()
        print(N, out[str(N)], flush=True)
    json.dump(out, open(path, "w"), indent=1)


def table(path):
    from so_kernel_stats import kernel_stats, short
    st = kernel_stats(os.path.join(ROOT, "gnss-sdr-rs_amd", "lib", "libgnss_mi355x.so"))
    meas = json.load(open(path)) if path and os.path.exists(path) else {}
    by = {}
    for r in st.values():
        n = short(r["demangled"])
        m = re.match(r"void (acq_corr_kernel|acq_corr_ws31_kernel|comp_corr_kernel|comp_corr_ws_kernel)<(?:Hybrid)?(?:Plan|CorrPlan\w*)?<?(\d+)[^>]*>+,? ?(.*)>$", n)
        if not m:
            continue
        kern, base, rest = m.group(1), int(m.group(2)), m.group(3)
        flags = [f.strip() for f in rest.split(",")] if rest else []
        if kern == "acq_corr_kernel" and flags[-2:] != ["false", "false"]:
            continue                                   # the reference_products variant
        if kern == "acq_corr_ws31_kernel" and flags[-1:] != ["false"]:
            continue
        if kern == "comp_corr_kernel" and (len(flags) > 1 and flags[1] == "true"):
            continue                                   # the strict_sum_order (planes) variant
        if kern == "comp_corr_ws_kernel" and flags[-1] == "true":
            continue
        q = int(flags[0].rstrip("u")) if kern.startswith("comp") else 1
        by[(q, base)] = (kern, r["scratch_insts"], r["vgpr"], r["lds_bytes"])
    rows = []
    for q, base in [(1, n) for n in IN_LDS] + COMPOSITE:
        k = by.get((q, base))
        if not k and q == 1 and base == 16368:
            k = next((v for (qq, bb), v in by.items() if v[0] == "acq_corr_ws31_kernel"), None)
        me = meas.get(str(q * base), {})
        rows.append((q * base, ("%d x %d" % (q, base)) if q > 1 else "in LDS", k, me))
    print("| N | form | stage-C kernel | scratch_* instructions | VGPRs | LDS | stage C, ms (32 x 9 x 2) | ns per transform-sample |")
    print("|---|---|---|---|---|---|---|---|")
    for N, form, k, me in rows:
        kn, sc, vg, lds = k if k else ("?", "?", "?", "?")
        print("| %d | %s | `%s` | %s | %s | %s | %s | %s |" % (N, form, kn, sc, vg, lds, ("%.3f" % me["corr_ms"]) if "corr_ms" in me else "—",
                                                       ("%.3f" % me["ns_per_transform_sample"]) if "corr_ms" in me else "—"))


if __name__ == "__main__":
    if "--measure" in sys.argv:
        measure(sys.argv[sys.argv.index("--measure") + 1])
    else:
        table(sys.argv[sys.argv.index("--table") + 1] if "--table" in sys.argv else None)
