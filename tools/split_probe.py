"""Diagnostic: acq_corr_kernel's tail split must not change a single word of the metrics.  Runs the configs[1] scene with the
split on (default) several times and once with GM_CORR_SPLIT=0 in a child process, and reports every (worker, bin) that differs."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def planes(tag):
    from gnss_sdr_rs_amd import _lib, acquisition as A, synth
    _lib.init(0)
    sc = synth.cfg2_scene(A.ca_code_table())
    x = synth.to_i8_iq(sc["x"])
    eng = A.AcquisitionEngine(sc["fs"], sc["f_if"], sc["N"], doppler_hz=sc["doppler_hz"], n_integrations=sc["M"])
    outs = []
    for _ in range(4):
        eng.search(x)
        outs.append(np.stack([a.view(np.uint32) for a in eng.metrics()]))
    np.save(os.path.join(ROOT, "gpurun_out", "split_%s.npy" % tag), np.stack(outs))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        planes(sys.argv[1])
        sys.exit(0)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    subprocess.run([sys.executable, __file__, "on"], check=True)
    subprocess.run([sys.executable, __file__, "off"], check=True, env=dict(os.environ, GM_CORR_SPLIT="0"))
    on, off = np.load(os.path.join(ROOT, "gpurun_out", "split_on.npy")), np.load(os.path.join(ROOT, "gpurun_out", "split_off.npy"))
    for i in range(1, 4):
        print("off run", i, "== off run 0:", bool((off[i] == off[0]).all()))
    for i in range(4):
        d = np.argwhere(on[i] != off[0])
        print("on run", i, "differs from off in", len(d), "words")
        for q, p, b in d[:12]:
            print("   plane", q, "worker", p, "bin", b, on[i][q, p, b].view(np.float32) if q != 1 else on[i][q, p, b],
                  off[0][q, p, b].view(np.float32) if q != 1 else off[0][q, p, b])
