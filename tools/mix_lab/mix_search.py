"""tools/mix_lab/mix_search.py: every radix plan the FFT core can express for stage F's forward transform at N = 8000 (3 - 5 passes on
512 / 768 / 1024 lanes), built into mix_lab through -DGM_MIX_PLAN_8000 (here, six at a time); the binaries mixlab_<name> are timed on
the GPU box by time_all.sh ("stage F ... median").  Same method as tools/corr_lab/plan_search.py."""
import os, subprocess, sys, tempfile, shutil
from concurrent.futures import ThreadPoolExecutor
HERE = os.path.dirname(os.path.abspath(__file__))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
RAD = [4, 5, 8, 10, 16, 20, 25] if N == 8000 else [3, 4, 8, 11, 16, 24, 31, 33]      # 16368 = 2^4 * 3 * 11 * 31
TS = (512, 768, 1024) if N == 8000 else (768, 1024)


def facts(n, k):
    if k == 1:
        return [(n,)] if n in RAD else []
    return [(r,) + t for r in RAD if n % r == 0 for t in facts(n // r, k - 1)]


def build(c):
    name = "_".join(map(str, c))
    d = tempfile.mkdtemp(prefix="mix_")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                        "-fhip-fp32-correctly-rounded-divide-sqrt", "-DGM_MIX_PLAN_%d=Plan<%d,%s>" % (N, N, ",".join(map(str, c))), "-DLAB_PLAN=gm::Plan%d" % N,
                        os.path.join(HERE, "mix_lab.hip"), "-o", os.path.join(d, "lab")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=d)
    ok = r.returncode == 0
    if ok:
        shutil.copy(os.path.join(d, "lab"), os.path.join(HERE, "mixlab_" + name))
    shutil.rmtree(d)
    return name, ok


cands = []
for k in (3, 4, 5):
    for f in facts(N, k):
        for T in TS:
            if N // max(f) > 2 * T or T * min(f) > 2 * N:       # no pass with > 2 butterflies per lane... nor a workgroup that is mostly idle in its widest pass
                continue
            cands.append((T,) + f)
cands = sorted(set(cands))
print(len(cands), "candidates", flush=True)
with ThreadPoolExecutor(6) as ex:
    for name, ok in ex.map(build, cands):
        print(name, "ok" if ok else "compile failed", flush=True)
