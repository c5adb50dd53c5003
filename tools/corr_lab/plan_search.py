"""tools/corr_lab/plan_search.py N [--threads T1,T2] [--passes 3,4] [--max N]: enumerate the radix plans Plan<N, T, r...> the in-LDS FFT core can
express for N (ordered factorisations into the radices fft_core.h has a Dft<> for, one pass-0 butterfly per lane), build the stage-C
laboratory for each (plan_try.sh, here, no GPU; 6 at a time) and list scratch_* instructions / registers.  The zero-scratch binaries
lab_<name> are then timed on the GPU box: `for b in lab_sN_*; do ./$b 32 41 10; done`."""
import itertools, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
HERE = os.path.dirname(os.path.abspath(__file__))
RAD = [3, 4, 5, 7, 8, 10, 11, 13, 15, 16, 20, 24, 25, 31, 32, 33]


def facts(n, k):
    if k == 1:
        return [(n,)] if n in RAD else []
    out = []
    for r in RAD:
        if n % r == 0:
            out += [(r,) + t for t in facts(n // r, k - 1)]
    return out


def main():
    N = int(sys.argv[1])
    arg = lambda k, d: sys.argv[sys.argv.index(k) + 1] if k in sys.argv else d
    Ts = [int(t) for t in arg("--threads", "256,384,512,768,1024").split(",")]
    passes = [int(p) for p in arg("--passes", "3,4").split(",")]
    limit = int(arg("--max", "80"))
    IT0 = int(arg("--it0", "1"))
    cands = []
    for k in passes:
        for f in facts(N, k):
            if max(f) > 33 or (k == 4 and min(f) < 4 and N % 3):
                continue
            for T in Ts:
                if N // f[0] > IT0 * T or N // max(f) > 4 * T or T * max(f) > 3 * N:    # IT0 = 1; no pass with > 4 butterflies per lane; no mostly idle workgroup
                    continue
                cands.append((T,) + f)
    cands = sorted(set(cands))[:limit]
    print(len(cands), "candidates", flush=True)

    def build(c):
        name = "s%d_%s" % (N, "_".join(map(str, c)))
        plan = "gm::Plan<%d,%s>" % (N, ",".join(map(str, c)))
        r = subprocess.run(["bash", os.path.join(HERE, "plan_try_iso.sh"), name, plan], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return name, r.stdout.strip().splitlines()[-1:] or ["?"]
    with ThreadPoolExecutor(6) as ex:
        for name, out in ex.map(build, cands):
            print(name, out[0], flush=True)


main()
