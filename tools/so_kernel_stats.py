"""Per-kernel register / scratch statistics of a built library (default: the product library): the code objects are taken out of
the fat binary with `llvm-objdump --offloading`, the kernel metadata read with `llvm-readelf --notes`, and the scratch_* instructions
of every kernel counted in the disassembly.  Usage: python tools/so_kernel_stats.py [path/to/lib.so] [--all] [--json out.json]
Prints the kernels that use scratch memory (every kernel with --all), demangled, largest first."""
import json, os, re, shutil, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_stats(lib):
    tmp = tempfile.mkdtemp(prefix="gm_co_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        out = {}
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            co = os.path.join(tmp, f)
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], stdout=subprocess.PIPE, text=True).stdout
            for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
                g = lambda k: re.search(r"\.%s:\s*(\S+)" % k, blk)
                name = g("name").group(1)
                out[name] = dict(vgpr=int(g("vgpr_count").group(1)), sgpr=int(g("sgpr_count").group(1)),
                                 scratch_bytes=int(g("private_segment_fixed_size").group(1)),
                                 vgpr_spills=int(g("vgpr_spill_count").group(1)) if g("vgpr_spill_count") else 0,
                                 lds_bytes=int(g("group_segment_fixed_size").group(1)), scratch_insts=0, code_object=f)
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], stdout=subprocess.PIPE, text=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
                if m:
                    cur = m.group(1) if m.group(1) in out else None
                elif cur and re.search(r"\bscratch_(load|store)", line):
                    out[cur]["scratch_insts"] += 1
        names = list(out)
        dem = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True).stdout.splitlines()
        for n, d in zip(names, dem):
            out[n]["demangled"] = d
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def short(d):
    d = re.sub(r"gm::|\(anonymous namespace\)::", "", d)
    depth, cut = 0, len(d)
    for i, ch in enumerate(d):                 # the argument list starts at the first '(' outside the template brackets
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            cut = i
            break
    return d[:cut]


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(ROOT, "gnss-sdr-rs_amd", "lib", "libgnss_mi355x.so")
    st = kernel_stats(lib)
    if "--json" in sys.argv:
        json.dump(st, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    rows = sorted(st.values(), key=lambda r: -r["scratch_insts"])
    print("%s: %d kernels, %d with scratch instructions, %.1f MB" % (lib, len(rows), sum(1 for r in rows if r["scratch_insts"]), os.path.getsize(lib) / 1e6))
    for r in rows:
        if r["scratch_insts"] or "--all" in sys.argv:
            print("%5d scratch insts %6d B  vgpr %3d  lds %6d  %s" % (r["scratch_insts"], r["scratch_bytes"], r["vgpr"], r["lds_bytes"], short(r["demangled"])[:170]))
