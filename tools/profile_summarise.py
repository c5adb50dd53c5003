"""Turn gpurun_out/prof (tools/profile_round.sh) into the committed summaries under profiles/ for round RR:
   rRR_bench_line.json, rRR_bench_kernel_stats.csv, rRR_pmc_fetch_write.json, traffic.json."""
import csv
import glob
import json
import os
import re
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rr = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(root, "gpurun_out", "prof")
dst = os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "bench_line.json"), os.path.join(dst, f"{rr}_bench_line.json"))
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
shutil.copy(stats[0], os.path.join(dst, f"{rr}_bench_kernel_stats.csv"))


def by_grid():
    """Per (kernel, grid size) launch count and average duration from the --stats pass's per-dispatch trace: bench.py's
    other legs launch acq_corr_kernel<Plan8000> too (the GPS family of the cfg4 grid, one-PRN searches), so the plain
    --stats average of that name mixes shapes; the headline launches are the rows with the headline grid size."""
    tr = glob.glob(os.path.join(src, "stats", "**", "*kernel_trace.csv"), recursive=True)
    if not tr:
        return
    acc = {}
    for row in csv.DictReader(open(tr[0])):
        k = (short(row["Kernel_Name"]), int(row["Grid_Size_X"]) // max(1, int(row["Workgroup_Size_X"])), int(row["Workgroup_Size_X"]))
        a = acc.setdefault(k, [0, 0.0, 1e30, 0.0])
        dt = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        a[0] += 1; a[1] += dt; a[2] = min(a[2], dt); a[3] = max(a[3], dt)
    with open(os.path.join(dst, f"{rr}_bench_kernel_by_grid.csv"), "w") as f:
        f.write("kernel,workgroups,workgroup_size,launches,avg_ns,min_ns,max_ns\n")
        for (k, g, w), (n, t, lo, hi) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            if k.startswith("gm::"):
                f.write(f"{k},{g},{w},{n},{t / n:.0f},{lo:.0f},{hi:.0f}\n")


def short(name):
    m = re.search(r"gm::(?:\(anonymous namespace\)::)?(\w+)", name)
    if not m:
        return name.split("(")[0][:60]
    k = "gm::" + m.group(1)
    pl = re.search(r"Plan<(\d+)", name)          # one entry per transform size: the bench workload is N = 8000
    if pl and pl.group(1) != "8000":
        k += f"<N={pl.group(1)}>"
    arms = re.search(r"trk_persistent_kernel<(\d+)", name)
    if arms and arms.group(1) != "3":
        k += f"<arms={arms.group(1)}>"
    return k


def wgs(row):
    return int(row["Grid_Size"]) // max(1, int(row["Workgroup_Size"]))


# FETCH_SIZE / WRITE_SIZE per (kernel, grid size in workgroups): bench.py's other legs launch acq_corr_kernel<Plan8000> at
# other grid sizes (the 22-code B1I family of the cfg4 grid: 1320 workgroups; one-PRN searches), so a per-name average
# would mix shapes.  "headline" = the grid size with the most launches of that kernel name (the timed region's).
out, by_shape = {}, {}
for ctr, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    files = glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True)
    acc = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != ctr:
                continue
            k = short(row["Kernel_Name"])
            a = acc.setdefault((k, wgs(row)), [0, 0.0])
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    per = {}
    for (k, g), (n, v) in acc.items():
        if k.startswith("gm::"):
            per.setdefault(k, {})[g] = {"launches": n, "avg_counter_KB": v / n}
    by_shape[ctr] = per
    out[ctr] = {}
    for k, shapes in per.items():
        # the benchmarked shape = the grid size that carries most of the kernel's counter total (launches x average): bench.py's M = 1
        # leg launches the same kernel name more often than the timed region does, at a grid size that moves a tenth of the bytes
        head = max(shapes, key=lambda g: shapes[g]["launches"] * shapes[g]["avg_counter_KB"])
        out[ctr][k] = dict(shapes[head], workgroups=head,
                           other_grid_sizes={str(g): v for g, v in sorted(shapes.items()) if g != head})
json.dump(out, open(os.path.join(dst, f"{rr}_pmc_fetch_write.json"), "w"), indent=1)
by_grid()


def kb(ctr, k):
    return out[ctr].get(k, {}).get("avg_counter_KB", 0.0)


def hbm_bytes(k):
    return int((2 * kb("FETCH_SIZE", k) + kb("WRITE_SIZE", k)) * 1024)


# MI355X_MICROARCH.md §HBM: FETCH_SIZE under-reports coalesced streaming reads by 2x on gfx950 (64 B per 128-B request);
# WRITE_SIZE is exact.  Units: KB.
traffic = {
    "source": f"profiles/{rr}_pmc_fetch_write.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py workload); "
              "per kernel the launches of its most frequent grid size only (the timed region's shape)",
    "correction": "FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request), WRITE_SIZE exact; KB -> bytes x1024",
    "acq_corr_kernel_hbm_bytes_per_launch": hbm_bytes("gm::acq_corr_kernel"),
    "acq_corr_kernel_workgroups": out["FETCH_SIZE"].get("gm::acq_corr_kernel", {}).get("workgroups"),
    "acq_mix_fft_kernel_hbm_bytes_per_launch": hbm_bytes("gm::acq_mix_fft_kernel"),
    "trk_persistent_kernel_hbm_bytes_per_launch": hbm_bytes("gm::trk_persistent_kernel"),
    "trk_persistent_kernel_workgroups": out["FETCH_SIZE"].get("gm::trk_persistent_kernel", {}).get("workgroups"),
    "acq_corr_ws31_kernel_N16368_hbm_bytes_per_launch": hbm_bytes("gm::acq_corr_ws31_kernel<N=16368>"),
    "acq_corr_ws31_kernel_N16368_workgroups": out["FETCH_SIZE"].get("gm::acq_corr_ws31_kernel<N=16368>", {}).get("workgroups"),
    "comp_corr_kernel_hbm_bytes_per_launch": {k: hbm_bytes(k) for k in out["FETCH_SIZE"] if "comp_corr" in k},
    "note": "fabric (L2 memory-side) bytes; Infinity-Cache hits are counted, so true HBM traffic is at most this",
}
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))


# SQ / LDS counters (tools/pmc_sq.sh -> gpurun_out/pmc_sq/{a,b,c}): per-launch averages per (kernel, grid size); the entry of
# a kernel name is its most frequent grid size (the benchmarked shape), the others are listed under "other_grid_sizes"
sqdir = os.path.join(root, "gpurun_out", "pmc_sq")
sq, sq_shapes = {}, {}
for f in glob.glob(os.path.join(sqdir, "*", "*counter_collection.csv")):
    acc = {}
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if not k.startswith("gm::"):
            continue
        a = acc.setdefault((k, wgs(row), row["Counter_Name"]), [0, 0.0])
        a[0] += 1
        a[1] += float(row["Counter_Value"])
    for (k, g, c), (n, v) in acc.items():
        e = sq_shapes.setdefault(k, {}).setdefault(g, {})
        e[c] = v / n
        e["launches_" + os.path.basename(os.path.dirname(f))] = n
for k, shapes in sq_shapes.items():
    def weight(g):      # launches x wave-cycles (or the first counter present): the shape that carries most of the kernel's time
        n = max(v for c, v in shapes[g].items() if c.startswith("launches_"))
        vals = [v for c, v in sorted(shapes[g].items()) if not c.startswith("launches_")]
        return n * shapes[g].get("SQ_WAVE_CYCLES", vals[0] if vals else 1.0)
    head = max(shapes, key=weight)
    sq[k] = dict(shapes[head], workgroups=head)
    others = {str(g): v for g, v in sorted(shapes.items()) if g != head}
    if others:
        sq[k]["other_grid_sizes"] = others
if sq:
    sq["_source"] = ("rocprofv3 --pmc (SQ_* / GRBM_* groups in separate passes, --kernel-trace only) -- python3 bench.py --steps 6 "
                     "--warmup 2 --no-cpu-baseline; per kernel the launches of its most frequent grid size (`workgroups`); "
                     "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves")
    json.dump(sq, open(os.path.join(dst, "sq_counters.json"), "w"), indent=1, sort_keys=True)
    shutil.copy(os.path.join(dst, "sq_counters.json"), os.path.join(dst, f"{rr}_sq_counters.json"))
    print("sq counters:", sorted(k for k in sq if not k.startswith("_")))
