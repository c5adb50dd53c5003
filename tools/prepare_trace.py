"""Reads a rocprofv3 kernel trace of tools/prepare_ab.py and prints the kernels of a few consecutive dwells (start / end relative to
the first, queue) from the middle of the LAST prepared leg at N = 16368."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "16368" in r["Kernel_Name"] or "decide" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the prepared legs have mix kernels on another queue than the corr kernels
mix = [r for r in rows if "mix_fft" in r["Kernel_Name"]]
corr_q = {r["Queue_Id"] for r in rows if "corr" in r["Kernel_Name"]}
side = [r for r in mix if r["Queue_Id"] not in corr_q]
print("mix launches", len(mix), "on a side queue", len(side))
if side:
    t_mid = int(side[len(side) - 40]["Start_Timestamp"])
    sel = [r for r in rows if int(r["Start_Timestamp"]) >= t_mid][:16]
    t0 = int(sel[0]["Start_Timestamp"])
    for r in sel:
        n = r["Kernel_Name"]
        n = "stage F" if "mix_fft" in n else ("stage C" if "corr" in n else "decision")
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print(f"  {n:9s} queue {r['Queue_Id']:>3s} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8s} start {s:8.1f} end {e:8.1f} dur {e - s:7.1f} us")
