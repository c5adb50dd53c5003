#!/bin/bash
# Runs ON THE GPU BOX: counter passes (counters + --kernel-trace only) of ONE tool command, summarised for one kernel name.
#   tools/pmc_kernel.sh <out-subdir> <kernel-substring> <python script> [args...]       e.g.  tools/pmc_kernel.sh pmc_cfg5 trk_persistent tools/cfg5_time.py
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; KERN=$2; shift 2
rm -rf $OUT && mkdir -p $OUT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o run -- python3 $R/"$@" > $OUT/t.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/a -o run -- python3 $R/"$@" > $OUT/a.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/b -o run -- python3 $R/"$@" > $OUT/b.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/c -o run -- python3 $R/"$@" > $OUT/c.log 2>&1 || echo "pass c failed (optional)"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -o run -- python3 $R/"$@" > $OUT/f.log 2>&1 || echo "fetch failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -o run -- python3 $R/"$@" > $OUT/w.log 2>&1 || echo "write failed"
python3 - "$OUT" "$KERN" <<'PY' | tee $OUT/summary.txt
import csv, glob, os, sys
out, kern = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(os.path.join(out, "t", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0][-60:], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
            a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for (k, g), (n, t) in sorted(acc.items()):
    print("trace   %-62s wgs %5d launches %4d avg %.1f us" % (k, g, n, t / n / 1e3))
ctr = {}
for d in "abcfw":
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                k = (r["Kernel_Name"].split("(")[0][-60:], int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])), r["Counter_Name"])
                a = ctr.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
for (k, g, c), (n, v) in sorted(ctr.items()):
    print("counter %-62s wgs %5d %-28s launches %4d avg %.4g" % (k, g, c, n, v / n))
PY
find $OUT -name "*.csv" -size +1M -delete
du -sh $OUT
