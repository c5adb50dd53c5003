#!/bin/bash
# Runs ON THE GPU BOX: SQ / LDS counter passes of a short bench run (counters only: no trace domains beside kernel-trace).
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_sq
rm -rf $OUT && mkdir -p $OUT
ARGS="--steps 6 --warmup 2 --no-cpu-baseline ${BENCH_EXTRA:---no-tracking}"
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/a -o run -- python3 $R/bench.py $ARGS > $OUT/a.log 2>&1 || exit 1
echo "pass a done"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/b -o run -- python3 $R/bench.py $ARGS > $OUT/b.log 2>&1 || exit 2
echo "pass b done"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $OUT/c -o run -- python3 $R/bench.py $ARGS > $OUT/c.log 2>&1 || echo "pass c failed (optional)"
find $OUT -name "*kernel_trace.csv" -size +2M -delete
du -sh $OUT
