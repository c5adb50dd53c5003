#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the judged measurements of one round.
#   1. python bench.py                              -> gpurun_out/prof/bench_line.json
#   2. rocprofv3 --kernel-trace --stats of the SAME command -> kernel_stats.csv
#   3. separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (short bench, no CPU legs)
# Copy the summaries to profiles/ afterwards (tools/profile_summarise.py).
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout -k 10 400 python3 bench.py > $OUT/bench_stdout.log 2> $OUT/bench_stderr.log || exit 1
grep '^{' $OUT/bench_stdout.log | tail -1 > $OUT/bench_line.json
echo "bench done" 
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $R/bench.py > $OUT/stats_stdout.log 2>&1 || exit 2
echo "stats done"
export GM_BENCH_NO_TRK256=1   # the 256-channel tracking leg launches the same kernel at the same grid size as configs[2]: keep the counter averages per shape
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o run -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || exit 3
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o run -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || exit 4
echo "write done"
find $OUT -name "*.csv" | head -30
# the stats pass's per-dispatch trace stays (a few MB): profile_summarise.py averages the dominant kernel per grid size from it,
# because other bench legs launch the same kernel instantiation at other grid sizes and --stats averages over all of them
du -sh $OUT
