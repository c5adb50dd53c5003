for v in "" _vILP _vTRK; do
  L=$PWD/gnss-sdr-rs_amd/lib/libgnss_mi355x$v.so
  echo "== ${v:-default}"
  GM_LIB_PATH=$L timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-tracking 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']
        print('ms_per_step', round(d['ms_per_step'],4), 'corr_ms', round(d['roofline']['avg_launch_ms'],4), 'cfg1', round(c.get('cfg1_corr_kernel_ms',0),4), 'cfg4gal', round(c.get('cfg4_galileo_corr_kernel_ms',0),4), 'cfg4grid', round(c.get('cfg4_grid_ms_per_dwell',0),4))
" || exit 1
  GM_LIB_PATH=$L timeout -k 10 300 python tools/trk256_time.py 2>&1 | tail -2 || exit 1
  GM_LIB_PATH=$L timeout -k 10 300 python tools/cfg5_kernel_time.py 2>&1 | tail -1 || exit 1
done
