#!/bin/bash
# time every mixlab_* binary (GPU box): "us name" per line into gpurun_out/mix_times.txt
cd "$(dirname "$0")"
out=../../gpurun_out/mix_times.txt; : > $out
i=0
for b in mixlab_*; do
  t=$(timeout -k 5 60 ./$b ${MIX_ARGS:-41 10} | sed -n 's/^stage F.*median \([0-9.]*\) us.*/\1/p')
  echo "$t $b" >> $out
  i=$((i+1)); [ $((i % 100)) = 0 ] && echo "$i done"
done
echo finished
