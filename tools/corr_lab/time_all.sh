#!/bin/bash
# time every lab_* binary of tools/corr_lab (GPU box): "us name" per line into gpurun_out/plan_times.txt
cd "$(dirname "$0")"
out=../../gpurun_out/plan_times.txt; : > $out
i=0
for b in lab_*; do
  t=$(timeout -k 5 60 ./$b 32 41 10 | tail -1 | sed -n 's/.*median \([0-9.]*\) us.*/\1/p')
  echo "$t $b" >> $out
  i=$((i+1)); [ $((i % 50)) = 0 ] && echo "$i done"
done
echo finished
