#!/bin/bash
# builds corr_lab variants for N = 16368 (round 4 candidates); run: for f in lab16_*; do ./$f 32 29 10; done
set -e
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function"
b() { name=$1; shift; hipcc $F "$@" corr_lab.hip -o lab16_$name 2> log_$name.txt & }
b base   -DLAB_PLAN='gm::Plan<16368,768,33,16,31>'
b t1024_16_33_31 -DLAB_PLAN='gm::Plan<16368,1024,16,33,31>'
b t1024_33_16_31 -DLAB_PLAN='gm::Plan<16368,1024,33,16,31>'
b t1024_33_31_16 -DLAB_PLAN='gm::Plan<16368,1024,33,31,16>'
wait
b t1024_31_33_16 -DLAB_PLAN='gm::Plan<16368,1024,31,33,16>'
b t1024_16_31_33 -DLAB_PLAN='gm::Plan<16368,1024,16,31,33>'
b t1024_16_3_11_31 -DLAB_PLAN='gm::Plan<16368,1024,16,3,11,31>'
b t1024_16_11_3_31 -DLAB_PLAN='gm::Plan<16368,1024,16,11,3,31>'
wait
b t1024_11_16_3_31 -DLAB_PLAN='gm::Plan<16368,1024,11,16,3,31>'
b t1024_16_33_31_pf0 -DLAB_PLAN='gm::Plan<16368,1024,16,33,31>' -DGM_CORR_PREFETCH_PAIRS=0
b t1024_48_11_31 -DLAB_PLAN='gm::Plan<16368,1024,48,11,31>'
wait
ls -la lab16_* | wc -l
