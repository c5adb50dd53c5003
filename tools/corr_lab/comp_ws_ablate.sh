#!/bin/bash
# Runs ON THE GPU BOX.  Timing-only ablations of comp_corr_ws_kernel<16000> (Q = 2, the BASELINE configs[3] Galileo geometry) on random
# data: the shipped kernel against copies whose pass-0 waves skip (a) the code-side loads, (b) the spectrum loads, (c) both — the question
# VERDICT round 5 item 5 asks: is the kernel bound by its load BYTES (then sharing the code-side loads between the Q sub-transforms pays)
# or by the load PATH's occupancy / the rest?  The ablated kernels compute garbage; nothing of this is in the product sources: the copies
# are made by sed from csrc/ into a scratch directory.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
L=$R/tools/corr_lab
W=/tmp/comp_ws_ablate; rm -rf $W; mkdir -p $W/tools/corr_lab
variant() {   # name, sed expression applied to acq_comp_ws.h
    mkdir -p $W/$1/gnss-sdr-rs_amd $W/$1/tools/corr_lab $W/$1/include
    cp -r $R/gnss-sdr-rs_amd/csrc $W/$1/gnss-sdr-rs_amd/; cp $R/include/gnss_mi355x.h $W/$1/include/
    [ -n "$2" ] && sed -i -E "$2" $W/$1/gnss-sdr-rs_amd/csrc/acq_comp_ws.h
    cp $L/comp_ws_stamps.hip $W/$1/tools/corr_lab/
    (cd $W/$1/tools/corr_lab && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize comp_ws_stamps.hip -o lab 2> build.log) || { tail -5 $W/$1/tools/corr_lab/build.log; return 1; }
    for i in 1 2 3; do printf "%-22s " $1; $W/$1/tools/corr_lab/lab | grep "^kernel:"; done
}
variant shipped ""
variant no_code_loads   's/c\[h\]\[k1\] = __builtin_amdgcn_raw_buffer_load_b128\(crs.*$/c[h][k1] = x[h][k1] ^ u32x4{1u, 2u, 3u, 4u};/'
variant no_spectrum_loads 's/x\[h\]\[k1\] = __builtin_amdgcn_raw_buffer_load_b128\(xrs.*$/x[h][k1] = u32x4{0x3f800000u + unsigned(v16), 0x3f000000u, 0x3f800000u + unsigned(st), 0x3e800000u};/'
