#!/bin/bash
# tools/corr_lab/plan_try.sh NAME 'gm::Plan<...>' : build the stage-C laboratory for one plan (here, no GPU), print the scratch_*
# instruction count and the register count of acq_corr_kernel; the binary lab_NAME runs on the GPU box (arguments: workers bins integrations).
set -e
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt \
      -DLAB_PLAN="$2" $3 corr_lab.hip -o lab_$1 -save-temps=obj 2>/dev/null || { echo "$1: compile failed"; exit 1; }
S=$(ls corr_lab-hip-amdgcn-amd-amdhsa-gfx950.s 2>/dev/null || ls lab_$1-hip-amdgcn*.s 2>/dev/null | head -1)
python3 - "$S" "$1" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"^(_ZN2gm\d+(?:acq_corr|comp_corr)\w*kernel[^:\s]*):[^\n]*\n(.*?)^\.Lfunc_end", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    sc = len(re.findall(r"\bscratch_(?:load|store)", body))
    k = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(.*?)\.end_amdhsa_kernel", txt, re.S)
    vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", k.group(1)).group(1) if k else "?"
    print(sys.argv[2], name[15:75], "scratch_insts", sc, "vgprs", vg)
PY
rm -f corr_lab-hip-* corr_lab-host-* lab_$1-hip-* lab_$1-host-* *.bc *.hipfb *.hipi *.cui *.o 2>/dev/null
