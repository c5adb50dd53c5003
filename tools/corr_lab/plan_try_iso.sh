#!/bin/bash
# plan_try.sh in a directory of its own (parallel builds: -save-temps writes fixed file names)
set -e
H="$(cd "$(dirname "$0")" && pwd)"
D=$(mktemp -d /tmp/plan_XXXX)
cd $D
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt \
      -I$H -DLAB_PLAN="$2" $H/corr_lab.hip -o $D/lab -save-temps=obj 2>/dev/null || { echo "compile failed"; rm -rf $D; exit 0; }
python3 - corr_lab-hip-amdgcn-amd-amdhsa-gfx950.s <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
best = None
for m in re.finditer(r"^(_ZN2gm\d+(?:acq_corr|comp_corr)\w*kernel[^:\s]*):[^\n]*\n(.*?)^\.Lfunc_end", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    sc = len(re.findall(r"\bscratch_(?:load|store)", body))
    k = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(.*?)\.end_amdhsa_kernel", txt, re.S)
    vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", k.group(1)).group(1) if k else "?"
    if name.endswith("Lb0EEEvPKNS_2cfES5_S5_PfPjS6_PKjiiiiiiiS6_S7_i") or best is None:
        best = (sc, vg)
print("scratch_insts %d vgprs %s" % best)
PY
cp $D/lab $H/lab_$1; rm -rf $D
