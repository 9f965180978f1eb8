#!/bin/bash
# Register / scratch / LDS use of every kernel (clang's kernel-resource-usage remarks) and the device assembly.
#   scripts/kernel_info.sh [out_dir]     -> out_dir/resources.txt, out_dir/rt_lib.s
OUT=${1:-/tmp/rt_kinfo}
mkdir -p $OUT
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o $OUT/rt_lib.s \
  -Rpass-analysis=kernel-resource-usage raytracer_amd/csrc/rt_lib.hip 2> $OUT/remarks.txt
python3 - "$OUT" <<'PY'
import re, sys
out = sys.argv[1]
txt = open(out + "/remarks.txt").read()
rows = []
for blk in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = blk.split("\n")[0].strip().split(" [-R")[0]
    import subprocess
    name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(rtd::Scene.*", "", name).replace("void ", "")
    g = lambda k: (re.search(k + r": (\d+)", blk) or [None, "?"])[1]
    rows.append((name, g("VGPRs"), g("AGPRs"), g("SGPRs"), g("ScratchSize \[bytes/lane\]"), g("Occupancy \[waves/SIMD\]"), g("LDS Size \[bytes/block\]")))
with open(out + "/resources.txt", "w") as f:
    for r in rows:
        f.write("%-52s vgpr %s agpr %s sgpr %s scratch %s occ %s lds %s\n" % r)
print(open(out + "/resources.txt").read())
PY
