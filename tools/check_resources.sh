#!/bin/bash
# Register / scratch / occupancy report of the library's kernels (run after growing a kernel: a parameter block in scratch memory costs ~35 us per launch,
# spills show up as WRITE_SIZE traffic -- DESIGN.md "rules learnt on the hardware").  usage: tools/check_resources.sh [kernel-name-substring ...]
cd "$(dirname "$0")/../mvil-fusion_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -instcombine-max-copied-from-constant-users=10000 -mllvm -disable-machine-licm \
  -Rpass-analysis=kernel-resource-usage -c ${SRC:-vilsolve.hip} -o /tmp/vilsolve_res.o 2> /tmp/vilsolve_res.txt
python3 - "$@" <<'PY'
import re, sys
txt = open('/tmp/vilsolve_res.txt').read()
pat = re.compile(r"Function Name: (\S+).*?TotalSGPRs: (\d+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", re.S)
for m in pat.finditer(txt):
    name = m.group(1)
    if len(sys.argv) > 1 and not any(a in name for a in sys.argv[1:]): continue
    print("%-60s sgpr %3s vgpr %3s scratch %4s occ %s lds %s" % (name[:60], m.group(2), m.group(3), m.group(4), m.group(5), m.group(6)))
PY
