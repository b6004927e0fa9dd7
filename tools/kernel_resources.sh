#!/bin/bash
# Register, scratch and occupancy figures of every kernel in one HIP source file, from the
# compiler's own resource-usage remarks.  usage: tools/kernel_resources.sh file.hip [grep-pattern]
set -e
f=$1; pat=${2:-.}
dir=$(dirname "$f")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$dir" -c "$f" -o /dev/null \
    -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re, sys
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else:
        cur[k.split()[0]] = v
print("%5s %5s %7s %4s  %s" % ("VGPR", "AGPR", "scratch", "occ", "kernel"))
for r in rows:
    print("%5s %5s %7s %4s  %s" % (r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r["name"]))
' | grep -E "VGPR|$pat"
