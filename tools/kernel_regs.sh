#!/bin/bash
# registers / scratch / LDS / occupancy of the kernels of one source file, as the compiler reports them:
#   tools/kernel_regs.sh [file.hip] [name filter] [extra flags...]
f=${1:-kernels.hip}; filter=${2:-k_trace_shade}; shift 2 2>/dev/null
cd "$(dirname "$0")/../rust-raytracer_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-result --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /dev/null 2>&1 | python3 -c '
import re, subprocess, sys
flt = sys.argv[1]; name = None; row = []
for l in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", l)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip(); row = []
    elif name:
        row.append(t)
        if t.startswith("LDS Size"):
            if flt in name:
                dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                keep = [r for r in row if re.match(r"(VGPRs|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill)", r)]
                print(dn.split("(")[0], "|", " | ".join(keep))
            name = None
' "$filter"
