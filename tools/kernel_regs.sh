#!/bin/bash
# registers / scratch / LDS / occupancy of the kernels of one source file, as the compiler reports them:
#   tools/kernel_regs.sh [file.hip] [name filter] [extra flags...]
f=${1:-kernels.hip}; filter=${2:-k_trace_shade}; shift 2 2>/dev/null
cd "$(dirname "$0")/../rust-raytracer_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-result --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /dev/null 2>&1 | awk -v f="$filter" '
/Function Name:/ {name=$NF; keep = (name ~ f)} 
keep && /VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size|SGPRs:|Spill/ {sub(/.*remark: [^ ]* /,""); line = line " | " $0}
/LDS Size/ { if (keep) { cmd = "c++filt " name; cmd | getline dn; close(cmd); print dn line }; line="" }'
