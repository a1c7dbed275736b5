#!/bin/bash
# kernel statistics of the photon index build (GPU box, repo root): tools/index_prof.sh [photons]
set -u
root=$PWD
out=$root/gpurun_out/index_prof
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o ix -- python3 $root/tools/index_ab.py ${1:-1000000} > "$out/run.log" 2>&1 || { tail -5 "$out/run.log"; exit 1; }
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print("%-90s calls %5s avg_us %9.1f total_ms %8.3f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
