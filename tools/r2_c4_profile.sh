#!/bin/bash
# round 2: counter evidence for C4 (1 M-triangle mesh, 4K) and C3 (scene2 photon mode, 1 M photons): kernel statistics
# and the four PMC passes of `bench_extra.py mesh|photon`.  usage (GPU box, repo root): tools/r2_c4_profile.sh <tag>
set -u
tag=${1:-r02}
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
for w in mesh photon; do
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${w}_stats" -o $w -- python3 $root/bench_extra.py $w > "$out/${w}_stats.json.log" 2>&1 ) || { tail -5 "$out/${w}_stats.json.log"; exit 1; }
  tools/pmc_passes.sh ${tag}_$w -- python3 $root/bench_extra.py $w || exit 1
done
find "$out" -name "*kernel_stats.csv" -exec sh -c 'echo "== $1"; head -6 "$1"' _ {} \;
