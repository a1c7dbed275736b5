#!/bin/bash
# a round's evidence in one GPU call: kernel statistics and the PMC passes of the bench's own command (configs[3], mesh)
# and of the photon leg (configs[2]), then the default bench line.  usage (GPU box, repo root): tools/profile_all.sh <tag, e.g. r03>
# (tools/collect_profiles.py <tag> turns gpurun_out/ into profiles/<tag>/ afterwards)
set -u
tag=${1:-r03}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p "$out"
bench="python3 $root/bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 3"
photon="python3 $root/bench_extra.py photon"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/c4_stats" -o c4 -- $bench > "$out/c4_stats.json.log" 2>&1 ) || { tail -5 "$out/c4_stats.json.log"; exit 1; }
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/c3_stats" -o c3 -- $photon > "$out/c3_stats.json.log" 2>&1 ) || { tail -5 "$out/c3_stats.json.log"; exit 1; }
tools/pmc_passes.sh ${tag}_c4 -- $bench || exit 1
tools/pmc_passes.sh ${tag}_c3 -- $photon || exit 1
tools/pmc_mem_passes.sh ${tag}_c4_mem -- $bench || exit 1
for f in "$out"/c4_stats/*kernel_stats.csv "$out"/c3_stats/*kernel_stats.csv; do echo "== $f"; head -5 "$f" | cut -c1-160; done
