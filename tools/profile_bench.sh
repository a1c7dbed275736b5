#!/bin/bash
# rocprofv3 kernel statistics of bench.py: the default command (frames in flight) and the one-frame-at-a-time command
# whose per-launch durations the roofline figures are computed from.   usage (GPU box, repo root): tools/profile_bench.sh
set -u
out=$PWD/gpurun_out/prof_bench
mkdir -p "$out"
root=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/default" -o default -- python3 $root/bench.py --no-extra --no-cpu-baseline > "$out/default.json.log" 2>&1 || { tail -5 "$out/default.json.log"; exit 1; }
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/one" -o one -- python3 $root/bench.py --no-extra --no-cpu-baseline --frames-in-flight 1 --steps 20 --warmup 3 > "$out/one.json.log" 2>&1 || { tail -5 "$out/one.json.log"; exit 1; }
find "$out" -name "*kernel_stats.csv" -exec sh -c 'echo "== $1"; head -4 "$1"' _ {} \;
tail -1 "$out/default.json.log" | cut -c1-300; tail -1 "$out/one.json.log" | cut -c1-300
