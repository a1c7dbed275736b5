#!/bin/bash
# kernel timeline of the last warm frame of a bench_extra leg (GPU box, repo root).  usage: tools/frame_timeline.sh csg|photon|mirror|glass [kernels]
root=$PWD
leg=${1:-csg}; n=${2:-16}
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/prof_frame
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/prof_frame -o fr -- python3 $root/bench_extra.py $leg > /dev/null 2>&1 || exit 1
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$root/gpurun_out/prof_frame/fr_kernel_trace.csv"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last run of consecutive rrt kernels that starts with a level-0 sized k_trace_shade
idx=[i for i,r in enumerate(rows) if "k_trace_shade" in r["Kernel_Name"]]
big=max(int(rows[i].get("Grid_Size_X", rows[i].get("Grid_Size", 0)) or 0) for i in idx)
starts=[i for i in idx if int(rows[i].get("Grid_Size_X", rows[i].get("Grid_Size", 0)) or 0)==big]
start=starts[int("${3:-0}")] if "${3:-}" else (starts[-2] if len(starts)>1 else starts[-1])
t0=int(rows[start]["Start_Timestamp"]); prev=t0
for r in rows[start:start+$n]:
    nme=r["Kernel_Name"].split("(")[0].replace("void rrt::","").replace("rrt::","")[:34]
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print("%-34s start %8.1f us  gap %6.1f  dur %8.1f us  grid %s" % (nme,(s-t0)/1e3,(s-prev)/1e3,(e-s)/1e3,r.get("Grid_Size_X", r.get("Grid_Size",""))))
    prev=e
PY
