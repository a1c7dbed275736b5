#!/bin/bash
# tools/probe_run.sh <variant>: per-primitive probe (k_batch_trace durations) for variants/librrt_<variant>.so
v=$1
export RRT_LIB=$GRAFT_REPO_ROOT/variants/librrt_$v.so
cd /tmp && export TMPDIR=/tmp
timeout -k 10 180 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/probe_$v -o probe -- python3 $GRAFT_REPO_ROOT/tools/gpu_latency_probe.py > $GRAFT_REPO_ROOT/gpurun_out/probe_$v.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/probe_$v.log; exit 1; }
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/probe_$v/probe_kernel_trace.csv")) if "batch_trace" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=["plane","+3quad","+2sph","+pipe","+coin","+3box","c2"]
print("== $v")
for i in range(0,len(rows),3):
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows[i:i+3]]
    print("%-8s 64K: %5.1f us   2M: %6.1f us"%(names[i//3],d[1],d[2]))
PY
