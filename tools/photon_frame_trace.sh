#!/bin/bash
# kernel timeline of the last warm photon-mapped frame for several heavy-query thresholds (GPU box, repo root)
root=$PWD
cd /tmp && export TMPDIR=/tmp
for thr in ${THRS:-4294967295 0 256}; do
  rm -rf $root/gpurun_out/prof_phframe
  THR=$thr timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/prof_phframe -o ph -- python3 $root/tools/photon_frame_trace.py > /dev/null 2>&1 || exit 1
  echo "== threshold $thr"
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$root/gpurun_out/prof_phframe/ph_kernel_trace.csv"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_trace_shade" in r["Kernel_Name"]]
start=idx[-2]
t0=int(rows[start]["Start_Timestamp"])
for r in rows[start:start+5]:
    n=r["Kernel_Name"].split("(")[0].replace("void rrt::","").replace("rrt::","")[:28]
    print("%-28s start %8.1f us  dur %8.1f us  grid %s" % (n,(int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r.get("Grid_Size_X", r.get("Grid_Size",""))))
PY
done
