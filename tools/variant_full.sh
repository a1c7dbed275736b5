#!/bin/bash
# bench.py with its extra legs (C2 frame, C3 photon pass and frame) for several kernel variants: tools/variant_full.sh name1 name2 ...
for v in "$@"; do
  echo "== $v"
  RRT_LIB=$PWD/variants/librrt_$v.so timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); e=j['extra']
        print('C4 Mrays/s %.0f ms/step %.3f kernel_ms %.3f'%(j['value'],j['ms_per_step'],j['roofline']['kernel_ms_per_launch']))
        c=e['csg_1080p']; print('C2 Mrays/s %.0f ms/frame %.3f kernels %.3f; 3 in flight %.0f'%(c['value'],c['ms_per_frame'],c['roofline']['kernel_ms_per_frame'],c['value_3_frames_in_flight']))
        p=e['photon_1m']; print('C3', {k:(round(v,3) if isinstance(v,float) else v) for k,v in p.items() if not isinstance(v,(dict,str))})
" || exit 1
done
