#!/bin/bash
# run bench.py headline for several kernel variants: tools/variant_bench.sh name1 name2 ...
for v in "$@"; do
  echo "== $v"
  RRT_LIB=$PWD/variants/librrt_$v.so timeout -k 10 120 python3 bench.py --no-extra --no-cpu-baseline --steps 30 --warmup 5 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('Mrays/s %.0f ms/step %.3f kernel_ms %.3f frac %.4f'%(j['value'],j['ms_per_step'],j['roofline']['kernel_ms_per_launch'],j['roofline']['frac']))
" || exit 1
done
