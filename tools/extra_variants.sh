#!/bin/bash
# mesh + photon legs for several kernel variants: tools/extra_variants.sh name1 name2 ...
for v in "$@"; do
  echo "== $v"
  RRT_LIB=$PWD/variants/librrt_$v.so timeout -k 10 200 python3 bench_extra.py all 2>&1 | python3 -c "
import sys,json
t=sys.stdin.read(); j=json.loads(t[t.index('{'):])
m=j['mesh_1m']; p=j['photon_1m']
print('mesh Mrays/s %.0f ms/frame %.3f | photon ms %s'%(m['value'],m['ms_per_frame'],p['ms']))
" || exit 1
done
