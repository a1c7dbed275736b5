#!/bin/bash
# mirror / glass / csg / photon legs for several kernel variants: tools/extra_variants.sh name1 name2 ...
for v in "$@"; do
  echo "== $v"
  RRT_LIB=$PWD/variants/librrt_$v.so timeout -k 10 300 python3 bench_extra.py all 2>/dev/null | python3 -c "
import sys,json
t=sys.stdin.read(); j=json.loads(t[t.index('{'):])
for k,m in j.items():
    print('  %-22s'%k, ' '.join('%s=%s'%(a, ('%.4g'%m[a]) if isinstance(m[a],float) else m[a]) for a in ('value','ms_per_frame','ms','kernel_ms') if a in m), m.get('error',''))
" || exit 1
done
