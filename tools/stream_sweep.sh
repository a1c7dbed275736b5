#!/bin/bash
# headline bench for several render_streams settings, moving and static camera
for args in "--streams 1" "--streams 2" "--streams 4" "--streams 1 --static-camera" "--streams 4 --static-camera" "--streams 8 --static-camera"; do
  echo "== $args"
  timeout -k 10 120 python3 bench.py --no-extra --no-cpu-baseline --steps 40 --warmup 5 $args 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('Mrays/s %.0f ms/step %.3f trace_ms %.3f repeated %s'%(j['value'],j['ms_per_step'],j['roofline']['kernel_ms_per_frame'],j['config'].get('frames_repeated')))
" || exit 1
done
