#!/bin/bash
# mesh variants (phong / mirror / glass) of bench_extra for several kernel variants: tools/variant_extra.sh name1 name2 ...
for v in "$@"; do
  echo "== $v"
  RRT_LIB=$PWD/variants/librrt_$v.so timeout -k 10 200 python3 tools/r2_c4_breakdown.py 2>&1 | grep -v "amdgpu.ids\|^rrt:" || exit 1
done
