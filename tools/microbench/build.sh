#!/bin/bash
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../rust-raytracer_amd/csrc -o prim_cycles prim_cycles.hip
