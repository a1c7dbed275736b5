#!/bin/bash
# g++ + ASan/UBSan build of the scene-file fuzzer from librrt's host sources (no HIP runtime needed)
cd "$(dirname "$0")"
S=../../rust-raytracer_amd/csrc
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I../../include -I$S \
    fuzz_scene_file.cpp $S/scene_file.cpp $S/host_io.cpp $S/host_scene.cpp -o fuzz_scene_file
