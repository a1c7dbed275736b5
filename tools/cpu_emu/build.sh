#!/bin/bash
# g++ build of the device trace code for CPU emulation under ASan/UBSan
cd "$(dirname "$0")"
g++ -O1 -g -std=c++20 -pthread -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -I. -I../../rust-raytracer_amd/csrc -o emu emu.cpp
