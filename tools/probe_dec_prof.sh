#!/bin/bash
# builds an instrumented copy of the library and runs the decode kernel's phase timers
set -e
cd /root/repo
C=benchmarking-lvms_amd/csrc
mkdir -p scratch/proflib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDEC_PROF -Iinclude -o scratch/proflib/libblvm_hip.so $C/*.hip
