#!/bin/bash
# usage: tools/probe_build_variant.sh NAME [-Dflags...]  -> scratch/variants/NAME/libblvm_hip.so
set -e
cd /root/repo
C=benchmarking-lvms_amd/csrc
N=$1; shift
mkdir -p scratch/variants/$N
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -std=c++17 -fPIC -shared "$@" -Iinclude -o scratch/variants/$N/libblvm_hip.so $C/*.hip
echo built $N
