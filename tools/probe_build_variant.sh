#!/bin/bash
# usage: tools/probe_build_variant.sh NAME [-Dflags...]  -> scratch/variants/NAME/libblvm_hip.so (objects in scratch/variants/NAME/obj)
set -e
cd "$(dirname "${BASH_SOURCE[0]}")/.."
N=$1; shift
mkdir -p scratch/variants/$N
make -s -C benchmarking-lvms_amd/csrc -j8 OBJ="$PWD/scratch/variants/$N/obj" LIB="$PWD/scratch/variants/$N/libblvm_hip.so" EXTRA="-w $*"
echo built $N
