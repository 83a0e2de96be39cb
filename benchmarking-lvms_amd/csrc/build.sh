#!/bin/bash
# Build libblvm_hip.so for gfx950 in-tree (the .so travels to the GPU box with the repo snapshot).  One object per source, in
# parallel; BLVM_BUILD_JOBS bounds the parallelism (default: the CPUs this process may use, at most 8).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
JOBS="${BLVM_BUILD_JOBS:-$(n=$(nproc); echo $(( n > 8 ? 8 : n )))}"
make -s -C "$HERE" -j"$JOBS" "$@"
