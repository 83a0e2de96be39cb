#!/bin/bash
# Build libblvm_hip.so for gfx950 in-tree (the .so travels to the GPU box with the repo snapshot).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../blvm/lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -amdgpu-kernarg-preload-count: kernels with scalar arguments get their first 14 argument dwords preloaded into SGPRs by the
# command processor (stages.h lin1_stage_kernel); kernels with struct arguments are unaffected
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-kernarg-preload-count=16 -I"$HERE/../../include" \
  -o "$OUT/libblvm_hip.so" "$HERE"/core.hip "$HERE"/pchain.hip "$HERE"/gemm.hip "$HERE"/dmol.hip "$HERE"/kl.hip "$HERE"/vrnn.hip "$HERE"/vrnn_decode.hip "$HERE"/rnn.hip "$HERE"/srnn.hip "$HERE"/srnn_decode.hip "$HERE"/seqchain.hip "$HERE"/wavenet.hip "$HERE"/wavenet_decode.hip "$HERE"/rssm.hip "$HERE"/convcoder.hip
echo "built $OUT/libblvm_hip.so"
