#!/bin/bash
# usage: tools/kernel_regs.sh csrc-file.hip [name-filter]  — registers / spills / scratch / LDS per kernel (compile only, no GPU)
f=$1; pat=${2:-.}
cd /root/repo/benchmarking-lvms_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-kernarg-preload-count=16 -I../../include -w \
  -Rpass-analysis=kernel-resource-usage -c -o /dev/null $f 2>&1 | grep -E "Function Name|VGPRs:|SGPRs:|Spill|ScratchSize|Occupancy" | paste - - - - - - - | grep -E "$pat" | sed -E 's/[^ ]*remark: [^ ]* //g; s/ \[-Rpass-analysis=kernel-resource-usage\]//g' | cut -c1-330
