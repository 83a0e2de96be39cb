import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "benchmarking-lvms_amd"))
from blvm import ops
from blvm._hip import load, ptr, stream_ptr
torch.manual_seed(0)
C, B, L, d = 96, 64, 4000, 4
lib = load()
x = torch.randn(L, B, C, device="cuda")
cw = torch.randn(2 * C, C, 2, device="cuda") * 0.05; cb = torch.randn(2 * C, device="cuda") * 0.1
rw = torch.randn(2 * C, C, device="cuda") * 0.05; rb = torch.randn(2 * C, device="cuda") * 0.1
res = torch.empty(lib.blvm_wavenet_block_reserve_floats(L, B, C, d), device="cuda")
ws = torch.empty(lib.blvm_wavenet_block_workspace_floats(L, B, C, C, d), device="cuda")
o = torch.empty(L - d, B, C, device="cuda"); skip = torch.zeros(1000, B, C, device="cuda")
for _ in range(3):
    rc = lib.blvm_wavenet_block_fwd(ptr(x), ptr(cw), ptr(cb), ptr(rw), ptr(rb), L, B, C, C, d, 1000, 0.7071, ptr(o), ptr(skip), ptr(res), ptr(ws), stream_ptr())
torch.cuda.synchronize()
rows = (L - d) * B
act = res[rows * 2 * C:]
names = ["stage X", "conv+gate", "store act", "rs+outputs"]
for w in range(4):
    v = act[8 * w: 8 * w + 4].cpu().tolist()
    print("wave", w, {n: int(c) for n, c in zip(names, v)}, "total", int(sum(v)))
