import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "benchmarking-lvms_amd"))
from blvm import ops, _hip
from blvm._hip import ptr, stream_ptr, check
lib = _hip.load(); dev = "cuda:0"
B, Tp, Z = 64, 250, 256
g = torch.Generator().manual_seed(0)
mq, mp = torch.randn(Tp, B, Z, generator=g).to(dev), torch.randn(Tp, B, Z, generator=g).to(dev)
sq, sp = (torch.rand(Tp, B, Z, generator=g) + 0.5).to(dev), (torch.rand(Tp, B, Z, generator=g) + 0.5).to(dev)
x_sl = torch.full((B,), Tp * 64, dtype=torch.int32, device=dev)
for fn in (0.0, 2.0 / 256):
    kld = torch.zeros(B, device=dev, dtype=torch.float64); kfn = torch.zeros(B, device=dev, dtype=torch.float64)
    def f():
        check(lib.blvm_kl_fwd(ptr(mq), ptr(sq), ptr(mp), ptr(sp), ops.LAYOUT_TIME_MAJOR, ptr(x_sl), B, Tp, Z, 64, fn, ptr(kld), ptr(kfn), stream_ptr()), "kl")
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"kl_fwd [64,250,256] free_nats={fn:.4f}: {us:.1f} us = {4*B*Tp*Z*4/us/1e6:.2f} TB/s")
