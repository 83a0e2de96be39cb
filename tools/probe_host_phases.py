"""Host enqueue time of each phase of a train step (default VRNN [64,16000]; `probe_host_phases.py wavenet 4` = bench.py's model of
that name at batch 4), with the queue drained before every step (a phase that waits for the GPU shows up as long as the GPU work
before it).  Run on the GPU box."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "benchmarking-lvms_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "vrnn"
Bn = int(sys.argv[2]) if len(sys.argv) > 2 else 64
torch.manual_seed(0)
dev = torch.device("cuda", 0)
m = bench.build_model(name, dev)
free_nats = {"vrnn": 2.0, "srnn": 2.0, "cwvae": 4.0, "stcn": 4.0}.get(name)
params = list(m.parameters())
opt = torch.optim.Adam(params, lr=3e-4)
g = torch.Generator().manual_seed(0)
Tn = int(sys.argv[3]) if len(sys.argv) > 3 else (49152 if name == "cwvae" else 16000)
u = (torch.rand(Bn, Tn, generator=g) * 2 - 1) * 0.5
x = (u.sign() * torch.log1p(65535 * u.abs()) / math.log(65536)).to(dev)
x_sl = torch.full((Bn,), Tn, dtype=torch.int64)
acc = {}


def mark(name, t):
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
    return time.perf_counter()


def step():
    torch.cuda.synchronize()
    t0 = t = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    t = mark("zero_grad", t)
    loss, _, _ = m(x, x_sl, beta=1.0, free_nats=free_nats) if free_nats is not None else m(x, x_sl)
    t = mark("forward", t)
    loss.backward()
    t = mark("backward", t)
    torch.nn.utils.clip_grad_value_(params, 1000.0)
    torch.nn.utils.clip_grad_norm_(params, 3000.0)
    t = mark("clip", t)
    opt.step()
    t = mark("adam", t)
    torch.cuda.synchronize()
    mark("drain", t)
    acc["total"] = acc.get("total", 0.0) + time.perf_counter() - t0


for _ in range(5):
    step()
acc.clear()
n = 20
for _ in range(n):
    step()
for k, v in acc.items():
    print(f"{k:10s} {v / n * 1e3:7.2f} ms", flush=True)
