"""Host enqueue time of each phase of a VRNN [64,16000] train step, with the queue drained before every step (a phase that
waits for the GPU shows up as long as the GPU work before it).  Run on the GPU box."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "benchmarking-lvms_amd"))
from blvm.models import VRNNAudio  # noqa: E402

torch.manual_seed(0)
dev = torch.device("cuda", 0)
m = VRNNAudio(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True).to(dev)
params = list(m.parameters())
opt = torch.optim.Adam(params, lr=3e-4)
g = torch.Generator().manual_seed(0)
u = (torch.rand(64, 16000, generator=g) * 2 - 1) * 0.5
x = (u.sign() * torch.log1p(65535 * u.abs()) / math.log(65536)).to(dev)
x_sl = torch.full((64,), 16000, dtype=torch.int64)
acc = {}


def mark(name, t):
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
    return time.perf_counter()


def step():
    torch.cuda.synchronize()
    t0 = t = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    t = mark("zero_grad", t)
    loss, _, _ = m(x, x_sl, beta=1.0, free_nats=2.0)
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
