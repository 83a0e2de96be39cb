"""What the data-parallel gradient exchange adds to a VRNN [64,16000] train step on ONE GPU (world size 1): the packing
passes alone, then with the RCCL all-reduce.  Run on the GPU box."""
import math
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "benchmarking-lvms_amd"))
from blvm.models import VRNNAudio  # noqa: E402
from blvm.training.ddp import FlatGradAllReduce  # noqa: E402

torch.manual_seed(0)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
m = VRNNAudio(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True).to(dev)
params = list(m.parameters())
opt = torch.optim.Adam(params, lr=3e-4)
g = torch.Generator().manual_seed(0)
u = (torch.rand(64, 16000, generator=g) * 2 - 1) * 0.5
x = (u.sign() * torch.log1p(65535 * u.abs()) / math.log(65536)).to(dev)
x_sl = torch.full((64,), 16000, dtype=torch.int64)
reducer = None


def step():
    opt.zero_grad(set_to_none=True)
    loss, _, _ = m(x, x_sl, beta=1.0, free_nats=2.0)
    loss.backward()
    if reducer is not None:
        reducer(64.0 * 16000)
    torch.nn.utils.clip_grad_value_(params, 1000.0)
    torch.nn.utils.clip_grad_norm_(params, 3000.0)
    opt.step()


def timeit(name, n=20):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{name:40s} {(time.perf_counter() - t0) / n * 1e3:7.2f} ms/step (host {th / n * 1e3:.2f})", flush=True)


timeit("no exchange")
reducer = FlatGradAllReduce(params)
timeit("packing passes only (no process group)")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
timeit("packing + RCCL all-reduce, world 1")
reducer = None
timeit("no exchange, process group alive")
dist.destroy_process_group()

# GPU time of the exchange alone (events around the call, queue drained first), and of its parts
reducer = FlatGradAllReduce(params)
step()
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
tot = 0.0
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e[0].record()
    reducer(64.0 * 16000)
    e[1].record()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tot += e[0].elapsed_time(e[1])
print(f"exchange alone: GPU {tot / 20:.3f} ms, host {th * 1e3:.3f} ms, {len(reducer.params)} tensors, {reducer.flat.numel() * 4 / 1e6:.1f} MB", flush=True)
