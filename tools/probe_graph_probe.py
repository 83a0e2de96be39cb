import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "benchmarking-lvms_amd"))
from blvm.models import VRNNAudio
torch.manual_seed(0)
m = VRNNAudio(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True).cuda()
import math
_g = torch.Generator().manual_seed(0)
_u = (torch.rand(64, 16000, generator=_g) * 2 - 1) * 0.5
x, x_sl = _u.sign() * torch.log1p(65535 * _u.abs()) / math.log(65536), torch.full((64,), 16000, dtype=torch.int64)  # synthetic mu-law batch
x = x.cuda()
def step(bwd):
    loss, _, _ = m(x, x_sl)
    if bwd:
        m.zero_grad(set_to_none=True)
        loss.backward()
for bwd in (False, True):
    for _ in range(4): step(bwd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 8
    for _ in range(n): step(bwd)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"graphs={os.environ.get('BLVM_GRAPHS', '1')} bwd={bwd}: host enqueue {t_host / n * 1e3:.2f} ms/step, total {t_all / n * 1e3:.2f} ms/step", flush=True)
