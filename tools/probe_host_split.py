import os, sys, time, torch, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "benchmarking-lvms_amd"))
from blvm.models import VRNNAudio
import blvm._hip as H
lib = H.load()
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
class Timed:
    def __init__(self, name, fn): self.name, self.fn = name, fn
    def __call__(self, *a):
        t = time.perf_counter(); r = self.fn(*a); acc[self.name] += time.perf_counter() - t; cnt[self.name] += 1; return r
class Wrap:
    def __init__(self, lib): object.__setattr__(self, "_lib", lib); object.__setattr__(self, "_c", {})
    def __getattr__(self, n):
        if n not in self._c: self._c[n] = Timed(n, getattr(self._lib, n))
        return self._c[n]
H._lib = Wrap(lib)
torch.manual_seed(0)
m = VRNNAudio(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True).cuda()
opt = torch.optim.Adam(m.parameters(), lr=3e-4)
import math
_g = torch.Generator().manual_seed(0)
_u = (torch.rand(64, 16000, generator=_g) * 2 - 1) * 0.5
x, x_sl = _u.sign() * torch.log1p(65535 * _u.abs()) / math.log(65536), torch.full((64,), 16000, dtype=torch.int64)  # synthetic mu-law batch
x = x.cuda()
def step():
    loss, _, _ = m(x, x_sl)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 1000.0)
    opt.step()
for _ in range(4): step()
torch.cuda.synchronize(); acc.clear(); cnt.clear()
n = 8; t0 = time.perf_counter()
for _ in range(n): step()
th = time.perf_counter() - t0; torch.cuda.synchronize(); tt = time.perf_counter() - t0
print(f"host {th/n*1e3:.2f} ms/step, total {tt/n*1e3:.2f} ms/step")
tot = 0
for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {k:34s} {v/n*1e3:7.3f} ms/step  ({cnt[k]//n} calls)"); tot += v
print(f"  all library calls {sum(acc.values())/n*1e3:.2f} ms/step; python + torch outside them {(th - sum(acc.values()))/n*1e3:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(8): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
