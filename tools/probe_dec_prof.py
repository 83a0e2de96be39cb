import os, sys, time, torch
sys.path.insert(0, "benchmarking-lvms_amd")
import blvm._hip as H
H._LIB_PATH = os.path.abspath("scratch/proflib/libblvm_hip.so")
from blvm import ops
import blvm.ops as O
# give x_out 12 extra floats for the phase timers
_empty = torch.empty
def patched_empty(*a, **k):
    if len(a) == 2 and k.get("dtype") == torch.float32 and a == (PB, PN):
        return _empty(PB * PN + 12, **k)[: PB * PN + 12]
    return _empty(*a, **k)
from blvm.models import WaveNet
from blvm.modules.distributions import DiscretizedLogisticMixtureDense
torch.manual_seed(0)
m = WaveNet(likelihood=DiscretizedLogisticMixtureDense(64, 1, num_mix=10, num_bins=2**16), n_layers=10, n_stacks=5, res_channels=64).cuda()
PB, PN = 4, 1000
# call ops.wavenet_decode by hand with an over-allocated output
import ctypes
lib = H.load()
rs, lik = m.res_stack, m.likelihood
t_in = rs.in_transform
hw, hb = lik.params.weight, lik.params.bias
pad_w = torch.zeros(2, 64, device="cuda")
parts = [m.causal.conv.weight, m.causal.conv.bias, t_in.weight.view(64, -1), t_in.bias, *(p for b in rs.res_blocks for p in b.kernel_params()),
         m.out_transform.linear.weight, m.out_transform.linear.bias, hw, pad_w, hb, torch.zeros(2, device="cuda")]
packed = torch.cat([p.detach().float().reshape(-1) for p in parts])
dil = (ctypes.c_int * 50)(*rs.dilations)
queues = torch.empty(lib.blvm_wavenet_decode_scratch_floats(dil, 50, PB, 64, 64), device="cuda")
x = torch.zeros(PB * PN + 12, device="cuda")
u = torch.empty(PN, PB, 10, device="cuda").uniform_(1e-5, 1 - 1e-5); v = torch.empty(PN, PB, device="cuda").uniform_(1e-8, 1 - 1e-8)
for _ in range(2):
    torch.cuda.synchronize(); t = time.time()
    rc = lib.blvm_wavenet_decode(packed.data_ptr(), dil, 50, PB, 64, 64, 64, 10, PN, rs.res_blocks[0].inv_std, 1.0 / m.variance_scale, -7.0,
                                 u.data_ptr(), v.data_ptr(), queues.data_ptr(), x.data_ptr(), None)
    torch.cuda.synchronize(); dt = time.time() - t
print("rc", rc, "ms/frame", dt / PN * 1e3)
p = x[PB * PN:].cpu().tolist()
names = ["a:stage", "barriers", "b:conv mfma+issue", "c:gate", "d:rs mfma", "-"]
for w0, lab in ((0, "thread 0"), (6, "last thread")):
    tot = sum(p[w0:w0 + 6])
    print(lab, {n: f"{1000*c/(PN*50):.0f} cyc" for n, c in zip(names, p[w0:w0 + 6])}, f"total/block {1000*tot/(PN*50):.0f} cyc")
