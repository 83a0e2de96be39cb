"""GPU: what a persistent recurrent launch does when its hand-off protocol is attacked from the INPUT side.

The chains hand data from link to link through slabs pre-filled with the word 0xFFFFFFFF; a consumer re-reads a fragment until no
word of it is that sentinel (csrc/pchain.h).  0xFFFFFFFF is a NaN whose payload no arithmetic on finite data produces — but NaN
propagation CAN carry it from an input (a weight, a waveform sample) into an output word, which a consumer would then wait for
forever.  Contract under test: every spin is bounded, a wave that gives up aborts its launch, the grid drains, and the host learns
about it (`blvm_async_errors_take`) — in bounded time, without a hang, and without poisoning later launches."""
import time

import pytest
import torch

from blvm import _hip
from blvm.models import VRNNAudio

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SENTINEL = torch.tensor([-1], dtype=torch.int32).view(torch.float32)  # bits 0xFFFFFFFF


def _model():
    torch.manual_seed(0)
    return VRNNAudio(likelihood="DMoL", input_size=16, hidden_size=64, latent_size=32, residual_posterior=True).to(DEV)


def _step(m, x, x_sl):
    loss, _, out = m(x, x_sl, beta=1.0, free_nats=2.0)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss)


@pytest.mark.timeout(120)
@pytest.mark.parametrize("where", ["weight", "input"])
def test_planted_sentinel_is_reported_not_waited_for(where):
    assert torch.cuda.is_available() and _hip.load().blvm_device_ok() == 1
    m = _model()
    B, T_ = 5, 16 * 12
    x = (torch.rand(B, T_, generator=torch.Generator().manual_seed(1)) - 0.5).to(DEV)
    x_sl = torch.full((B,), T_, dtype=torch.int64)
    _hip.take_async_errors()  # start from a clean count whatever ran before
    clean = _step(m, x, x_sl)
    assert torch.isfinite(torch.tensor(clean)) and _hip.take_async_errors() == (0, 0)

    if where == "weight":  # a weight the recurrent chain multiplies by in every step
        with torch.no_grad():
            m.vrnn.vrnn_cell.prior[2].weight[3, 5] = SENTINEL.to(DEV)[0]
        assert m.vrnn.vrnn_cell.prior[2].weight.view(torch.int32)[3, 5].item() == -1
    else:  # a sample of the waveform: reaches the chain through the encoder MLP
        x = x.clone()
        x.view(torch.int32)[2, 40] = -1
    m.zero_grad(set_to_none=True)
    t0 = time.time()
    poisoned = _step(m, x, x_sl)
    dt = time.time() - t0
    n, code = _hip.take_async_errors()
    print(f"planted sentinel in {where}: {n} aborted launch(es), code step {code >> 4} link {code & 15}, loss {poisoned}, {dt:.2f} s")
    # told either way, never silently wrong and never hung: an aborted launch is counted, or the NaN is visible in the result
    assert n >= 1 or poisoned != poisoned, (n, code, poisoned)
    assert dt < 60, f"{dt:.1f} s: the spins are bounded (~2.5 s each, the abort drains the rest of the grid)"
    assert _hip.take_async_errors() == (0, 0)  # read-and-clear: reported once

    # the next launches are unaffected (a new epoch per launch, nothing sticky on the device)
    m2 = _model()
    again = _step(m2, (torch.rand(B, T_, generator=torch.Generator().manual_seed(1)) - 0.5).to(DEV), x_sl)
    assert again == pytest.approx(clean, rel=1e-6)
    assert _hip.take_async_errors() == (0, 0)
