"""CPU, world_size=2 over gloo: the data-parallel gradient exchange (blvm/training/ddp.py) reproduces the
single-process gradient on the concatenated batch EXACTLY in the sense of the reference's loss definition
(loss normalised by the batch's total frame count, blvm/models/vrnn.py:277) — also for ragged shards, where a plain
mean of per-rank gradients would be wrong.  The per-rank gradients come from the CPU oracle (test infrastructure);
the code under test is the host-side bucket/scale/all-reduce logic, which is device-agnostic."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_grads(sd_vals, keys, x, x_sl, eps):
    import blvm_oracle as O

    sd = {k: v.clone().requires_grad_(True) for k, v in zip(keys, sd_vals)}
    out = O.vrnn_audio_forward(sd, x, x_sl, eps, beta=0.8, free_nats=1.0, stack=8)
    out["loss"].backward()
    return [sd[k].grad for k in keys], float(x_sl.sum())


def _worker(rank, world, port, q):
    for p in (PKG, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import blvm_oracle as O
    from blvm.models import VRNNAudio
    from blvm.training.ddp import FlatGradAllReduce

    torch.manual_seed(0)
    model = VRNNAudio(likelihood="DMoL", input_size=8, hidden_size=16, latent_size=16, residual_posterior=True)
    keys = [k for k, _ in model.named_parameters()]
    vals = [p.detach() for _, p in model.named_parameters()]
    B, T = 6, 40
    x, _ = O.synth_batch(B, T, seed=3)
    x_sl = torch.tensor([40, 37, 30, 22, 15, 9])  # ragged: the two shards hold 107 and 46 frames
    x = x * (torch.arange(T).unsqueeze(0) < x_sl.unsqueeze(1))
    eps = torch.randn(5, B, 16, generator=torch.Generator().manual_seed(1))
    sl = slice(rank * 3, rank * 3 + 3)
    Ts = int(x_sl[sl].max())
    grads, n_local = _oracle_grads(vals, keys, x[sl, :Ts], x_sl[sl], eps[: (Ts + 7) // 8, sl])
    for p, g in zip(model.parameters(), grads):
        p.grad = g.clone()
    n_global = FlatGradAllReduce(model.parameters())(n_local)
    if rank == 0:
        full, n_full = _oracle_grads(vals, keys, x, x_sl, eps)
        errs = [float((p.grad.double() - g.double()).norm() / (g.double().norm() + 1e-30)) for p, g in zip(model.parameters(), full)]
        naive = [float((a.double() - g.double()).norm() / (g.double().norm() + 1e-30)) for a, g in zip(grads, full)]
        q.put((max(errs), float(n_global), n_full, max(naive)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, n_global, n_full, naive = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert n_global == n_full == 153.0
    assert err < 1e-5, err  # fp32 round-off only
    assert naive > 1e-2  # a rank's own gradient is NOT the global one: the exchange is doing real work


def test_reducer_hands_out_bucket_slices_without_a_process_group():
    """Single process, no process group: the exchange is the identity on the gradients, and afterwards every p.grad is a slice
    of the flat bucket (no copy back) that clipping and the optimizer can work on in place."""
    sys.path.insert(0, PKG)
    from blvm.training.ddp import FlatGradAllReduce

    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(3, 5)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2, 2))]
    loss = sum((p * p).sum() * (i + 1) for i, p in enumerate(params))
    loss.backward()
    expect = [p.grad.clone() for p in params]
    red = FlatGradAllReduce(params)
    n = red(1234.0)
    assert float(n) == 1234.0
    lo, hi = red.flat.data_ptr(), red.flat.data_ptr() + red.flat.numel() * 4
    for p, e in zip(params, expect):
        assert lo <= p.grad.data_ptr() < hi, "gradient is not a slice of the bucket"
        torch.testing.assert_close(p.grad, e, rtol=1e-6, atol=0)
    torch.nn.utils.clip_grad_norm_(params, 1e-3)  # in place on the slices
    assert float(red.flat[:-1].norm()) <= 1e-3 * 1.001
    # a second step with fresh gradients (zero_grad(set_to_none=True) semantics) overwrites the bucket
    for p in params:
        p.grad = None
    sum((p * p).sum() for p in params).backward()
    expect2 = [p.grad.clone() for p in params]
    red(10.0)
    for p, e in zip(params, expect2):
        torch.testing.assert_close(p.grad, e, rtol=1e-6, atol=0)
