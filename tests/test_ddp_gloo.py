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
    red = FlatGradAllReduce(model.parameters())
    n_global = red(n_local, status=float(rank == 1))  # rank 1 reports one aborted launch: every rank must see it
    assert float(red.status) == 1.0
    # the collective form of the abort check (every rank raises together on the MAX of the counts): no abort -> no raise, on gloo too
    from blvm import _hip

    assert _hip.take_async_errors() == (0, 0)
    _hip.check_async("two-rank test", group=True)
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
    assert float(red.flat[:-2].norm()) <= 1e-3 * 1.001 and float(red.status) == 0.0
    # a second step with fresh gradients (zero_grad(set_to_none=True) semantics) overwrites the bucket
    for p in params:
        p.grad = None
    sum((p * p).sum() for p in params).backward()
    expect2 = [p.grad.clone() for p in params]
    red(10.0)
    for p, e in zip(params, expect2):
        torch.testing.assert_close(p.grad, e, rtol=1e-6, atol=0)


def test_clip_and_step_leaves_the_model_alone_when_the_gradient_norm_is_not_finite():
    """`skip_nonfinite` must do what experiment_srnn_audio.py:236-240 does — no update at all.  (Multiplying the gradients
    by isfinite(norm) = 0 leaves NaN * 0 = NaN in them, and Adam turns every parameter into NaN.)"""
    for p in (PKG, os.path.join(ROOT, "experiments"), ROOT):
        sys.path.insert(0, p)
    import _common as C

    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5))]
    opt = torch.optim.Adam(params, lr=0.1)
    for p in params:  # one ordinary step first, so that Adam holds moments that a wrong skip would disturb
        p.grad = torch.ones_like(p)
    assert C.clip_and_step(params, opt, 1000.0, 3000.0, skip_nonfinite=True)
    before = [p.detach().clone() for p in params]
    state = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in opt.state[p].items()} for p in params]
    for max_value in (1000.0, float("inf")):  # NaN survives the clamp by value; inf only when that clamp is off (inf -> max_value otherwise,
        for p in params:  #                      in the reference's loop as well: clip_grad_value_ runs first there too)
            p.grad = torch.ones_like(p)
        params[1].grad[2] = float("nan") if max_value == 1000.0 else float("inf")
        assert not C.clip_and_step(params, opt, max_value, 3000.0, skip_nonfinite=True)
        for p, b, st in zip(params, before, state):
            assert torch.equal(p.detach(), b) and torch.isfinite(p).all()
            for k, v in st.items():
                assert torch.equal(opt.state[p][k], v) if torch.is_tensor(v) else opt.state[p][k] == v
    for p in params:  # the run recovers: the next finite gradient is applied
        p.grad = torch.ones_like(p)
    assert C.clip_and_step(params, opt, 1000.0, 3000.0, skip_nonfinite=True)
    assert all(not torch.equal(p.detach(), b) and torch.isfinite(p).all() for p, b in zip(params, before))
    # without the flag the reference's VRNN loop steps regardless (experiment_vrnn_audio.py:224-228)
    params[0].grad[0, 0] = float("nan")
    assert C.clip_and_step(params, opt, 1000.0, 3000.0, skip_nonfinite=False)


def _steps_worker(rank, world, port, q):
    for p in (PKG, os.path.join(ROOT, "experiments"), ROOT):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from blvm.data.samplers import LengthEvalSampler, LengthTrainSampler
    from blvm.evaluation import LossMetric, Tracker

    import _common as C

    # 5 utterances in batches of 2 -> a 1-utterance tail batch: fewer examples than ranks
    train = C.SyntheticUtterances(5, 64, 2, 0, 8, seed=3, rank=rank, world=world, train=True)
    test = C.SyntheticUtterances(5, 64, 2, 0, 8, seed=3, rank=rank, world=world, train=False)
    n_train = sum(1 for _ in train)
    shards = [None if x is None else int(x.shape[0]) for x, _ in test]
    # the same rule in the length-bucketed samplers (lengths chosen so that the last batch holds one example)
    lens = [10, 10, 10, 10, 10]
    tr = LengthTrainSampler(lens, batch_len=20, min_pool_size=1, drop_last=False, shuffle=False, rank=rank, world_size=world)
    ev = LengthEvalSampler(lens, batch_len=20, rank=rank, world_size=world)
    tr_steps, ev_shards = [len(b) for b in tr], [len(b) for b in ev]
    # every training step is a blocking collective: both ranks must reach it the same number of times
    flat = torch.zeros(1)
    for _ in range(n_train):
        dist.all_reduce(flat)
    # metric merge across ranks = the reference's weighted running mean over all examples
    tracker = Tracker()
    if rank == 0:
        tracker.update([LossMetric(torch.tensor([1.0, 3.0]), weight_by=2)], source="test")
    else:
        tracker.update([LossMetric(torch.tensor([8.0]), weight_by=1)], source="test")
    tracker.all_reduce("test")
    q.put((rank, n_train, shards, tr_steps, ev_shards, len(tr), tracker.values("test")["loss"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_take_the_same_number_of_steps_with_a_one_example_tail_batch():
    """a global batch with fewer examples than ranks must not leave a rank without a step (the other would hang in
    the gradient all-reduce): training drops it on every rank, evaluation hands out an empty shard; metrics merge across ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_steps_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, n0, sh0, tr0, ev0, ltr0, v0), (_, n1, sh1, tr1, ev1, ltr1, v1) = res
    assert n0 == n1 == 2  # the 1-utterance training batch is dropped on both ranks
    assert sh0 == [1, 1, 1] and sh1 == [1, 1, None]  # evaluation keeps it: rank 1 gets an empty shard
    assert tr0 == tr1 == [1, 1] and ltr0 == ltr1 == 2
    assert sorted(ev0) == [1, 1, 1] and sorted(ev1) == [0, 1, 1]
    assert v0 == v1 == pytest.approx((1.0 + 3.0 + 8.0) / 3)


def test_bench_self_launches_two_ranks():
    """`bench.py --gpus 2` run bare starts its own two ranks (before any GPU call) and counts them with an all-reduce; a world
    size that differs from --gpus is refused in both directions."""
    import json
    import subprocess

    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2
    # a launcher-provided world that disagrees with --gpus is an error, not a silently mislabelled number
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], env=env2, capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in (bad.stderr + bad.stdout)
