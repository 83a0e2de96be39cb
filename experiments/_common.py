"""Shared body of the experiment entry points: synthetic length-bucketed data, one process per GPU, and the reference's
loop order (experiments/experiment_vrnn_audio.py:213-298): forward -> zero_grad -> backward -> [gradient all-reduce] ->
clip by value -> clip by norm -> optimizer step; lr_scheduler.step() per epoch; evaluation every `test_every` epochs,
optionally on split sequences with carried state; checkpoint when the test metric improves."""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "benchmarking-lvms_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from blvm import _hip  # noqa: E402
from blvm.data.transforms import MuLawEncode  # noqa: E402
from blvm.evaluation import Tracker  # noqa: E402
from blvm.training.ddp import FlatGradAllReduce  # noqa: E402


class SyntheticUtterances:
    """Batches of synthetic mu-law waveforms with the shape of the reference's collated batches: x [B,T] float32 in
    (-1,1) zero-padded on the right, x_sl [B] int64 on the host, longest first (batchers.py:145-151)."""

    def __init__(self, n_utterances, max_length, batch_size, batch_len, bits, seed, rank=0, world=1, min_frac=0.5, train=True):
        g = torch.Generator().manual_seed(seed)
        lens = (torch.rand(n_utterances, generator=g) * (1 - min_frac) + min_frac) * max_length
        lens = lens.long().clamp(min=1).sort(descending=True).values.tolist()
        self.batches, cur = [], []
        for n in lens:  # greedy buckets bounded by examples or by total samples (LengthTrainSampler idea)
            cur.append(n)
            full = len(cur) >= batch_size if batch_size else sum(cur) + n > batch_len
            if full:
                self.batches.append(cur)
                cur = []
        if cur:
            self.batches.append(cur)
        # utterances shard by rank.  Every rank must take the same number of TRAINING steps (each one is a blocking all-reduce):
        # a training batch with fewer utterances than ranks is dropped on every rank alike.  An evaluation batch is never dropped:
        # ranks beyond its size get an empty shard and skip it (no per-step collective in evaluation).
        if train:
            self.batches = [b for b in self.batches if len(b) >= world]
        self.batches = [b[rank::world] for b in self.batches]
        self.mulaw, self.seed, self.rank = MuLawEncode(bits), seed, rank

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        for i, lens in enumerate(self.batches):
            if not lens:  # empty evaluation shard
                yield None, None
                continue
            g = torch.Generator().manual_seed(self.seed * 7919 + i * 31 + self.rank)
            T = max(lens)
            x = self.mulaw((torch.rand(len(lens), T, generator=g) * 2 - 1) * 0.5)
            x_sl = torch.tensor(lens, dtype=torch.int64)
            yield (x * (torch.arange(T).unsqueeze(0) < x_sl.unsqueeze(1))), x_sl


def real_data_loaders(args, num_bits, rank, world, mu_law=True):
    """Source-CSV datasets behind the reference's pipeline: AudioLoader -> (RandomSegment) -> MuLawEncode -> padded, longest-first
    batches from the length-bucketed samplers; every rank builds the same batches and takes its strided share."""
    import random

    from blvm.data.base_dataset import BaseDataset
    from blvm.data.batchers import DynamicTensorBatcher
    from blvm.data.loaders import AudioLoader
    from blvm.data.samplers import LengthEvalSampler, LengthTrainSampler
    from blvm.data.transforms import Compose, RandomSegment

    random.seed(args.seed)  # identical batches on every rank
    enc = [MuLawEncode(num_bits)] if mu_law else []
    seg = getattr(args, "random_segment_size", None) or args.input_length
    tr_t = Compose(*([RandomSegment(seg)] if seg else []), *enc)
    bl = args.batch_len or 64 * 16000
    train_ds = BaseDataset(args.dataset, [(AudioLoader(args.audio_ext), tr_t, DynamicTensorBatcher())])
    test_ds = BaseDataset(args.test_source or args.dataset, [(AudioLoader(args.audio_ext), Compose(*enc), DynamicTensorBatcher())])
    lens = None
    if seg:  # the sampler must bucket by the CROPPED lengths
        from blvm.data.samplers.length_samplers import load_field
        lens = [min(n, seg) for n in load_field(args.dataset, "length")]
    tr_s = LengthTrainSampler(lens if lens is not None else args.dataset, batch_len=bl, min_pool_size=min(512, max(len(train_ds) // 4, 1)),
                              rank=rank, world_size=world)  # fmt: skip
    te_s = LengthEvalSampler(args.test_source or args.dataset, batch_len=bl, rank=rank, world_size=world)
    mk = lambda ds, sm: torch.utils.data.DataLoader(ds, batch_sampler=sm, collate_fn=ds.collate, num_workers=args.num_workers)  # noqa: E731

    class _XY:  # (x, x_sl) pairs like SyntheticUtterances
        def __init__(self, loader):
            self.loader = loader

        def __len__(self):
            return len(self.loader)

        def __iter__(self):
            for xy, _ in self.loader:  # an empty evaluation shard collates to (None, None)
                yield xy if xy is not None else (None, None)

    return _XY(mk(train_ds, tr_s)), _XY(mk(test_ds, te_s))


def setup(args):
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local_rank if args.device == "auto" else int(args.device))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", args.ddp_master_addr or "127.0.0.1")
        if args.ddp_master_port:
            os.environ.setdefault("MASTER_PORT", args.ddp_master_port)
        dist.init_process_group("nccl", device_id=dev)
    torch.manual_seed(args.seed)  # same weights on every rank
    if args.use_amp:
        # the reference wraps forward in torch.autocast (experiment_vrnn_audio.py:219-230); here: bf16 operands / fp32 accumulation
        # for the persistent recurrent chains and the K6 GEMMs, everything else (and everything stored) fp32
        from blvm import _hip

        _hip.set_operand_dtype("bf16")
        if rank == 0:
            print("note: --use_amp: bf16 matrix operands with fp32 accumulation (libblvm_hip operand dtype bf16)", file=sys.stderr)
    if isinstance(args.batch_len, float):
        args.batch_len = int(16000 * args.batch_len)
    return rank, world, dev


def wavenet_split_eval(model, x, x_sl, tracker, length):
    """The reference's split evaluation of WaveNet (experiments/experiment_wavenet_audio.py:224-231): splits that overlap by the
    receptive field (wavenet.py:230-242), receptive-field padding on the first one only, every split's metrics merged into the tracker."""
    out = None
    for i, (xs, xs_sl) in enumerate(zip(*model.split_sequence(x, x_sl, length=length))):
        _, metrics, out = model.forward_split(xs.contiguous(), xs_sl, i_split=i)
        tracker.update(metrics)
    return out


def cwvae_split_eval(model, x, x_sl, tracker, length):
    """The reference's split evaluation of the Clockwork VAE (experiments/experiment_clockwork_audio.py:255-271): strideable splits
    overlapping by rf - stride, per-level (z, h) carried from split to split, same padding on the last split only.  As in the
    reference this completes only for utterances that fit ONE split: `forward(pad_same=False)` raises IndexError there for every
    shape (tests/golden/split_eval.npz `cw_not_last_raises`), and so does this build."""
    state0, out, metrics = None, None, []
    xs_list, sl_list = model.split_sequence(x, x_sl, length=length)
    for i, (xs, xs_sl) in enumerate(zip(xs_list, sl_list)):
        _, metrics_, out = model.forward_split(xs.contiguous(), xs_sl, state0=state0, is_last_split=i == len(xs_list) - 1)
        metrics.extend(metrics_)
        state0 = [(z.contiguous(), h.contiguous()) for z, h in out.state_n]
    tracker.update(metrics)  # once per utterance batch, repeated names merged (tracker.update(metrics, check_unique=False), :271)
    return out


def clip_and_step(params, optimizer, max_grad_value, max_grad_norm, skip_nonfinite=False):
    """clip by value -> clip by norm -> optimizer step (experiment_vrnn_audio.py:224-228).  With `skip_nonfinite` a step whose
    gradient norm is NaN / inf is NOT taken (experiment_srnn_audio.py:236-240): parameters and optimizer state stay as they are.
    (Scaling the gradients by a 0/1 flag cannot do this: NaN * 0 is NaN, and `clip_grad_norm_` has already multiplied every
    gradient by a NaN coefficient.)  Like the reference's `if`, the test reads the norm on the host, i.e. waits for backward; in a
    data-parallel run the norm is that of the all-reduced gradient, so every rank decides alike.  Returns whether it stepped."""
    torch.nn.utils.clip_grad_value_(params, max_grad_value)
    total_norm = torch.nn.utils.clip_grad_norm_(params, max_grad_norm)
    if skip_nonfinite and not bool(torch.isfinite(total_norm)):
        return False
    optimizer.step()
    return True


def run(args, model, forward_train, forward_eval, best_metric, num_bits, split_eval_fn=None, clip=True, skip_nonfinite=False):
    """forward_train(model, x, x_sl) / forward_eval(model, x, x_sl) -> (loss, metrics, outputs)."""
    rank, world, dev = setup(args)
    model = model.to(dev)
    params = [p for p in model.parameters() if p.requires_grad]
    optimizer = getattr(torch.optim, args.optimizer or "Adam")(params, lr=args.lr, **args.optimizer_kwargs)
    scheduler = getattr(torch.optim.lr_scheduler, args.lr_scheduler)(optimizer, **args.lr_scheduler_kwargs)
    reducer = FlatGradAllReduce(params) if world > 1 else None
    bs, bl = args.batch_size, (args.batch_len or (0 if args.batch_size else 64 * 16000))
    if args.dataset == "synthetic":
        train = SyntheticUtterances(args.synthetic_utterances, args.synthetic_length, bs, bl, num_bits, args.seed, rank, world)
        test = SyntheticUtterances(max(args.synthetic_utterances // 8, 1), args.synthetic_length, bs, bl, num_bits, args.seed + 1, rank, world, train=False)
    else:
        train, test = real_data_loaders(args, num_bits, rank, world, mu_law=getattr(args, "input_coding", "mu_law") == "mu_law")
    tracker = Tracker()
    best = None
    for epoch in tracker.epochs(args.epochs):
        model.train()
        t0, frames, skipped, aborted = time.time(), 0, 0, 0
        for x, x_sl in tracker.steps(train, source="train"):
            x = x.to(dev, non_blocking=True)
            loss, metrics, _ = forward_train(model, x, x_sl)
            optimizer.zero_grad(set_to_none=True)
            loss.backward()
            if reducer is not None:
                reducer(float(x_sl.sum()), status=aborted)
            if clip:
                skipped += not clip_and_step(params, optimizer, args.max_grad_value, args.max_grad_norm, skip_nonfinite)
            else:
                optimizer.step()
            tracker.update(metrics)  # reads this step's sums back: the host has waited for everything enqueued so far
            # a persistent launch that gave up on a bounded spin leaves garbage: stop HERE, not at the end of the epoch.  One
            # process: raise at once.  Data-parallel: a rank that raised alone would leave the others blocked in their next
            # all-reduce, so the local count rides in the next gradient exchange and every rank raises on the summed count.
            aborted += _hip.take_async_errors()[0]
            if reducer is None and aborted:
                raise _hip.BlvmHipError(f"epoch {epoch}, step {tracker.step}: {aborted} persistent launch(es) aborted on a bounded spin; results are invalid")
            if reducer is not None and float(reducer.status) > 0:
                raise _hip.BlvmHipError(f"epoch {epoch}, step {tracker.step}: persistent launch(es) aborted on a bounded spin on some rank "
                                        f"({aborted} here); results are invalid")  # fmt: skip
            frames += int(x_sl.sum())
        scheduler.step()
        torch.cuda.synchronize()
        _hip.check_async("training epoch", group=world > 1)  # whatever the last exchange did not carry; collective: all ranks raise together
        if skipped and rank == 0:
            print(f"epoch {epoch:4d} | {skipped} step(s) skipped: non-finite gradient norm", flush=True)
        tracker.all_reduce("train")
        if rank == 0:
            vals = ", ".join(f"{k} {v:.4f}" for k, v in tracker.values("train").items())
            print(f"epoch {epoch:4d} | {frames * world / (time.time() - t0):.3e} frames/s | {vals}", flush=True)
        if (epoch - 1) % args.test_every == 0:
            model.eval()
            # the test ELBO decides about checkpoints and is what runs are compared by: always in fp32 operands, also under --use_amp
            # (the bf16-operand mode is a training-speed mode whose parity is unpinned)
            train_dtype = _hip.get_operand_dtype()
            _hip.set_operand_dtype("f32")
            with torch.no_grad():
                for x, x_sl in tracker.steps(test, source="test"):
                    if x is None:  # this rank's shard of the batch is empty
                        continue
                    x = x.to(dev, non_blocking=True)
                    if split_eval_fn is not None:
                        split_eval_fn(model, x, x_sl, tracker)
                    else:
                        tracker.update(forward_eval(model, x, x_sl)[1])
            torch.cuda.synchronize()
            _hip.set_operand_dtype(train_dtype)
            _hip.check_async("evaluation", group=world > 1)  # before anyone decides about a checkpoint
            tracker.all_reduce("test")  # the whole test set's value on every rank (rank 0 decides about the checkpoint)
            value = tracker.values("test").get(best_metric)
            if rank == 0 and value is not None:
                print(f"           test {best_metric} {value:.4f}", flush=True)
                improved = best is None or (value > best if best_metric == "elbo" else value < best)  # "elbo" in nats: max; bits / loss: min
                if improved:
                    best = value
                    if args.save_checkpoints and args.checkpoint_dir:
                        model.save(args.checkpoint_dir)
        tracker.log()
    if world > 1:
        dist.destroy_process_group()
    return tracker
