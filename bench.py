#!/usr/bin/env python
"""bench.py — training throughput of the VRNN hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]

N > 1: one rank per GPU over RCCL.  Under `python -m torch.distributed.run ... bench.py --gpus N` the ranks come from the
launcher (RANK / LOCAL_RANK / WORLD_SIZE); run bare, `bench.py --gpus N` starts that launcher itself as a child process BEFORE
anything touches the GPU, forwards rank 0's JSON line and exits with the children's return code.  A world size that differs from
--gpus in either direction is an error.  `--dry-launch` walks the same launch path on the CPU (gloo, no GPU call): the CPU test
of the N-rank launch.

A "step" is one full optimisation step of VRNNAudio(DMoL, s=64, h=256, z=256) on a synthetic 16 kHz µ-law batch
[B,16000] per GPU that is resident in HBM: forward (encoder MLP, recurrent cell over T'=250 steps, decoder MLP, DMoL
head, KL) + backward (BPTT) + gradient all-reduce (N > 1) + clip-by-value + clip-by-norm + Adam — the loop body of
the reference's experiments/experiment_vrnn_audio.py:213-232 in fp32.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "benchmarking-lvms_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 (--dtype bf16 only)
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_*_f32 dense peak (= fp32 vector peak)
_T0 = time.time()


def log(msg):
    """Progress on stderr (stdout carries exactly one JSON line)."""
    print(f"[bench {time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def cell_macs(X, H, Z, R):
    prior = R * H + 2 * H * H + 2 * Z * H
    post = (R + X) * H + 2 * H * H + 2 * Z * H
    phi = Z * H + 3 * H * H
    gru = 3 * R * (X + H) + 3 * R * R
    return prior + post + phi + gru  # 2,686,976 at X=H=Z=256, R=512


def cwvae_macs(model, T):
    """Multiply-accumulates per utterance of one CW-VAE forward: 1x1 convs + depthwise taps of every block at its own rate,
    projections, RSSM cells (gru_in, GRU, 2x3-layer MLP + heads per step), DMoL head."""
    cw = model.cwvae
    h, macs = cw.h_size[0], 0
    lens = [T]
    for s in cw.strides:
        lens.append(math.ceil(lens[-1] / s))
    for coder in (cw.encoder, cw.decoder):
        for l, level in enumerate(coder.levels):
            n = lens[l] if not coder.transposed else lens[l + 1]
            for blk in level:  # lengths ignore the (small) same-padding
                n_out = n * blk.stride if coder.transposed else math.ceil(n / blk.stride)
                macs += n * h * 4 * h + n_out * (5 * 4 * h + 4 * h * h)
                n = n_out
    macs += T * h + T * h * h + T * h * 30  # encoder in-projection, decoder out-projection, DMoL Linear
    for l, cell in enumerate(cw.cells):
        Z, C, E = cell.z_dim, cell.c_dim, cell.e_dim
        per_step = (Z + C) * h + 6 * h * h + (h * h + 2 * h * h + 2 * Z * h) + ((h + E) * h + 2 * h * h + 2 * Z * h)
        macs += lens[l + 1] * (per_step + (Z + h) * h)  # + decoder in-projection of cat(z, h)
    return macs


def pmc_traffic(model, B, T):
    """HBM-side bytes per train step of the dominant kernel, from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE in separate runs, FETCH_SIZE doubled on gfx950 per MI355X_MICROARCH.md) — counters cannot be read from inside
    the timed process, so this is the profile of the same command, valid for the workload it was taken on only."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[0-9]_vrnn_pmc_traffic.json")))  # the latest round's passes
    if model != "vrnn" or (B, T) != (64, 16000) or not files:
        return None
    with open(files[-1]) as f:
        d = json.load(f)
    return d["cell_stage_kernels_read_bytes_per_step"] + d["cell_stage_kernels_write_bytes_per_step"]


def pmc_mfma(model, B, T, dtype):
    """Matrix-pipe utilisation of the recurrent cell's kernels from the committed rocprofv3 counter pass over this same command
    (tools/probe_pmc_mfma.sh: SQ_VALU_MFMA_BUSY_CYCLES over 4 SIMDs x 256 CUs x duration x effective clock, tools/pmc_mfma_summary.py):
    the persistent chain kernel, the grouped weight-gradient launches of the step (`gemm_group_kernel`: the chain's and the encoder /
    decoder MLPs'), and their duration-weighted mean (what `roofline.frac` estimates from time alone).  Counters cannot be read inside the timed process: valid for the workload the pass was taken on only."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r03_vrnn_pmc_mfma_v*.json")))
    if model != "vrnn" or (B, T) != (64, 16000) or dtype != "f32" or not files:
        return None
    with open(files[-1]) as f:
        d = json.load(f)
    p, w = d.get("pchain"), d.get("wgrad_gemm")
    if not p or not w:
        return None
    ms = p["ms_per_step"] + w["ms_per_step"]
    return dict(source=os.path.basename(files[-1]), cell=(p["mfma_busy"] * p["ms_per_step"] + w["mfma_busy"] * w["ms_per_step"]) / ms,
                pchain_kernel=p["mfma_busy"], wgrad_gemm=w["mfma_busy"], pchain_ms_per_step=p["ms_per_step"], wgrad_ms_per_step=w["ms_per_step"],
                pchain_waves_parked=p["parked"], pchain_clock_GHz=p["clock_GHz"])  # fmt: skip


def host_cpu():
    """(CPU model string, physical cores of the machine, logical CPUs this process may run on) from /proc/cpuinfo + the affinity
    mask.  Physical cores = distinct (physical id, core id) pairs; a container's CPU share may be smaller than either."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name":
                    model = v
                elif k == "physical id":
                    phys = v
                elif k == "core id":
                    core = v
                elif not k and phys is not None:
                    cores.add((phys, core))
                    phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    return model, len(cores) or (os.cpu_count() or 1), len(os.sched_getaffinity(0))


def cpu_baseline(B, T, steps, threads, warmup=5, gpu_batch=64):
    """The CPU oracle (a restatement of the reference's PyTorch path, pinned to it by golden vectors) timed on the
    host cores: forward + backward + clip + Adam on a bounded sample of the same workload, BASELINE.md §4's protocol
    (5 warm-up steps, median of >= 20 timed steps)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import blvm_oracle as O
    from blvm.models import VRNNAudio

    torch.set_num_threads(threads)
    torch.manual_seed(0)
    m = VRNNAudio(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True)
    sd = {k: v.clone().requires_grad_(True) for k, v in m.state_dict().items()}
    opt = torch.optim.Adam(list(sd.values()), lr=3e-4)
    x, x_sl = O.synth_batch(B, T, seed=0)
    Tp = math.ceil(T / 64)
    times = []
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        eps = torch.randn(Tp, B, 256)
        opt.zero_grad()
        out = O.vrnn_audio_forward(sd, x, x_sl, eps, beta=1.0, free_nats=2.0, stack=64)
        out["loss"].backward()
        torch.nn.utils.clip_grad_value_(list(sd.values()), 1000.0)
        torch.nn.utils.clip_grad_norm_(list(sd.values()), 3000.0)
        opt.step()
        if i % 5 == 4 or i == 0:
            log(f"cpu_baseline step {i}: {time.perf_counter() - t0:.2f} s")
        if i == 0:
            bpd0 = out["bpd"]  # random-init weights: comparable with the GPU line's bits_per_dim
        if i >= warmup:
            times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    cpu_model, physical, usable = host_cpu()
    why = "" if B == gpu_batch else f"; [{B},{T}] is NOT the GPU's [{gpu_batch},{T}] (the CPU path's frames/s grows with B: 3.5e5 at 16, 6.5e5 at 64 on 16 cores)"
    return dict(value=B * T / dt, unit="frames/s", cores=threads, kind="port",
                sample=f"oracle VRNN train step (fwd+bwd+clip+Adam, fp32, seeds data 0 / init 0) on [{B},{T}], median of {steps} steps after "
                       f"{warmup} warm-up, torch.set_num_threads({threads}){why}",
                cpu_model=cpu_model, physical_cores=physical, usable_cpus=usable,
                ms_per_step=dt * 1e3, bits_per_dim=bpd0, bits_per_dim_after_updates=out["bpd"])  # fmt: skip


def self_launch(n):
    """`bench.py --gpus N` without a launcher: start N ranks as children of a process that has only IMPORTED torch (no HIP call
    yet — a process that touched the GPU must never exec or be re-used as a launcher), pass their output through, return their rc."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]  # fmt: skip
    log(f"--gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def count_ranks(dev):
    """Number of ranks the collective backend really connects: an all-reduce of ones."""
    one = torch.ones(1, device=dev)
    dist.all_reduce(one)
    return int(one.item())


def dry_launch(args, rank, world):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    n = count_ranks("cpu")
    dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": dist.get_world_size(), "rccl_ranks": n, "backend": "gloo"}), flush=True)
    dist.destroy_process_group()
    if n != args.gpus:
        raise SystemExit(f"bench.py: all-reduce of ones counted {n} ranks, --gpus {args.gpus}")


def build_model(name, dev, S=64, H=256, Z=256):
    from blvm.models import STCN, CWVAEAudio, LSTMAudio, SRNNAudio, VRNNAudio, WaveNet
    from blvm.modules.distributions import DiscretizedLogisticMixtureDense

    torch.manual_seed(0)  # identical weights on every rank
    if name == "vrnn":
        return VRNNAudio(likelihood="DMoL", input_size=S, hidden_size=H, latent_size=Z, residual_posterior=True).to(dev)
    if name == "srnn":
        return SRNNAudio(likelihood="DMoL", input_size=S, hidden_size=H, latent_size=Z, residual_posterior=True, smoothing=True).to(dev)
    if name == "wavenet":
        lik = DiscretizedLogisticMixtureDense(96, 1, num_mix=10, num_bins=2**16)
        return WaveNet(likelihood=lik, n_layers=10, n_stacks=5, res_channels=96, kernel_size=2, base_dilation=2, n_stack_frames=1).to(dev)
    if name == "stcn":  # the reference's default STCN, 64-sample frames
        return STCN(likelihood="DMoL", n_layers=5, latent_size=[256, 128, 64, 32, 16], res_channels=256, n_stack_frames=S, dense=True).to(dev)
    if name == "cwvae":  # BASELINE config C4
        return CWVAEAudio(z_size=[128, 64, 32], h_size=192, strides=[64, 16, 16], num_level_layers=8, stride_per_layer=2,
                          precision_posterior=True, likelihood="DMoL", num_bins=2**16).to(dev)  # fmt: skip
    return LSTMAudio(stack_size=S, hidden_size=H, num_layers=1, num_mix=10, num_bins=2**16).to(dev)


def model_macs(name, model, B, T, S=64, H=256, Z=256):
    """(forward multiply-accumulates of the kernel the roofline is quoted on, its name)."""
    Tp, R = math.ceil(T / S), 2 * H
    if name == "vrnn":
        return cell_macs(H, H, Z, R) * B * Tp, "VRNN recurrent cell, forward+BPTT (blvm_vrnn_seq_fwd + blvm_vrnn_seq_bwd: recurrent chain + hoisted MFMA GEMMs)"
    if name == "srnn":
        return 2 * ((R + Z) * H + 2 * H * H + 2 * Z * H) * B * Tp, "SRNN latent chain, forward+BPTT (blvm_srnn_latent_fwd + blvm_srnn_latent_bwd)"
    if name == "wavenet":  # algorithmic MAC per frame, 5x10 blocks, C=96, k=2
        return 2777088 * B * T, "WaveNet whole train step (conv/MFMA path: 50 gated residual blocks as shifted-view GEMMs)"
    if name == "stcn":
        C, zs = 256, [256, 128, 64, 32, 16]
        per_step = S * C * 2 + C * C + 25 * (2 * C * 2 * C + C * 2 * C) + sum(zs) * C + 5 * (2 * C * 2 * C + C * 2 * C) + C * 30 * S + 900 * S
        for l, z in enumerate(zs):  # 4 three-layer MLPs per level (prior / posterior x mean / sd)
            cin = C + (zs[l + 1] if l + 1 < len(zs) else 0)
            per_step += 4 * (cin * C + C * C + C * z)
        return per_step * B * Tp, "STCN whole train step (time-parallel: 30 gated residual blocks + 20 latent MLPs as MFMA GEMMs, DMoL head)"
    if name == "cwvae":
        return cwvae_macs(model, T) * B, "CW-VAE whole train step (1x1-conv GEMMs of 48 separable blocks + 3 RSSM levels; the depthwise/norm passes are HBM-bound)"
    return (S * H + 2 * H * H + 8 * H * H + 2 * H * H + H * 30 * S + 900 * S) * B * (Tp - 1), "LSTMAudio whole train step (matmul FLOPs of embedding + LSTM + decoder + DMoL Linear)"


def measure(name, model, B, T, steps, warmup, rank, dev, use_dist, reducer_cls):
    """W untimed + K timed full train steps of `model` on a resident synthetic [B,T] batch.  Returns the wall time of the K
    steps between barrier + synchronize on both sides (MAX over ranks), every step's GPU duration from HIP events on the
    launching stream, and the HIP-event time of the recurrent-cell calls (the dominant kernels)."""
    from blvm import _hip, ops

    params = list(model.parameters())
    # the reference's optimizer (experiment_vrnn_audio.py:136: torch.optim.Adam) in torch's single-launch form: the same update
    # rule, one multi-tensor kernel instead of ~10 per step (BLVM_FUSED_ADAM=0: the foreach form)
    fused = os.environ.get("BLVM_FUSED_ADAM", "1") != "0"
    opt = torch.optim.Adam(params, lr=3e-4, fused=True) if fused else torch.optim.Adam(params, lr=3e-4, foreach=True)
    reducer = reducer_cls(params) if use_dist else None
    # synthetic mu-law batch, resident in HBM before the timed region (rank-offset seed: different utterances per GPU)
    g = torch.Generator().manual_seed(1000 + rank)
    u = (torch.rand(B, T, generator=g) * 2 - 1) * 0.5
    x = (u.sign() * torch.log1p(65535 * u.abs()) / math.log(65536)).to(dev)
    x_sl = torch.full((B,), T, dtype=torch.int64)
    torch.manual_seed(123 + rank)  # eps stream differs per rank

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    cur = {"i": -1}
    slot = {"fwd_begin": 0, "fwd_end": 1, "bwd_begin": 2, "bwd_end": 3}
    host = {"fwd": 0.0, "bwd": 0.0, "t": 0.0}

    def hook(tag):  # HIP events around the recurrent-cell calls, on the stream they are launched on
        if cur["i"] >= 0:
            ev[cur["i"]][slot[tag]].record()
            if tag.endswith("begin"):  # host-side enqueue time of the call
                host["t"] = time.perf_counter()
            else:
                host[tag[:3]] += time.perf_counter() - host["t"]

    ops.seq_timer_hook = hook
    free_nats = {"vrnn": 2.0, "srnn": 2.0, "cwvae": 4.0, "stcn": 4.0}.get(name)  # the experiment scripts' defaults
    bpd_name = {"cwvae": "elbo (bpt)", "stcn": "elbo (bpx)"}.get(name, "bpd")
    last = {}

    def step():
        opt.zero_grad(set_to_none=True)
        if free_nats is not None:
            loss, metrics, out = model(x, x_sl, beta=1.0, free_nats=free_nats)
        else:
            loss, metrics, out = model(x, x_sl)
        loss.backward()
        if reducer is not None:
            reducer(float(B * T))
        torch.nn.utils.clip_grad_value_(params, 1000.0)  # experiment_vrnn_audio.py:41-42 defaults
        torch.nn.utils.clip_grad_norm_(params, 3000.0)
        opt.step()
        last["metrics"] = metrics

    bpd0 = None
    _hip.take_async_errors()  # start from a clean count
    for i in range(warmup):
        step()
        if i == 0:
            torch.cuda.synchronize()
            bpd0 = {m.name: m.value for m in last["metrics"]}[bpd_name]  # forward of the random-init weights
            log(f"{name} [{B},{T}]: first step done")
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        cur["i"] = i
        step()
        marks[i + 1].record()
    cur["i"] = -1
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ops.seq_timer_hook = None
    # a persistent launch that gave up on a bounded spin DRAINS, i.e. gets faster: a step timed over one is not a measurement
    n_abort, code = _hip.take_async_errors()
    if use_dist:
        t = torch.tensor([n_abort], device=dev, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n_abort = int(t)
    if n_abort:
        log(f"{name} [{B},{T}]: {n_abort} persistent launch(es) ABORTED inside the timed region (last code step {code >> 4}, link {code & 15}): no number")
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    res = dict(dt=dt, ms_mean=dt / steps * 1e3, ms_median=per_step[len(per_step) // 2], ms_min=per_step[0], ms_max=per_step[-1],
               host_fwd_ms=host["fwd"] / steps * 1e3, host_bwd_ms=host["bwd"] / steps * 1e3, bpd_step0=bpd0, async_errors=n_abort,
               bpd_last={m.name: m.value for m in last["metrics"]}[bpd_name])  # fmt: skip
    if name in ("vrnn", "srnn"):
        cell = [(e[0].elapsed_time(e[1]), e[2].elapsed_time(e[3])) for e in ev]
        res["fwd_ms"] = sorted(c[0] for c in cell)[len(cell) // 2]
        res["bwd_ms"] = sorted(c[1] for c in cell)[len(cell) // 2]
    else:  # no hooked recurrent-cell call: whole step against the model's matmul FLOPs
        res["fwd_ms"], res["bwd_ms"] = res["ms_median"] / 3, 2 * res["ms_median"] / 3
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU (default 64 = the reference's --batch_len 64 s of audio; cwvae: 8)")
    ap.add_argument("--length", type=int, default=None, help="samples per utterance (default 16000 = 1 s at 16 kHz; cwvae: 49152)")
    ap.add_argument("--model", default="vrnn", choices=["vrnn", "srnn", "lstm", "wavenet", "cwvae", "stcn"], help="vrnn = BASELINE headline (configs[1])")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="operand type of the matrix products: f32 (headline) or bf16 operands / fp32 "
                    "accumulation for the persistent chains and K6 (the reference's --use_amp regime; a separate mode, never the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=64, help="utterances of the CPU baseline step (default: the GPU's 64; 25 steps of it are ~40 s on 16 cores)")
    ap.add_argument("--cpu-steps", type=int, default=20)
    ap.add_argument("--no-sweep", action="store_true", help="skip the large-batch sweep (N=1, vrnn only)")
    ap.add_argument("--dry-launch", action="store_true", help="CPU rehearsal of the N-rank launch: gloo, no GPU call, no model")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank number as {args.gpus} GPUs")
    if args.dry_launch:
        return dry_launch(args, rank, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # BLVM_BENCH_FORCE_DIST=1 under torch.distributed.run with one rank walks the whole RCCL path (init, gradient all-reduce,
    # barriers, MAX over ranks) on a one-GPU box; the numbers are those of N=1
    use_dist = world > 1 or (os.environ.get("BLVM_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    rccl_ranks = 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        rccl_ranks = count_ranks(dev)
        if rccl_ranks != args.gpus or dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: RCCL connects {rccl_ranks} ranks (world {dist.get_world_size()}), --gpus {args.gpus}")

    from blvm import _hip
    from blvm.training.ddp import FlatGradAllReduce

    assert _hip.load().blvm_device_ok() == 1, "libblvm_hip: no gfx950 device"
    _hip.set_operand_dtype(args.dtype)
    if args.batch is None:
        args.batch = 8 if args.model == "cwvae" else 64  # SURVEY §8d shapes per config (C4: [8, 49152] per GPU)
    if args.length is None:
        args.length = 49152 if args.model == "cwvae" else 16000
    B, T, S = args.batch, args.length, 64
    Tp = math.ceil(T / S)
    model = build_model(args.model, dev)
    log(f"rank {rank}/{world}: model resident, {args.warmup} warm-up + {args.steps} timed steps")
    m = measure(args.model, model, B, T, args.steps, args.warmup, rank, dev, use_dist, FlatGradAllReduce)
    log(f"timed {args.steps} steps: mean {m['ms_mean']:.2f} ms/step, median {m['ms_median']:.2f} (min {m['ms_min']:.2f}, max {m['ms_max']:.2f}); "
        f"host enqueue per step: seq_fwd {m['host_fwd_ms']:.2f} ms, seq_bwd {m['host_bwd_ms']:.2f} ms")
    n_ranks = dist.get_world_size() if use_dist else 1
    frames = n_ranks * B * T * args.steps
    macs, kname = model_macs(args.model, model, B, T)
    flops_fb = 3 * 2 * macs  # forward + dgrad + wgrad
    achieved = flops_fb / ((m["fwd_ms"] + m["bwd_ms"]) * 1e-3) / 1e12

    peak = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
    async_errors = m["async_errors"]
    if rank == 0:
        res = {
            "metric": f"audio frames/sec training ({args.model.upper()}, 16 kHz mu-law)",
            "value": frames / m["dt"],
            "unit": "frames/s",
            "n_gpus": n_ranks,
            "rccl_ranks": rccl_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": m["ms_median"],
            "ms_per_step_note": "median GPU time of the K timed steps (HIP events at step boundaries); `value` = frames / wall time of all K steps (mean)",
            "ms_per_step_mean": m["ms_mean"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "async_errors": 0,  # persistent launches that aborted inside any timed region (filled in below; non-zero = exit 1, no line)
            "bits_per_dim": m["bpd_step0"],
            "bits_per_dim_note": f"ELBO bits/dim of the first forward (random-init weights, seeded); after {args.warmup + args.steps} Adam updates on this one fixed batch: {m['bpd_last']:.4f}",
            "config": {
                "workload": {
                    "cwvae": "experiment_clockwork_audio.py: CWVAEAudio DMoL h=192 z=[128,64,32] strides [64,16,16] 8 blocks/level precision posterior",
                    "wavenet": "experiment_wavenet_audio.py: WaveNet DMoL 5 stacks x 10 layers, 96 channels, sample-level (50 gated residual blocks)",
                    "stcn": "experiment_stcn_audio.py: STCN DMoL 5 latent levels [256,128,64,32,16], 256 channels, 64-sample frames",
                    "lstm": f"experiment_lstm_audio.py: LSTMAudio DMoL s=64 h=256 (T'={Tp} recurrent steps)",
                }.get(args.model, f"experiment_{args.model}_audio.py: {type(model).__name__} DMoL s=64 h=256 z=256 (T'={Tp} recurrent steps)")
                + f", synthetic mu-law [{B},1,{T}] per GPU, full train step fwd+bwd+clip+Adam, random init",
                "batch_per_gpu": B, "global_batch": n_ranks * B, "samples_per_utterance": T, "parallelism": f"dp{n_ranks}",
            },
            "roofline": {
                "kernel": kname,
                "bound": "mfma",
                "achieved": achieved,
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": achieved / peak,
                "traffic": pmc_traffic(args.model, B, T),
                "mfma_busy": pmc_mfma(args.model, B, T, args.dtype),
                "flops_per_call": flops_fb,
                "fwd_ms": m["fwd_ms"],
                "bwd_ms": m["bwd_ms"],
            },
        }  # fmt: skip
        if n_ranks == 1 and not use_dist and not args.no_sweep and args.model == "vrnn" and (B, T) == (64, 16000):
            # the large-batch regime of the same kernels (the chain's cost per link does not depend on B until B ~ 256): same
            # model, same step, fewer timed steps.  Not the headline: `value` above stays the B = 64 configuration.
            sweep = [dict(batch_per_gpu=B, ms_per_step=m["ms_median"], frames_per_s=B * T / (m["ms_median"] * 1e-3), roofline_frac=achieved / peak)]
            # 128: the persistent launches on 16-row tiles; 256, 512: on row groups of four row tiles (csrc/pchain_rt.h); beyond:
            # a launch per link on 32x32 tiles, where the chain turns from latency- into throughput-bound
            for Bs in (128, 256, 512, 1024, 4096):
                ms_ = measure(args.model, model, Bs, T, 5, 2, rank, dev, False, FlatGradAllReduce)
                mc, _ = model_macs(args.model, model, Bs, T)
                tf = 6 * mc / ((ms_["fwd_ms"] + ms_["bwd_ms"]) * 1e-3) / 1e12
                async_errors += ms_["async_errors"]
                sweep.append(dict(batch_per_gpu=Bs, ms_per_step=ms_["ms_median"], frames_per_s=Bs * T / (ms_["ms_median"] * 1e-3), roofline_frac=tf / peak,
                                  async_errors=ms_["async_errors"]))  # fmt: skip
                log(f"sweep B={Bs}: {ms_['ms_median']:.1f} ms/step, cell {tf:.1f} TF/s")
            res["sweep"] = sweep
        if n_ranks == 1 and not args.no_cpu_baseline and args.model == "vrnn":
            # the box's CPU share for one GPU is 16 cores (more threads than that only thrash the cgroup quota)
            threads = max(1, min(16, os.cpu_count() or 1, len(os.sched_getaffinity(0))))
            res["cpu_baseline"] = cpu_baseline(args.cpu_batch, T, args.cpu_steps, threads, gpu_batch=B)
        res["async_errors"] = async_errors
        if not async_errors and math.isfinite(m["bpd_last"]):
            print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()
    if async_errors or not math.isfinite(m["bpd_last"]):
        raise SystemExit(f"bench.py: {async_errors} aborted persistent launch(es), bits/dim after the run {m['bpd_last']}: the timed steps are not valid")


if __name__ == "__main__":
    main()
