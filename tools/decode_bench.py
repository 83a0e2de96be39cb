"""Single-step decode regime: milliseconds per generated frame stack / sample and the weight bytes that one
step has to stream, for VRNN, SRNN (B = 2, 16) and WaveNet (B = 1..64; window path and the one-launch cached kernel).
python tools/decode_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "benchmarking-lvms_amd"))
from blvm.models import SRNNAudio, VRNNAudio, WaveNet  # noqa: E402
from blvm.modules.distributions import DiscretizedLogisticMixtureDense  # noqa: E402


def timed(fn, n):
    fn(3)
    torch.cuda.synchronize()
    t = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def gpu_ms(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    from blvm import _hip, ops
    from blvm.models import CWVAEAudio

    torch.manual_seed(0)
    T = 250  # 1 s of audio at 64 samples per step
    for name, cls in (("VRNN", VRNNAudio), ("SRNN", SRNNAudio)):
        kw = dict(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True)
        if cls is SRNNAudio:
            kw["smoothing"] = True
        m = cls(**kw).cuda()
        # what one step multiplies: encoder, recurrent cell without its posterior, decoder, head
        if cls is VRNNAudio:
            used = [m.vrnn.encoder, m.vrnn.vrnn_cell.prior, m.vrnn.vrnn_cell.phi_z, m.vrnn.vrnn_cell.gru_cell, m.vrnn.decoder, m.vrnn.likelihood]
        else:
            used = [m.srnn.encoder, m.srnn.prior, m.srnn.d_forward_recurrent, m.srnn.decoder, m.srnn.likelihood]
        wbytes = sum(p.numel() for mod in used for p in mod.parameters()) * 4
        for B in (2, 16, 64, 128):
            gen = (lambda fused: m.generate(n_samples=B, max_timesteps=T, fused=fused)) if cls is VRNNAudio else (
                lambda fused: m.srnn.generate(x=torch.zeros(B, 1, 64, device="cuda"), n_samples=B, max_timesteps=T, fused=fused))
            step = timed(lambda n: (m.generate(n_samples=B, max_timesteps=n, fused=False) if cls is VRNNAudio else
                                    m.srnn.generate(x=torch.zeros(B, 1, 64, device="cuda"), n_samples=B, max_timesteps=n, fused=False)), 40) * 1e3
            one = gpu_ms(lambda: gen(True)) / T
            _hip.check_async()
            print(f"{name} generate B={B}: step by step {step:.3f} ms per 64-sample stack | one persistent launch {one:.4f} ms per stack "
                  f"({64 * B / one * 1e3:.3g} samples/s); weights read per step {wbytes / 1e6:.1f} MB -> {wbytes / one / 1e6:.0f} GB/s from the L2s", flush=True)
            if cls is VRNNAudio:
                real = ops.vrnn_decode
                ops.vrnn_decode = lambda *a, **k: real(*a, whole_chip=False, **k)
                k1c = gpu_ms(lambda: gen(True)) / T
                ops.vrnn_decode = real
                print(f"     one launch, 16 utterances per CU (K1c, round 1): {k1c:.4f} ms per stack", flush=True)
    # CW-VAE: one persistent launch per level (top-down), then the K11 context decoders and the head
    m = CWVAEAudio(z_size=[128, 64, 32], h_size=192, strides=[64, 16, 16], num_level_layers=8, stride_per_layer=2, likelihood="DMoL",
                   num_bins=2**16, precision_posterior=True).cuda()
    for B in (2, 8):
        n = 49152
        ms = gpu_ms(lambda: m.generate(n_samples=B, max_timesteps=n))
        _hip.check_async()
        print(f"CW-VAE generate B={B}, {n} samples: {ms:.1f} ms = {ms / (n // 64) * 1e3:.1f} us per bottom-level step ({B * n / ms * 1e3:.3g} samples/s)", flush=True)
    m = WaveNet(likelihood=DiscretizedLogisticMixtureDense(64, 1, num_mix=10, num_bins=2**16), n_layers=10, n_stacks=5, res_channels=64).cuda()
    wbytes = sum(p.numel() for p in m.parameters()) * 4
    for B in (1, 16, 64):
        dw = timed(lambda n: m.generate(B, n), 20)
        dc = timed(lambda n: m.generate(B, n, cached=True), 2000)
        print(f"WaveNet generate B={B}: window re-evaluation {dw * 1e3:.3f} ms/sample, cached one-launch kernel {dc * 1e3:.3f} ms/sample "
              f"({B / dc:.3g} samples/s; {wbytes / 1e6:.1f} MB of weights per sample and 16-utterance group -> {wbytes * ((B + 15) // 16) / dc / 1e9:.0f} GB/s)", flush=True)


if __name__ == "__main__":
    main()
