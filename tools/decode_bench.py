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


def main():
    torch.manual_seed(0)
    for name, cls in (("VRNN", VRNNAudio), ("SRNN", SRNNAudio)):
        kw = dict(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True)
        if cls is SRNNAudio:
            kw["smoothing"] = True
        m = cls(**kw).cuda()
        wbytes = sum(p.numel() for p in m.parameters()) * 4
        for B in (2, 16, 64, 1024):
            dt = timed(lambda n: m.generate(n_samples=B, max_timesteps=n), 60)
            print(f"{name} generate B={B}: {dt * 1e3:.3f} ms per 64-sample stack ({64 * B / dt:.3g} samples/s); parameters {wbytes / 1e6:.1f} MB "
                  f"-> {wbytes / dt / 1e9:.0f} GB/s of weights per step", flush=True)
            if cls is VRNNAudio:  # K1c: every step in one launch; 12.5 MB of weights are read per step and 16-utterance group
                df = timed(lambda n: m.generate(n_samples=B, max_timesteps=n, fused=True), 400)
                print(f"{name} generate B={B}, one launch (K1c): {df * 1e3:.3f} ms per stack ({64 * B / df:.3g} samples/s) "
                      f"-> {12.5e6 * ((B + 15) // 16) / df / 1e9:.0f} GB/s of weights", flush=True)
    m = WaveNet(likelihood=DiscretizedLogisticMixtureDense(64, 1, num_mix=10, num_bins=2**16), n_layers=10, n_stacks=5, res_channels=64).cuda()
    wbytes = sum(p.numel() for p in m.parameters()) * 4
    for B in (1, 16, 64):
        dw = timed(lambda n: m.generate(B, n), 20)
        dc = timed(lambda n: m.generate(B, n, cached=True), 2000)
        print(f"WaveNet generate B={B}: window re-evaluation {dw * 1e3:.3f} ms/sample, cached one-launch kernel {dc * 1e3:.3f} ms/sample "
              f"({B / dc:.3g} samples/s; {wbytes / 1e6:.1f} MB of weights per sample and 16-utterance group -> {wbytes * ((B + 15) // 16) / dc / 1e9:.0f} GB/s)", flush=True)


if __name__ == "__main__":
    main()
