#!/usr/bin/env python
"""Clockwork VAE on audio waveforms — entry point with the reference's flags (experiments/experiment_clockwork_audio.py:
flags :31-66, model :84-96, train loop :213-232, split evaluation with carried per-level state :247-262, checkpoint
criterion "elbo (bpt)" :288)."""
from _common import cwvae_split_eval, run  # noqa: I001

import torch

from blvm.models import CWVAEAudio
from blvm.training.annealers import CosineAnnealer
from blvm.utils.argparsers import parser, str2bool

parser.set_defaults(epochs=1000, save_checkpoints=True, test_every=20, optimizer="Adam", lr=3e-4, lr_scheduler="MultiStepLR",
                    lr_scheduler_kwargs=dict(milestones=[1500, 3000, 4500], gamma=0.1), max_grad_norm=3000.0,
                    max_grad_value=1000.0)  # fmt: skip
g = parser.add_argument_group("model")
g.add_argument("--hidden_size", default=512, type=int, nargs="+")
g.add_argument("--latent_size", default=128, type=int, nargs="+")
g.add_argument("--global_size", default=0, type=int)
g.add_argument("--strides", default=[64, 16, 16], type=int, nargs="+")
g.add_argument("--stride_per_layer", default=2, type=int)
g.add_argument("--num_level_layers", default=8, type=int)
g.add_argument("--num_rssm_gru_cells", default=1, type=int)
g.add_argument("--num_bits", default=16, type=int)
g.add_argument("--num_mix", default=10, type=int)
g.add_argument("--residual_posterior", default=False, type=str2bool)
g.add_argument("--precision_posterior", default=False, type=str2bool)
g.add_argument("--random_segment_size", default=None, type=int)
g.add_argument("--coder_type", default="convolutional", type=str)
g.add_argument("--likelihood", default="DMoL", type=str)
g.add_argument("--input_coding", default="mu_law", type=str, choices=["mu_law", "linear"])
g.add_argument("--beta_anneal_steps", default=0, type=int)
g.add_argument("--beta_start_value", default=0, type=float)
g.add_argument("--free_nats_steps", default=0, type=int)
g.add_argument("--free_nats_start_value", default=4, type=float)
g.add_argument("--split_eval", default=False, type=str2bool)


def _one_or_list(v):
    return v[0] if isinstance(v, list) and len(v) == 1 else v


if __name__ == "__main__":
    args = parser.parse_args()
    model = CWVAEAudio(z_size=_one_or_list(args.latent_size), h_size=_one_or_list(args.hidden_size), g_size=args.global_size,
                       strides=args.strides, num_level_layers=args.num_level_layers, stride_per_layer=args.stride_per_layer,
                       num_mix=args.num_mix, num_bins=2**args.num_bits, residual_posterior=args.residual_posterior,
                       precision_posterior=args.precision_posterior, likelihood=args.likelihood)  # fmt: skip
    beta = CosineAnnealer(anneal_steps=args.beta_anneal_steps, start_value=args.beta_start_value, end_value=1)
    fn = CosineAnnealer(anneal_steps=args.free_nats_steps // 2, constant_steps=args.free_nats_steps // 2,
                        start_value=args.free_nats_start_value, end_value=0)  # fmt: skip

    def strideable(x):
        """Right-pad the collated batch to a multiple of the overall stride.  The reference derives the same-padding of
        level l > 0 from ceil(T / strides[l-1]) instead of the level's true input length (clockwork_vae.py:245); for lengths
        where the two disagree in parity its decoder context comes out one frame short and forward raises IndexError
        (reproduced here).  Multiples of the overall stride are always consistent; x_sl masks the padding."""
        pad = (-x.shape[1]) % int(model.overall_stride)
        return torch.nn.functional.pad(x, (0, pad)) if pad else x

    def split_eval(model, x, x_sl, tracker):  # per-level (z, h) carried between splits, same padding only on the last one
        cwvae_split_eval(model, x, x_sl, tracker, args.random_segment_size)

    run(args, model, lambda m, x, sl: m(strideable(x), sl, beta=beta.step(), free_nats=fn.step(), pad_strideable=True),
        lambda m, x, sl: m(strideable(x), sl, pad_strideable=True), "elbo (bpt)", args.num_bits,
        split_eval if args.split_eval and args.random_segment_size else None)  # fmt: skip
