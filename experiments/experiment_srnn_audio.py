#!/usr/bin/env python
"""SRNN on audio waveforms — entry point with the reference's flags (experiments/experiment_srnn_audio.py)."""
from _common import run  # noqa: I001

from blvm.models import SRNNAudio
from blvm.training.annealers import CosineAnnealer
from blvm.utils.argparsers import parser, str2bool

parser.set_defaults(epochs=2000, batch_size=64, save_checkpoints=True, test_every=10, optimizer="Adam", lr=3e-4,
                    lr_scheduler="MultiStepLR", lr_scheduler_kwargs=dict(milestones=[1500, 3000, 4500], gamma=0.1),
                    max_grad_norm=3000.0, max_grad_value=1000.0)  # fmt: skip
g = parser.add_argument_group("model")
g.add_argument("--stack_frames", default=200, type=int)
g.add_argument("--hidden_size", default=512, type=int)
g.add_argument("--latent_size", default=256, type=int)
g.add_argument("--residual_posterior", default=True, type=str2bool)
g.add_argument("--smoothing", default=True, type=str2bool)
g.add_argument("--dropout", default=0.0, type=float)
g.add_argument("--input_coding", default="linear", type=str, choices=["mu_law", "linear"])
g.add_argument("--num_bits", default=8, type=int)
g.add_argument("--random_segment_size", default=None, type=int)
g.add_argument("--likelihood", default="GMM", type=str)
g.add_argument("--num_mix", default=10, type=int)
g.add_argument("--beta_anneal_steps", default=50000, type=int)
g.add_argument("--beta_start_value", default=0, type=float)
g.add_argument("--free_nats_steps", default=0, type=int)
g.add_argument("--free_nats_start_value", default=0.0625, type=float)
g.add_argument("--split_eval", default=False, type=str2bool)

if __name__ == "__main__":
    args = parser.parse_args()
    model = SRNNAudio(likelihood=args.likelihood, input_size=args.stack_frames, hidden_size=args.hidden_size,
                      latent_size=args.latent_size, num_mix=args.num_mix, num_bins=2**args.num_bits,
                      residual_posterior=args.residual_posterior, smoothing=args.smoothing, dropout=args.dropout)  # fmt: skip
    beta = CosineAnnealer(anneal_steps=args.beta_anneal_steps, start_value=args.beta_start_value, end_value=1)
    fn = CosineAnnealer(anneal_steps=args.free_nats_steps // 2, constant_steps=args.free_nats_steps // 2,
                        start_value=args.free_nats_start_value, end_value=0)  # fmt: skip

    def split_eval(model, x, x_sl, tracker):  # d_n, a_n, z_n carried between splits (experiment_srnn_audio.py:261-269)
        state = {}
        for xs, xs_sl in zip(*model.split_sequence(x, x_sl, length=args.random_segment_size)):
            _, metrics, out = model(xs, xs_sl, **state)
            tracker.update(metrics)
            state = dict(d_0=out.d_n, a_0=out.a_n, z_0=out.z_n)

    run(args, model, lambda m, x, sl: m(x, sl, beta=beta.step(), free_nats=fn.step()), lambda m, x, sl: m(x, sl), "elbo",
        args.num_bits, split_eval if args.split_eval and args.random_segment_size else None, skip_nonfinite=True)  # fmt: skip
