#!/usr/bin/env python
"""STCN on audio waveforms — entry point with the reference's flags (experiments/experiment_stcn_audio.py: flags :29-67,
model :84-96, train loop :222-241, checkpoint criterion "elbo (bpx)" :286-287)."""
from _common import run  # noqa: I001

from blvm.models import STCN
from blvm.training.annealers import CosineAnnealer
from blvm.utils.argparsers import parser, str2bool

parser.set_defaults(epochs=1000, batch_size=20, save_checkpoints=True, test_every=20, optimizer="Adam", lr=3e-4,
                    lr_scheduler="MultiStepLR", lr_scheduler_kwargs=dict(milestones=[2500, 3500, 4500], gamma=0.1),
                    max_grad_norm=3000.0, max_grad_value=1000.0)  # fmt: skip
g = parser.add_argument_group("model")
g.add_argument("--hidden_size", default=256, type=int)
g.add_argument("--latent_size", default=[256, 128, 64, 32, 16], type=int, nargs="+")
g.add_argument("--num_layers", default=5, type=int)
g.add_argument("--num_stacks", default=None, type=int)
g.add_argument("--input_coding", default="linear", type=str, choices=["mu_law", "linear"])
g.add_argument("--num_bits", default=16, type=int)
g.add_argument("--num_mix", default=10, type=int)
g.add_argument("--dense", default=False, type=str2bool)
g.add_argument("--precision_posterior", default=True, type=str2bool)
g.add_argument("--top_down", default=True, type=str2bool)
g.add_argument("--num_stack_frames", default=200, type=int)
g.add_argument("--random_segment_size", default=16000, type=int)
g.add_argument("--likelihood", default="DMoL", type=str)
g.add_argument("--beta_anneal_steps", default=50000, type=int)
g.add_argument("--beta_start_value", default=0, type=float)
g.add_argument("--free_nats_steps", default=0, type=int)
g.add_argument("--free_nats_start_value", default=4, type=float)
g.add_argument("--split_eval", default=False, type=str2bool)

if __name__ == "__main__":
    args = parser.parse_args()
    if args.split_eval:
        # experiment_stcn_audio.py:258 calls model.split_sequence, which raises NotImplementedError (stcn.py:328-330)
        raise SystemExit("--split_eval True: STCN.split_sequence is not implemented in the reference (stcn.py:328-330: raises "
                         "NotImplementedError); `STCN.forward_split(x, x_sl, i_split)` itself is available")
    model = STCN(likelihood=args.likelihood, n_layers=args.num_layers, n_stacks=args.num_stacks, latent_size=args.latent_size,
                 res_channels=args.hidden_size, n_stack_frames=args.num_stack_frames, precision_posterior=args.precision_posterior,
                 dense=args.dense, top_down=args.top_down)  # fmt: skip
    beta = CosineAnnealer(anneal_steps=args.beta_anneal_steps, start_value=args.beta_start_value, end_value=1)
    fn = CosineAnnealer(anneal_steps=args.free_nats_steps // 2, constant_steps=args.free_nats_steps // 2,
                        start_value=args.free_nats_start_value, end_value=0)  # fmt: skip
    run(args, model, lambda m, x, sl: m(x, sl, beta=beta.step(), free_nats=fn.step()), lambda m, x, sl: m(x, sl), "elbo (bpx)",
        args.num_bits)  # fmt: skip
