#!/usr/bin/env python
"""WaveNet on audio waveforms — entry point with the reference's flags (experiments/experiment_wavenet_audio.py);
no gradient clipping in this loop (:206-209)."""
from _common import run, wavenet_split_eval  # noqa: I001

from blvm.models import WaveNet
from blvm.modules.distributions import DiscretizedLogisticMixtureDense
from blvm.utils.argparsers import parser, str2bool

parser.set_defaults(lr=3e-4, epochs=3000, num_workers=8, save_checkpoints=True, optimizer="Adam")  # experiment_wavenet_audio.py:32-40
g = parser.add_argument_group("model")
g.add_argument("--n_layers", default=10, type=int)
g.add_argument("--n_stacks", default=4, type=int)
g.add_argument("--res_channels", default=64, type=int)
g.add_argument("--kernel_size", default=2, type=int)
g.add_argument("--base_dilation", default=2, type=int)
g.add_argument("--input_embedding_dim", default=1, type=int)
g.add_argument("--n_stack_frames", default=1, type=int)
g.add_argument("--generate_every", default=25, type=int)
g.add_argument("--input_coding", default="mu_law", type=str, choices=["mu_law", "linear"])
g.add_argument("--num_bits", default=16, type=int)
g.add_argument("--num_mix", default=10, type=int)
g.add_argument("--likelihood", default="DMoL", type=str)
g.add_argument("--random_segment_size", default=None, type=int)
g.add_argument("--split_eval", default=False, type=str2bool)

if __name__ == "__main__":
    args = parser.parse_args()
    if args.likelihood != "DMoL":
        raise NotImplementedError("libblvm_hip: WaveNet is built with the DMoL likelihood")
    lik = DiscretizedLogisticMixtureDense(args.res_channels, 1, num_mix=args.num_mix, num_bins=2**args.num_bits)
    model = WaveNet(likelihood=lik, n_layers=args.n_layers, n_stacks=args.n_stacks, res_channels=args.res_channels,
                    kernel_size=args.kernel_size, base_dilation=args.base_dilation, n_stack_frames=args.n_stack_frames)  # fmt: skip
    if args.split_eval and not args.random_segment_size:
        raise SystemExit("--split_eval True needs --random_segment_size (the split length, experiment_wavenet_audio.py:226)")

    def split_eval(model, x, x_sl, tracker):  # experiment_wavenet_audio.py:224-231 (benchmarks.txt:7 runs with --split_eval True)
        wavenet_split_eval(model, x, x_sl, tracker, args.random_segment_size)

    run(args, model, lambda m, x, sl: m(x, sl), lambda m, x, sl: m(x, sl), "loss", args.num_bits, split_eval if args.split_eval else None,
        clip=False)  # fmt: skip
