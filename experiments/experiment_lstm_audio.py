#!/usr/bin/env python
"""LSTM baseline on audio waveforms — entry point with the reference's flags (experiments/experiment_lstm_audio.py);
no gradient clipping in this loop (:186-189)."""
from _common import run  # noqa: I001

from blvm.models import LSTMAudio
from blvm.utils.argparsers import parser, str2bool

parser.set_defaults(epochs=2000, batch_size=40, save_checkpoints=True, test_every=5, optimizer="Adam", lr=3e-4,
                    lr_scheduler="MultiStepLR", lr_scheduler_kwargs=dict(milestones=[1500, 3000, 4500], gamma=0.1),
                    max_grad_norm=3000.0, max_grad_value=1000.0)  # declared but not applied by this loop (:186-189)  # fmt: skip
g = parser.add_argument_group("model")
g.add_argument("--stack_size", default=64, type=int)
g.add_argument("--hidden_size", default=256, type=int)
g.add_argument("--num_layers", default=1, type=int)
g.add_argument("--dropout", default=0.0, type=float)
g.add_argument("--input_coding", default="mu_law", type=str, choices=["mu_law", "linear"])
g.add_argument("--num_bits", default=16, type=int)
g.add_argument("--num_mix", default=10, type=int)
g.add_argument("--likelihood", default="DMoL", type=str)
g.add_argument("--random_segment_size", default=None, type=int)
g.add_argument("--split_eval", default=False, type=str2bool)

if __name__ == "__main__":
    args = parser.parse_args()
    if args.split_eval:
        # experiment_lstm_audio.py:205-207 calls model(xs, xs_sl, s0=s0) and reads output.sn, the model's names are s_0 / s_n
        # (lstm.py:76,129): the reference's split branch raises TypeError on its first split.  Refused up front here.
        raise SystemExit("--split_eval True: the reference's LSTM split evaluation cannot run (experiment_lstm_audio.py:205-207 passes "
                         "`s0=` to a forward whose argument is `s_0`); evaluate whole utterances, or carry `s_0` / `s_n` yourself")
    model = LSTMAudio(stack_size=args.stack_size, hidden_size=args.hidden_size, num_layers=args.num_layers,
                      dropout=args.dropout, num_mix=args.num_mix, num_bins=2**args.num_bits)  # fmt: skip
    run(args, model, lambda m, x, sl: m(x, sl), lambda m, x, sl: m(x, sl), "loss", args.num_bits, clip=False)
