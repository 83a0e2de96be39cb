"""Shared experiment argument parser with the reference's flag names and defaults (blvm/utils/argparsers.py:14-73).

Differences: `--dataset` defaults to `synthetic` (there is no dataset download on the target machines), the DDP flags
are honoured (the reference declares them but never reads them), wandb flags are accepted and ignored.
"""
import argparse
import json
import random


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def int_or_str(v):
    try:
        return int(v)
    except ValueError:
        return v


def float_or_str(v):
    try:
        return float(v)
    except ValueError:
        return v


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser()
    g = parser.add_argument_group("setup")
    g.add_argument("--seed", type=int, default=random.randint(0, 2**31 - 1))
    g.add_argument("--device", type=int_or_str, default="auto")
    g.add_argument("--use_amp", type=str2bool, default=False, help="bf16 matrix operands with fp32 accumulation while training (NOT the reference's fp16 autocast + GradScaler; evaluation stays fp32)")
    g.add_argument("--num_workers", type=int, default=8)
    g.add_argument("--save_checkpoints", type=str2bool, default=False)
    g.add_argument("--test_every", type=int, default=10)
    g = parser.add_argument_group("data")
    g.add_argument("--dataset", type=str, default="synthetic",
                   help="'synthetic', or the path of a TRAIN source CSV (`filename,length.<ext>.samples`, WAV files next to it)")
    g.add_argument("--test_source", type=str, default=None, help="source CSV of the test split (defaults to the train source)")
    g.add_argument("--audio_ext", type=str, default="wav")
    g.add_argument("--input_length", type=int, default=None, help="random training segment in samples (RandomSegment)")
    g.add_argument("--synthetic_utterances", type=int, default=256, help="utterances per synthetic epoch")
    g.add_argument("--synthetic_length", type=int, default=49152, help="maximum utterance length in samples (TIMIT-like 3 s)")
    g.add_argument("--checkpoint_dir", type=str, default=None)
    g = parser.add_argument_group("training")
    g.add_argument("--epochs", type=int, default=10)
    g.add_argument("--batch_size", type=int, default=0, help="Batch size in number of examples")
    g.add_argument("--batch_len", type=float_or_str, default=0, help="Batch size in sequence length (seconds if float)")
    g.add_argument("--lr", type=float, default=3e-4)
    g.add_argument("--length_sampler", type=str2bool, default=False)
    g = parser.add_argument_group("optimizer")
    g.add_argument("--optimizer", type=str, default=None)
    g.add_argument("--optimizer_kwargs", type=json.loads, default=dict())
    g.add_argument("--max_grad_norm", type=float, default=float("inf"))
    g.add_argument("--max_grad_value", type=float, default=float("inf"))
    g.add_argument("--lr_scheduler", type=str, default="ExponentialLR")
    g.add_argument("--lr_scheduler_kwargs", type=json.loads, default=dict(gamma=1))
    g = parser.add_argument_group("distributed data parallel")
    g.add_argument("--ddp_master_addr", default=None, type=str)
    g.add_argument("--ddp_master_port", default=None, type=str)
    g.add_argument("--nodes", "-n", default=None, type=int)
    g.add_argument("--gpus", "-g", default=None, type=int)
    g.add_argument("--node_rank", "-nr", default=None, type=int)
    g = parser.add_argument_group("wandb")
    for flag in ("entity", "project", "id", "name", "group", "notes", "resume", "mode", "job_type"):
        g.add_argument(f"--{flag}", type=str, default=None)
    g.add_argument("--tags", type=str, nargs="+", default=None)
    return parser


parser = build_parser()
