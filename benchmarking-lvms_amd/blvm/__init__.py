"""blvm (MI355X-native): drop-in for the hot path of JakobHavtorn/benchmarking-lvms.

Same construction API as the reference's `blvm.models` / `blvm.modules`, same state_dict layout; the arithmetic of
the hot path runs in hand-written gfx950 HIP kernels behind `libblvm_hip.so` (see include/blvm_hip.h).
Unlike the reference (`blvm/settings.py:33-37`) importing this package never prompts for input.
"""
import os

__version__ = "0.1.0"

WANDB_PROJECT = os.environ.get("BLVM_WANDB_PROJECT", "blvm")
DATA_ROOT_DIRECTORY = os.environ.get("BLVM_DATA_ROOT_DIRECTORY", os.path.join(os.path.expanduser("~"), "blvm_data"))
DATA_DIRECTORY = os.path.join(DATA_ROOT_DIRECTORY, "data")
