"""Audio *file* I/O is never exercised: the oracle feeds synthetic tensors."""
from . import transforms  # noqa: F401


def info(*a, **k):
    raise RuntimeError("torchaudio is not available")


def load(*a, **k):
    raise RuntimeError("torchaudio is not available")
