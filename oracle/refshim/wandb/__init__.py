"""Experiment-logging no-ops (the reference imports wandb in its metrics/tracker modules)."""


class _Plot:
    @staticmethod
    def confusion_matrix(*a, **k):
        return None


plot = _Plot()
run = None


class Audio:
    def __init__(self, *a, **k):
        pass


def init(*a, **k):
    return None


def log(*a, **k):
    return None


def save(*a, **k):
    return None


def watch(*a, **k):
    return None


class Api:
    def __init__(self, *a, **k):
        raise RuntimeError("wandb is not available")
