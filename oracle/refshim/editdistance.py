"""Only used by ErrorRateMetric (ASR probing; off the ELBO path)."""


def eval(a, b):  # noqa: A001
    raise RuntimeError("editdistance is not available")
